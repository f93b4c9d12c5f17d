// include/gi/raytracer.h -- drop-in for the reference's RayTracer (include/raytracer.h:23-735) on the MI355X path.
// Same public data (photons, photon_depth, min_samples, max_samples, noise_thresh, ambient, _camera, width, height, p), same life cycle
// (RayTracer(camera); setScene(octree); start(); run(w, h) on a worker thread while the GUI thread polls getImage() every 32 ms; stop()),
// copyable with shared scene / photon map / image exactly as gui.h:19 / viewer.h:16 need, and the per-ray methods with the reference's
// signatures (trace, visible, samplePhotons, radiance, tracePhotons, secondaryRay, rayType, raymarch).
//
// run() flattens + uploads when the scene is not valid, emits photons on the GPU when the photon map is not valid, then renders the
// frame in stripes (`progressive_rows` = 16 rows first, doubling while a step is shorter than `progressive_ms`) through gi_render_host: after every stripe its pixels go through gamma 2.2 / clamp / the
// truncating 8-bit store into the shared Image (the display frame fills top to bottom as the reference's does row by row,
// include/raytracer.h:93-160), and `_running` is polled (include/raytracer.h:98) -- stop() ends run() within one stripe.
// trace / visible / samplePhotons / radiance / tracePhotons run on the GPU through the C ABI (one-element batches); there is no CPU
// renderer behind this class: without a usable HIP device run() leaves the image black and last_error() says why.
// secondaryRay / rayType / raymarch are the kernels' per-lane functions evaluated on the host for a single vertex (detail.h) -- the
// reference draws drand() inside them; here the draws come from the counter RNG keyed by (seed, rng_stream, rng_depth).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>
#include "../gi_hip.h"
#include "camera.h"
#include "entities.h"
#include "halton_enum.h"
#include "halton_sampler.h"
#include "image.h"
#include "octree.h"
#include "photonMap.h"

class RayTracer {
  public:
    int p = 0;
    int width = 0, height = 0;

    RayTracer() = delete;
    RayTracer(const Camera& camera) : _camera(camera), _st(std::make_shared<State>()), _image(std::make_shared<Image>(0, 0)) {}

    void setScene(Octree* scene)   // include/raytracer.h:35-39: the photon map lives in the scene's root box
    {
        _scene = scene;
        _st->photon_map = std::make_shared<PhotonMap>(scene->_root._bbox.min, scene->_root._bbox.max);
        _st->uploaded = false;
        _st->photons_uploaded = false;
        _st->fresh_map = true;
    }

    // RayTracer::run(w, h), include/raytracer.h:41-165
    void run(int w, int h)
    {
        _image = std::make_shared<Image>(w, h);
        width = w; height = h;
        if (!prepare()) return;
        _linear.assign((size_t)w * h * 3, 0.f);
        if (_st->group) { run_on_group(w, h); return; }
        int rows = progressive_rows > 0 ? std::min(progressive_rows, h) : h;
        std::vector<double> lin;
        std::vector<uint8_t> rgb;
        for (int y0 = 0; y0 < h;) {
            if (!_running) return;                                   // RayTracer::stop(), polled per row in the reference (include/raytracer.h:98)
            gi_render_params rp = params(w, h);
            rp.stripe_h = rows; rp.stripe_rank = y0 / rows; rp.stripe_world = (h + rows - 1) / rows;   // this call renders one stripe only (y0 is a multiple of rows)
            const int nr = std::min(rows, h - y0);
            rows_in_flight = nr;
            lin.resize((size_t)nr * w * 3);
            _cancel = _running ? 0 : 1;
            const auto t0 = std::chrono::steady_clock::now();
            if (check(gi_render_host(_st->ctx, &rp, lin.data(), 1, nullptr, &_cancel)) != 0) return;
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            show_rows(lin.data(), rgb, y0, nr, w);
            y0 += nr;
            rows = next_rows(rows, y0, ms);
        }
    }

  private:
    // Stripes grow while a step is shorter than `progressive_ms`: a render call has a floor of some 20 ms whatever its size (the chain of up to 64
    // dependent bounces of its deepest path), so a frame in 16-row steps would spend nearly all its time there -- the first stripe appears at once,
    // the following ones double (16, 16, 32, 64, ...; the next start row stays a multiple of the stripe height, which is what the stripe
    // parameters of the C ABI address) until a step takes long enough for the viewer's 32 ms repaint to show progress.
    int next_rows(int rows, int y0, double ms) const
    {
        return progressive_ms > 0 && progressive_rows > 0 && ms < progressive_ms && y0 % (2 * rows) == 0 ? 2 * rows : rows;
    }
    // the same frame over every visible GPU (gi_group_*, include/gi_hip.h): each progressive step renders one stripe per device
    void run_on_group(int w, int h)
    {
        const int n = gi_group_size(_st->group);
        int rows = progressive_rows > 0 ? std::min(progressive_rows, h) : (h + n - 1) / n;
        std::vector<double> frame((size_t)w * h * 3, 0.0);
        std::vector<uint8_t> rgb;
        for (int y0 = 0; y0 < h;) {
            if (!_running) return;
            gi_render_params rp = params(w, h);
            const int total = (h + rows - 1) / rows, k = y0 / rows;
            const int ns = std::min(n, total - k), nr = std::min(ns * rows, h - y0);
            rows_in_flight = nr;
            _cancel = _running ? 0 : 1;
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = gi_group_render_host(_st->group, &rp, rows, k, ns, frame.data(), 1, nullptr, &_cancel);
            if (rc != 0) { if (rc != GI_E_CANCELLED) { _st->err = gi_group_last_error(_st->group); fprintf(stderr, "gi: %s\n", _st->err.c_str()); } return; }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            show_rows(frame.data() + (size_t)y0 * w * 3, rgb, y0, nr, w);
            y0 += nr;
            rows = next_rows(rows, y0, ms);
        }
    }
    // gamma(color, 2.2), glm::clamp, (int)(255 c) of rows [y0, y0 + nr) into the shared image: include/raytracer.h:150-157
    void show_rows(const double* lin, std::vector<uint8_t>& rgb, int y0, int nr, int w)
    {
        const size_t n = (size_t)nr * w * 3;
        rgb.resize(n);
        gih_to_rgb8(lin, 1, (int64_t)n, rgb.data());
        for (int y = 0; y < nr; y++)
            for (int x = 0; x < w; x++) _image->setPixel8(x, y0 + y, &rgb[((size_t)y * w + x) * 3]);
        for (size_t i = 0; i < n; i++) _linear[(size_t)y0 * w * 3 + i] = (float)lin[i];
        rows_done = y0 + nr;
    }

  public:

    // ---- per-ray methods, signatures of include/raytracer.h:167,280,382,532,582 -- each is a one-element batch on the GPU
    bool trace(const Ray& ray, gi::dvec3& minHit, gi::dvec3& minNorm, gi::dvec2& minUV, Entity*& obj)
    {
        if (!ready()) return false;
        const double r[6] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.dir.x, ray.dir.y, ray.dir.z};
        int32_t hit = 0, ent = -1;
        double res[8];
        if (check(gi_trace(_st->ctx, 1, r, &hit, &ent, res)) != 0 || !hit) return false;
        minHit = gi::get3(res); minNorm = gi::get3(res + 3);
        obj = _scene->entities()[(size_t)ent];
        // `uv` as Entity::intersect leaves it: interpolated texCoords of a smooth triangle, asin / atan2 of a sphere, untouched for a flat triangle
        gi::dvec3 h2(0, 0, 0), n2(0, 0, 0);
        gi::dvec2 uv = minUV;
        if (obj->intersect(ray, h2, n2, uv)) minUV = uv;
        return true;
    }
    bool visible(const Ray& ray, double mt)
    {
        if (!ready()) return false;
        const double r[6] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.dir.x, ray.dir.y, ray.dir.z};
        int32_t vis = 0;
        return check(gi_visible_rays(_st->ctx, 1, r, &mt, &vis)) == 0 && vis != 0;
    }
    gi::dvec3 samplePhotons(gi::dvec3 pos, gi::dvec3 dir, int count)
    {
        if (count != GI_GATHER_K) { _st->err = "samplePhotons: the gather kernel selects 32 photons (the only count the reference uses)"; return gi::dvec3(0, 0, 0); }
        if (!ready() || !photons_ready()) return gi::dvec3(0, 0, 0);
        const double q[6] = {pos.x, pos.y, pos.z, dir.x, dir.y, dir.z};
        double res[3] = {0, 0, 0};
        check(gi_gather(_st->ctx, 1, q, res, nullptr));
        return gi::get3(res);
    }
    // radiance(ray, 0, sampler, enumerator, sample, (1,1,1)): `sample` is the Halton index = the stream of the counter RNG
    gi::dvec3 radiance(const Ray& ray, int depth, const Halton_sampler&, const Halton_enum&, int sample, gi::dvec3 contrib)
    {
        if (depth != 0 || contrib.x != 1 || contrib.y != 1 || contrib.z != 1) { _st->err = "radiance: paths start at depth 0 with contrib 1 (the recursion is a loop inside the kernel)"; return gi::dvec3(0, 0, 0); }
        if (!ready() || !photons_ready()) return gi::dvec3(0, 0, 0);
        const double r[6] = {ray.origin.x, ray.origin.y, ray.origin.z, ray.dir.x, ray.dir.y, ray.dir.z};
        const uint32_t stream = (uint32_t)sample;
        double out[3] = {0, 0, 0};
        check(gi_radiance(_st->ctx, 1, r, &stream, seed, out));
        return gi::get3(out);
    }
    // photon emission on the GPU; the stored photons are appended to the map (include/raytracer.h:582-715)
    void tracePhotons(int maxDepth, int count, Halton_sampler&, Halton_enum&) { tracePhotons(maxDepth, count); }
    void tracePhotons(int maxDepth, int count)
    {
        if (!ready() || count <= 0 || _scene->lights.empty()) return;
        std::vector<double> out((size_t)count * _scene->lights.size() * 9);
        int64_t tries = 0;
        const int n = gi_emit_photons(_st->ctx, count, maxDepth, seed, out.data(), (int32_t)(out.size() / 9), &tries);
        if (n < 0) { check(n); return; }
        _st->photon_map->reserve(_st->photon_map->size() + n);
        for (int i = 0; i < n; i++) _st->photon_map->push_back(new Photon(gi::get3(&out[(size_t)i * 9]), gi::get3(&out[(size_t)i * 9 + 3]), gi::get3(&out[(size_t)i * 9 + 6])));
    }

    // ---- single-vertex pieces of radiance(), include/raytracer.h:321-379,481-506,509-529: the kernels' per-lane functions on the host
    uint32_t rng_stream = 0, rng_depth = 0;   // key of the counter RNG for the draws these three make (the reference calls drand())
    int rayType(const Entity* entity, const Ray& ray, gi::dvec3& norm, gi::dvec2& minUV)
    {
        const gi::Mat m = mat_of(entity);
        gi::dvec2 uv = minUV;
        return gi::ray_type(m, entity->material.diffuse->getAlpha(uv), gi::to_lane_ray(ray), gi::to_v3(norm), lane_rng());
    }
    void secondaryRay(const Ray& ray, const Entity* current, gi::dvec3& norm, gi::dvec2& UV, double sx, double sy, gi::dvec3& refDir, gi::dvec3& f, double& roughness,
                      gi::dvec3& contrib, double& offset)
    {
        const gi::Mat m = mat_of(current);
        gi::V3 n = gi::to_v3(norm), rd, ff, cb = gi::to_v3(contrib);
        gi::secondary_ray(gi::to_lane_ray(ray), m, gi::to_v3(current->material.diffuse->get(UV)), current->material.diffuse->getAlpha(UV), n, sx, sy, rd, ff, roughness, cb, offset, lane_rng());
        norm = gi::from_v3(n); refDir = gi::from_v3(rd); f = gi::from_v3(ff); contrib = gi::from_v3(cb);
    }
    bool raymarch(const Ray& r, gi::dvec3& hit, gi::dvec3& col, double mint, double maxt)
    {
        if (!_scene) return false;
        std::vector<gi::FogD> fogs;
        std::vector<double> grid;
        for (AtmosphereEntity* a : _scene->at)
            if (HeightFog* hf = dynamic_cast<HeightFog*>(a)) {
                gi::FogD fd;
                memset(&fd, 0, sizeof fd);
                gi::put3(fd.pos, hf->pos); gi::put3(fd.size, hf->s); gi::put3(fd.col, hf->col);
                fd.d = hf->d; fd.sc = hf->sc;
                gi::put3(fd.bmin, hf->bbox.min); gi::put3(fd.bmax, hf->bbox.max);
                fd.grid_off = (int32_t)grid.size(); fd.grid_n = (int32_t)hf->noiseGrid.size();
                grid.insert(grid.end(), hf->noiseGrid.begin(), hf->noiseGrid.end());
                fogs.push_back(fd);
            }
        gi::Scene S{};
        S.fogs = fogs.data(); S.fog_grid = grid.data(); S.n_fog = (int32_t)fogs.size();
        gi::V3 h, c;
        if (!gi::raymarch(S, gi::to_lane_ray(r), h, c, mint, maxt, lane_rng(), gi::P_FOG_CAMERA)) return false;
        hit = gi::from_v3(h); col = gi::from_v3(c);
        return true;
    }

    bool running() const { return _running; }
    void stop() { _running = false; _cancel = 1; }
    void start() { _running = true; _cancel = 0; }

    int photons = 75000;
    int photon_depth = 5;
    int min_samples = 8;
    int max_samples = 32;
    double noise_thresh = 0.0015;
    gi::dvec3 ambient = gi::dvec3(0, 0, 0);
    uint64_t seed = 0x9E3779B97F4A7C15ull;   // counter-RNG seed (the reference seeds drand() with time(0))
    int progressive_rows = 16;                // rows of the first displayed stripe of run(); 0 = the whole frame in one call
    double progressive_ms = 50;               // stripes double while a step takes less than this many ms; 0 = every stripe progressive_rows high
    volatile int rows_in_flight = 0;          // rows of the step being rendered (they are painted when it ends)
    std::vector<int32_t> devices;             // HIP device ordinals to render on; empty = every visible GPU (the frame's stripes are dealt round-robin)
    volatile int rows_done = 0;               // rows of the current frame already in the Image
    int photons_on_device = 0;                // photons stored by the last on-device emission of run()

    std::shared_ptr<Image> getImage() const { return _image; }
    const std::vector<float>& linear() const { return _linear; }   // float tap of the frame (pre-gamma)
    const std::string& last_error() const { return _st->err; }
    PhotonMap* photonMap() const { return _st->photon_map.get(); }
    Camera _camera;

  private:
    struct State {
        gi_ctx* ctx = nullptr;        // the context of the per-ray methods and of photon emission; member 0 of the group when there is one
        gi_group* group = nullptr;    // every visible GPU, when there is more than one (or `devices` says so)
        std::shared_ptr<PhotonMap> photon_map;
        bool uploaded = false, photons_uploaded = false, fresh_map = true;
        std::string err;
        ~State() { if (group) gi_group_destroy(group); else if (ctx) gi_destroy(ctx); }
    };
    bool ensure_context()
    {
        if (_st->ctx) return true;
        const int want = devices.empty() ? gi_device_count() : (int)devices.size();
        int rc;
        if (want > 1) {
            rc = gi_group_create(&_st->group, (int32_t)devices.size(), devices.empty() ? nullptr : devices.data());
            if (rc == 0) _st->ctx = gi_group_ctx(_st->group, 0);
        } else
            rc = gi_create(&_st->ctx, devices.empty() ? 0 : devices[0]);
        if (rc != 0) { _st->err = "gi_create failed: no usable HIP device (this renderer has no CPU fallback)"; fprintf(stderr, "%s\n", _st->err.c_str()); return false; }
        return true;
    }
    bool ready()
    {
        if (!ensure_context() || !_scene) return false;
        if (!_scene->valid) { _scene->rebuild(); _st->uploaded = false; }
        return upload_scene();
    }
    // scene tables and photon map on the device; photons are emitted once per scene (the reference keeps a valid map across frames)
    bool prepare()
    {
        if (!ready()) return false;
        return photons_ready();
    }
    bool photons_ready()
    {
        PhotonMap* pm = _st->photon_map.get();
        if (!pm) return true;
        if (!pm->valid) {
            if (pm->size() == 0) {
                // nothing pushed by a caller: RayTracer::run's tracePhotons + rebuild (include/raytracer.h:61-72) entirely on the device(s) --
                // emission, octree, candidate ranges; the photons never visit the host (every device emits the same keyed set)
                const double box[6] = {pm->_root._bbox.min.x, pm->_root._bbox.min.y, pm->_root._bbox.min.z, pm->_root._bbox.max.x, pm->_root._bbox.max.y, pm->_root._bbox.max.z};
                const int members = _st->group ? gi_group_size(_st->group) : 1;
                for (int i = 0; i < members; i++) {
                    gi_ctx* cx = _st->group ? gi_group_ctx(_st->group, i) : _st->ctx;
                    const int n = gi_trace_photons(cx, _scene->lights.empty() ? 0 : photons, 5, seed, box, nullptr);
                    if (n < 0) { _st->err = gi_last_error(cx); fprintf(stderr, "gi: %s\n", _st->err.c_str()); return false; }
                    photons_on_device = n;
                }
                pm->valid = true;
                _st->photons_uploaded = true;
                return true;
            }
            pm->rebuild();
            _st->photons_uploaded = false;
        }
        if (!_st->photons_uploaded) {
            gi_photon_map_desc pd;
            gih_get_photon_desc(pm->handle(), &pd);
            if (check(_st->group ? gi_group_upload_photons(_st->group, &pd) : gi_upload_photons(_st->ctx, &pd)) != 0) return false;
            _st->photons_uploaded = true;
        }
        return true;
    }
    bool upload_scene()
    {
        if (_st->uploaded) return true;
        const double amb[3] = {ambient.x, ambient.y, ambient.z};
        gih_set_ambient(_scene->handle(), amb);
        gi_scene_desc d;
        if (gih_get_scene_desc(_scene->handle(), &d) != 0) { _st->err = "scene octree not built"; return false; }
        if (check(_st->group ? gi_group_upload_scene(_st->group, &d) : gi_upload_scene(_st->ctx, &d)) != 0) return false;
        _st->uploaded = true;
        if (_st->fresh_map) { if (_st->group) gi_group_clear_photons(_st->group); else gi_clear_photons(_st->ctx); _st->fresh_map = false; }   // setScene: a fresh PhotonMap; an edited scene keeps its map
        return true;
    }
    int check(int rc)
    {
        if (rc < 0 && rc != GI_E_CANCELLED) {
            _st->err = gi_last_error(_st->ctx);
            if (_st->err.empty() && _st->group) _st->err = gi_group_last_error(_st->group);
            fprintf(stderr, "gi: %s\n", _st->err.c_str());
        }
        return rc < 0 ? rc : 0;
    }
    gi_render_params params(int w, int h) const
    {
        gi_render_params rp;
        const gi::dvec3 *v[3] = {&_camera.pos, &_camera.up, &_camera.forward};
        double* dst[3] = {rp.cam_pos, rp.cam_up, rp.cam_forward};
        for (int k = 0; k < 3; k++) { dst[k][0] = v[k]->x; dst[k][1] = v[k]->y; dst[k][2] = v[k]->z; }
        rp.sensor_diag = _camera.sensorDiag; rp.focal_dist = _camera.focalDist;
        rp.width = w; rp.height = h;
        rp.stripe_h = h; rp.stripe_rank = 0; rp.stripe_world = 1;
        rp.min_samples = min_samples; rp.max_samples = max_samples; rp.noise_thresh = noise_thresh;
        rp.seed = seed;
        return rp;
    }
    static gi::Mat mat_of(const Entity* e)
    {
        gi::Mat m;
        memset(&m, 0, sizeof m);
        m.roughness = e->material.roughness; m.opacity = e->material.opacity; m.ior = e->material.IOR;
        m.dtex = -1; m.etex = -1;
        return m;
    }
    gi::Rng lane_rng() const { gi::Rng r = gi::rng_make(seed, rng_stream); r.depth = rng_depth; return r; }

    volatile bool _running = false;
    volatile int _cancel = 0;
    Octree* _scene = nullptr;
    std::shared_ptr<State> _st;
    std::shared_ptr<Image> _image;
    std::vector<float> _linear;
};
