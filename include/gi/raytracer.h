// include/gi/raytracer.h -- drop-in for the reference's RayTracer (include/raytracer.h:23-735) on the MI355X path.
// Same public data (photons, photon_depth, min_samples, max_samples, noise_thresh, ambient, _camera, width, height, p), same
// life cycle (RayTracer(camera); setScene(octree); start(); run(w, h) on a worker thread; getImage() polled by the GUI;
// stop()), copyable with shared scene / image exactly as gui.h:19 / viewer.h:16 need.  run() flattens + uploads when the
// scene is not valid, emits photons on the GPU when the photon map is not valid, renders through gi_render_host and stores
// gamma-2.2 / clamped / truncated 8-bit pixels like Image::setPixel.  Errors: the reference has no error channel (void
// returns, stdout); here last_error() carries the C ABI's message and run() leaves the image black.
#pragma once
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>
#include "../gi_hip.h"
#include "camera.h"
#include "image.h"
#include "octree.h"
#include "photonMap.h"

class RayTracer {
  public:
    int p = 0;
    int width = 0, height = 0;

    RayTracer() = delete;
    RayTracer(const Camera& camera) : _camera(camera), _st(std::make_shared<State>()), _image(std::make_shared<Image>(0, 0)) {}

    void setScene(Octree* scene)
    {
        _scene = scene;
        _st->photon_map = std::make_shared<PhotonMap>();
        _st->uploaded = false;
    }

    // RayTracer::run(w, h), include/raytracer.h:41-165
    void run(int w, int h)
    {
        _image = std::make_shared<Image>(w, h);
        width = w; height = h;
        if (!ensure_context() || !_scene) return;
        if (!_scene->valid) { _scene->rebuild(); _st->uploaded = false; }
        if (!upload_scene()) return;
        if (!_st->photon_map->valid) {
            tracePhotons(5, photons);
            _st->photon_map->rebuild(_scene);
            gi_photon_map_desc pd;
            gih_get_photon_desc(_scene->handle(), &pd);
            if (check(gi_upload_photons(_st->ctx, &pd)) != 0) return;
        }
        gi_render_params rp = params(w, h);
        std::vector<float> lin((size_t)w * h * 3);
        _cancel = _running ? 0 : 1;
        if (check(gi_render_host(_st->ctx, &rp, lin.data(), 0, nullptr, &_cancel)) != 0) return;
        _linear = lin;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const float* c = &lin[((size_t)y * w + x) * 3];
                // gamma(color, 2.2), glm::clamp(color, 0, 1), Image::setPixel (include/raytracer.h:150-157)
                gi::dvec3 g(std::pow((double)c[0], 1.0 / 2.2), std::pow((double)c[1], 1.0 / 2.2), std::pow((double)c[2], 1.0 / 2.2));
                g.x = std::fmin(std::fmax(g.x, 0.0), 1.0); g.y = std::fmin(std::fmax(g.y, 0.0), 1.0); g.z = std::fmin(std::fmax(g.z, 0.0), 1.0);
                _image->setPixel(x, y, g);
            }
    }

    // RayTracer::trace (include/raytracer.h:382-478) for a batch of rays [n][6] (origin, unit direction)
    bool trace(int n, const double* rays, int32_t* hit, int32_t* ent, double* res8) { return ready() && check(gi_trace(_st->ctx, n, rays, hit, ent, res8)) == 0; }
    // RayTracer::visible (include/raytracer.h:280-319): q [n][6] = shadow-ray origin, target
    bool visible(int n, const double* q, int32_t* vis) { return ready() && check(gi_visible(_st->ctx, n, q, vis)) == 0; }
    // RayTracer::samplePhotons(pos, dir, 32) (include/raytracer.h:532-579): q [n][6] = pos, dir
    bool samplePhotons(int n, const double* q, double* res3) { return ready() && check(gi_gather(_st->ctx, n, q, res3, nullptr)) == 0; }
    // RayTracer::tracePhotons (include/raytracer.h:582-715): emission on the GPU, photons appended to the map
    void tracePhotons(int maxDepth, int count)
    {
        if (!ready() || count <= 0 || _scene->lights.empty()) return;
        std::vector<double> out((size_t)count * _scene->lights.size() * 9);
        int64_t tries = 0;
        const int n = gi_emit_photons(_st->ctx, count, maxDepth, seed, out.data(), (int32_t)(out.size() / 9), &tries);
        if (n < 0) { check(n); return; }
        _st->photon_map->push_back_flat(out.data(), n);
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

    std::shared_ptr<Image> getImage() const { return _image; }
    const std::vector<float>& linear() const { return _linear; }   // float tap of the frame (pre-gamma)
    const std::string& last_error() const { return _st->err; }
    Camera _camera;

  private:
    struct State {
        gi_ctx* ctx = nullptr;
        std::shared_ptr<PhotonMap> photon_map;
        bool uploaded = false;
        std::string err;
        ~State() { if (ctx) gi_destroy(ctx); }
    };
    bool ensure_context()
    {
        if (_st->ctx) return true;
        const int rc = gi_create(&_st->ctx, 0);
        if (rc != 0) { _st->err = "gi_create failed: no usable HIP device (this renderer has no CPU fallback)"; fprintf(stderr, "%s\n", _st->err.c_str()); return false; }
        return true;
    }
    bool ready() { return ensure_context() && _scene && (_scene->valid || (_scene->rebuild(), true)) && upload_scene(); }
    bool upload_scene()
    {
        if (_st->uploaded) return true;
        const double amb[3] = {ambient.x, ambient.y, ambient.z};
        gih_set_ambient(_scene->handle(), amb);
        gi_scene_desc d;
        if (gih_get_scene_desc(_scene->handle(), &d) != 0) { _st->err = "scene octree not built"; return false; }
        if (check(gi_upload_scene(_st->ctx, &d)) != 0) return false;
        _st->uploaded = true;
        _st->photon_map->valid = false;
        return true;
    }
    int check(int rc)
    {
        if (rc < 0) { _st->err = gi_last_error(_st->ctx); fprintf(stderr, "gi: %s\n", _st->err.c_str()); }
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

    bool _running = false;
    volatile int _cancel = 0;
    Octree* _scene = nullptr;
    std::shared_ptr<State> _st;
    std::shared_ptr<Image> _image;
    std::vector<float> _linear;
};
