// oracle/ref/ref_driver.cpp -- fixture generator that drives the UNMODIFIED reference.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit #includes the reference headers *by path*
// from /root/reference (nothing of the reference is copied into this repository) and is linked
// with the reference's own five .cpp files (see oracle/ref/Makefile).  It exists only in the
// build container (the GPU box has no /root/reference); its outputs are the small .npy fixtures
// committed under tests/golden/, which pin oracle/gi_oracle.cpp.
//
// Determinism: std::time is wrapped at link time (-Wl,--wrap=time) to a constant and OpenMP is
// forced to one thread, so the reference's thread_local xorshift64* stream (include/util.h:52-80)
// is one reproducible chain.  SURVEY.md 8(c).
//
// Sub-commands:
//   halton OUT                         Halton_enum / Halton_sampler known answers
//   kat OUT                            util.{h,cpp} sampler / pow / refr / triBoxOverlap known answers
//   scene SCN OUT W H NPHOTONS         flattened scene tables, octree dump, trace / visible tables,
//                                      photon set + photon octree dump + samplePhotons table
//   chain SCN OUT W H SPP NPHOTONS MODE   whole-frame render on the pinned RNG chain
//                                      MODE=run: the reference's own RayTracer::run -> 8-bit image
//                                      MODE=lin: this driver's pixel loop calling radiance() -> f64 linear radiance
#include <ctime>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <map>

static const time_t PINNED_TIME = 1500000000;
extern "C" time_t __wrap_time(time_t* t) { if (t) *t = PINNED_TIME; return PINNED_TIME; }

#include <memory>
#include <algorithm>
#include <set>
#include <chrono>
#include <iostream>
#include <random>
#include <array>

#define private public
#include "camera.h"
#include "image.h"
#include "raytracer.h"
#undef private
#include "meshLoader.h"
#include "sceneLoader.h"

// ---------------------------------------------------------------- .npy writer (format 1.0)
static void save_npy(const std::string& path, const char* descr, const std::vector<size_t>& shape,
                     const void* data, size_t elem)
{
    std::string sh = "(";
    size_t n = 1;
    for (size_t i = 0; i < shape.size(); i++) { sh += std::to_string(shape[i]); sh += ","; n *= shape[i]; }
    sh += ")";
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + sh + ", }";
    size_t total = 10 + hdr.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    hdr += std::string(pad, ' ');
    hdr += "\n";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    unsigned char magic[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hdr.size() & 255), (unsigned char)(hdr.size() >> 8)};
    fwrite(magic, 1, 10, f);
    fwrite(hdr.data(), 1, hdr.size(), f);
    if (n) fwrite(data, elem, n, f);
    fclose(f);
}
static void save_f64(const std::string& p, const std::vector<double>& v, std::vector<size_t> shape) { save_npy(p, "<f8", shape, v.data(), 8); }
static void save_f32(const std::string& p, const std::vector<float>& v, std::vector<size_t> shape) { save_npy(p, "<f4", shape, v.data(), 4); }
static void save_i32(const std::string& p, const std::vector<int32_t>& v, std::vector<size_t> shape) { save_npy(p, "<i4", shape, v.data(), 4); }
static void save_u32(const std::string& p, const std::vector<uint32_t>& v, std::vector<size_t> shape) { save_npy(p, "<u4", shape, v.data(), 4); }
static void save_u8(const std::string& p, const std::vector<uint8_t>& v, std::vector<size_t> shape) { save_npy(p, "|u1", shape, v.data(), 1); }

static void push3(std::vector<double>& v, const glm::dvec3& a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }

// small private LCG for harness inputs (not the reference's RNG)
struct Lcg {
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed) {}
    double next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }
};

// ---------------------------------------------------------------- halton
static int cmd_halton(const std::string& out)
{
    Halton_sampler sampler;
    sampler.init_faure();

    const unsigned sizes[][2] = {{256, 256}, {512, 512}, {1920, 1080}, {3840, 2160}, {1000, 973}, {64, 48}, {1, 1}, {96, 54}};
    std::vector<uint32_t> en;   // per size: w,h,p2,p3,m_x,m_y,increment
    std::vector<uint32_t> gi;   // rows: size_id, s, x, y, index
    std::vector<float> sc;      // rows: size_id in, scale_x(in), scale_y(in)
    Lcg lcg(12345);
    for (unsigned k = 0; k < sizeof(sizes) / sizeof(sizes[0]); k++) {
        unsigned w = sizes[k][0], h = sizes[k][1];
        Halton_enum e(w, h);
        en.insert(en.end(), {w, h, e.m_p2, e.m_p3, e.m_x, e.m_y, e.m_increment});
        for (int t = 0; t < 96; t++) {
            unsigned x = (unsigned)(lcg.next() * w), y = (unsigned)(lcg.next() * h);
            if (t == 0) { x = 0; y = 0; }
            if (t == 1) { x = w - 1; y = h - 1; }
            unsigned smax = e.get_max_samples_per_pixel();
            unsigned s = (t < 4) ? (unsigned)t : (unsigned)(lcg.next() * std::min(smax, 1024u));
            if (t == 5) s = smax - 1;
            if (t == 6) s = smax;            // past the last sample a pixel can address: the 32-bit index wraps (include/halton_enum.h:109-113)
            if (t == 7) s = smax + 544;      // 4K: sample 1023 of BASELINE config 5
            gi.insert(gi.end(), {k, s, x, y, e.get_index(s, x, y)});
        }
        for (int t = 0; t < 8; t++) {
            float in = (float)lcg.next();
            sc.push_back((float)k); sc.push_back(in); sc.push_back(e.scale_x(in)); sc.push_back(e.scale_y(in));
        }
    }
    save_u32(out + "/halton_enum_params.npy", en, {en.size() / 7, 7});
    save_u32(out + "/halton_enum_index.npy", gi, {gi.size() / 5, 5});
    save_f32(out + "/halton_enum_scale.npy", sc, {sc.size() / 4, 4});

    // sample(dim, idx) for all 256 dims x 80 indices
    std::vector<uint32_t> idxs = {0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 31, 100, 242, 243, 244, 1023, 1024, 59048, 59049, 65535, 65536,
                                  1000003, 4478975, 4478976, 16777215, 16777216, 16777217, 123456789, 1146617855u, 2147483647u,
                                  2147483648u, 3486784400u, 3486784401u, 4294967295u};
    while (idxs.size() < 80) idxs.push_back((uint32_t)(lcg.next() * 4294967296.0));
    std::vector<float> sm;
    for (unsigned d = 0; d < 256; d++)
        for (size_t j = 0; j < idxs.size(); j++) sm.push_back(sampler.sample(d, idxs[j]));
    save_u32(out + "/halton_sample_idx.npy", idxs, {idxs.size()});
    save_f32(out + "/halton_sample.npy", sm, {256, idxs.size()});
    return 0;
}

// ---------------------------------------------------------------- util KATs
static int cmd_kat(const std::string& out)
{
    Lcg lcg(777);
    // fastPow / fastPrecisePow / gamma
    std::vector<double> pw;  // a, b, fastPow, fastPrecisePow(double b), std::pow
    for (int t = 0; t < 256; t++) {
        double a = lcg.next();
        double b = (t % 4 == 0) ? 0.5 : (t % 4 == 1) ? (1.0 / (1.0 / (0.05 + lcg.next()) + 1.0)) : (t % 4 == 2) ? 7.0 : 2.0 + lcg.next() * 3;
        if (t == 0) a = 1.0;
        if (t == 1) a = 0.64;
        if (t == 2) a = 1e-12;
        if (t == 3) a = 0.0;
        pw.insert(pw.end(), {a, b, fastPow(a, b), fastPrecisePow(a, b), std::pow(a, b)});
    }
    save_f64(out + "/kat_pow.npy", pw, {pw.size() / 5, 5});

    // samplers
    std::vector<double> hs;  // n(3) u v power | hemisphereSample_cos(n,u,v,power)(3) | hemisphereSample_cos(u,v,power)(3)
    std::vector<double> ph;  // outdir(3) n(3) power sx sy | sample_phong (3)
    std::vector<double> sc;  // n(3) u v power frac | sphereCapSample_cos(3)
    std::vector<double> ru;  // x y | randomUnitVec(3)
    std::vector<double> rf;  // inc(3) n(3) eta | refr(3) | glm::reflect(3)
    for (int t = 0; t < 200; t++) {
        glm::dvec3 n = glm::normalize(glm::dvec3(lcg.next() * 2 - 1, lcg.next() * 2 - 1, lcg.next() * 2 - 1));
        if (t == 0) n = glm::dvec3(0, 0, 1);
        if (t == 1) n = glm::dvec3(0, 0, -1);
        if (t == 2) n = glm::dvec3(1, 0, 0);
        if (t == 3) n = glm::dvec3(0, -1, 0);
        if (t == 4) n = glm::normalize(glm::dvec3(.3, .5, -.81));
        float u = (float)lcg.next(), v = (float)lcg.next();
        if (t == 4) { u = .37f; v = .61f; }
        if (t == 5) { u = 0.f; v = 0.f; }
        if (t == 6) { u = 0.99999994f; v = 0.99999994f; }
        double power = (t % 3 == 0) ? 2.0 : (t % 3 == 1) ? 1.0 : (1.0 / (0.05 + lcg.next()) + 1);
        glm::dvec3 r = hemisphereSample_cos(n, u, v, power);
        glm::dvec3 r2 = hemisphereSample_cos(u, v, power);
        hs.insert(hs.end(), {n.x, n.y, n.z, (double)u, (double)v, power, r.x, r.y, r.z, r2.x, r2.y, r2.z});

        glm::dvec3 od = glm::normalize(glm::dvec3(lcg.next() * 2 - 1, lcg.next() * 2 - 1, lcg.next() * 2 - 1));
        double sx = (double)(float)lcg.next(), sy = (double)(float)lcg.next();
        glm::dvec3 p = sample_phong(od, n, power, sx, sy);
        ph.insert(ph.end(), {od.x, od.y, od.z, n.x, n.y, n.z, power, sx, sy, p.x, p.y, p.z});

        double frac = lcg.next();
        glm::dvec3 c = sphereCapSample_cos(n, u, v, (t % 2) ? 1.0 : 2.0, frac);
        sc.insert(sc.end(), {n.x, n.y, n.z, (double)u, (double)v, (t % 2) ? 1.0 : 2.0, frac, c.x, c.y, c.z});

        double x = lcg.next(), y = lcg.next();
        if (t == 0) { x = 0; y = 0; }
        if (t == 1) { x = 1; y = 1; }
        glm::dvec3 q = randomUnitVec(x, y);
        ru.insert(ru.end(), {x, y, q.x, q.y, q.z});

        glm::dvec3 inc = glm::normalize(glm::dvec3(lcg.next() * 2 - 1, lcg.next() * 2 - 1, lcg.next() * 2 - 1));
        glm::dvec3 nn = n;
        if (glm::dot(nn, inc) > 0) nn *= -1.0;
        double eta = (t % 2) ? 1.53 : 1.0 / 1.53;
        if (t % 7 == 0) eta = 0.575;
        glm::dvec3 rr = refr(inc, nn, eta);
        glm::dvec3 rl = glm::reflect(inc, nn);
        rf.insert(rf.end(), {inc.x, inc.y, inc.z, nn.x, nn.y, nn.z, eta, rr.x, rr.y, rr.z, rl.x, rl.y, rl.z});
    }
    save_f64(out + "/kat_hemi.npy", hs, {hs.size() / 12, 12});
    save_f64(out + "/kat_phong.npy", ph, {ph.size() / 12, 12});
    save_f64(out + "/kat_cap.npy", sc, {sc.size() / 10, 10});
    save_f64(out + "/kat_unitvec.npy", ru, {ru.size() / 5, 5});
    save_f64(out + "/kat_refr.npy", rf, {rf.size() / 13, 13});

    // triBoxOverlap: center(3) half(3) verts(9) | result
    std::vector<double> tb;
    for (int t = 0; t < 2000; t++) {
        glm::dvec3 c(lcg.next() * 2 - 1, lcg.next() * 2 - 1, lcg.next() * 2 - 1);
        glm::dvec3 hsz(lcg.next() * .5 + .01, lcg.next() * .5 + .01, lcg.next() * .5 + .01);
        glm::dvec3 tv[3];
        double spread = (t % 3 == 0) ? 0.2 : 1.5;
        glm::dvec3 base(lcg.next() * 2 - 1, lcg.next() * 2 - 1, lcg.next() * 2 - 1);
        for (int k = 0; k < 3; k++) tv[k] = base + spread * glm::dvec3(lcg.next() - .5, lcg.next() - .5, lcg.next() - .5);
        if (t % 10 == 0) { tv[1].y = tv[0].y; tv[2].y = tv[0].y; }  // axis aligned triangles
        glm::dvec3 cp[3] = {tv[0], tv[1], tv[2]};
        bool r = triBoxOverlap(c, hsz, cp);
        tb.insert(tb.end(), {c.x, c.y, c.z, hsz.x, hsz.y, hsz.z, tv[0].x, tv[0].y, tv[0].z, tv[1].x, tv[1].y, tv[1].z, tv[2].x, tv[2].y, tv[2].z, r ? 1.0 : 0.0});
    }
    save_f64(out + "/kat_tribox.npy", tb, {tb.size() / 16, 16});

    // the 8-bit sink (include/raytracer.h:150-157, include/image.h:14-16): gamma(color, 2.2), glm::clamp(color, 0, 1), Image::setPixel, read back.
    // Rows: linear colour (3) | the bytes QImage holds afterwards (3).  Includes negative channels (pow of a negative base is NaN, which
    // survives glm::clamp and makes the QColor invalid), values above 1, exact 0 and 1, and values around the k/255 thresholds.
    {
        std::vector<double> px;
        std::vector<glm::dvec3> in;
        in.push_back(glm::dvec3(0, 0, 0)); in.push_back(glm::dvec3(1, 1, 1)); in.push_back(glm::dvec3(0.5, 0.5, 0.5));
        in.push_back(glm::dvec3(-0.25, 0.3, 0.6)); in.push_back(glm::dvec3(0.3, -1e-9, 0.6)); in.push_back(glm::dvec3(0.3, 0.6, -3.0));
        in.push_back(glm::dvec3(-0.1, -0.2, -0.3)); in.push_back(glm::dvec3(2.5, 0.2, 1.0000001)); in.push_back(glm::dvec3(1e-12, 1e-300, 0.999999999));
        in.push_back(glm::dvec3(-0.0, 0.25, 0.75));
        for (int k = 1; k < 256; k += 7) { const double t = std::pow(k / 255.0, 2.2); in.push_back(glm::dvec3(t, std::nextafter(t, 0.0), std::nextafter(t, 1.0))); }
        for (int t = 0; t < 160; t++) in.push_back(glm::dvec3(lcg.next() * 1.2 - 0.1, lcg.next() * lcg.next(), lcg.next() * 3 - 0.5));
        Image img((int)in.size(), 1);
        for (size_t i = 0; i < in.size(); i++) {
            glm::dvec3 color = gamma(in[i], 2.2);
            img.setPixel((int)i, 0, glm::clamp(color, 0.0, 1.0));
        }
        for (size_t i = 0; i < in.size(); i++) {
            const glm::dvec3 q = img.getPixel((int)i, 0);
            px.insert(px.end(), {in[i].x, in[i].y, in[i].z, std::round(q.x * 255.), std::round(q.y * 255.), std::round(q.z * 255.)});
        }
        save_f64(out + "/kat_pixel.npy", px, {px.size() / 6, 6});
    }

    // xorshift64* stream from the pinned seed (include/util.h:52-80): first 64 drand() values of this thread
    std::vector<double> dr;
    for (int i = 0; i < 64; i++) dr.push_back(drand());
    save_f64(out + "/kat_drand.npy", dr, {dr.size()});
    return 0;
}

// ---------------------------------------------------------------- scene helpers
struct Loaded {
    Camera camera;
    RayTracer rt;
    Octree* scene;
    std::vector<Entity*> ents;            // root entity list in insertion order (captured before rebuild clears it)
    std::map<const Entity*, int> ent_id;
    Loaded() : camera({10, 5, 0}, {0, 0, 0}), rt(camera), scene(new Octree()) {}
};

static void load(Loaded& L, const char* scn)
{
    loadScene(L.scene, L.rt, scn);
    L.rt.setScene(L.scene);
    L.ents = L.scene->_root._entities;
    for (size_t i = 0; i < L.ents.size(); i++) L.ent_id[L.ents[i]] = (int)i;
}

static void dump_scene_tables(Loaded& L, const std::string& out)
{
    size_t T = L.ents.size();
    std::vector<double> pos, nrm, uv, fn, mat, ebox;
    std::vector<int32_t> kind;
    for (Entity* e : L.ents) {
        triangle* t = dynamic_cast<triangle*>(e);
        sphere* s = dynamic_cast<sphere*>(e);
        kind.push_back(t ? 0 : (s ? 1 : 2));
        if (t) {
            for (int k = 0; k < 3; k++) { push3(pos, t->vertices[k].pos); push3(nrm, t->vertices[k].norm); uv.push_back(t->vertices[k].texCoord.x); uv.push_back(t->vertices[k].texCoord.y); }
            push3(fn, t->norm);
        } else {
            // sphere: centre in vertex 0, radius in vertex 1 x
            push3(pos, e->pos); pos.push_back(s ? s->rad : 0); pos.push_back(0); pos.push_back(0); pos.push_back(0); pos.push_back(0); pos.push_back(0);
            for (int k = 0; k < 9; k++) nrm.push_back(0);
            for (int k = 0; k < 6; k++) uv.push_back(0);
            push3(fn, glm::dvec3(0, 0, 0));
        }
        glm::dvec2 z(0, 0);
        glm::dvec3 d = e->material.diffuse->get(z), em = e->material.emissive->get(z);
        mat.insert(mat.end(), {e->material.roughness, e->material.opacity, e->material.IOR, d.x, d.y, d.z, em.x, em.y, em.z});
        BoundingBox b = e->boundingBox();
        push3(ebox, b.min); push3(ebox, b.max);
    }
    save_f64(out + "/tri_pos.npy", pos, {T, 3, 3});
    save_f64(out + "/tri_nrm.npy", nrm, {T, 3, 3});
    save_f64(out + "/tri_uv.npy", uv, {T, 3, 2});
    save_f64(out + "/tri_fnorm.npy", fn, {T, 3});
    save_f64(out + "/tri_mat.npy", mat, {T, 9});
    save_f64(out + "/tri_bbox.npy", ebox, {T, 6});
    save_i32(out + "/ent_kind.npy", kind, {T});

    {   // textures (include/material.h:10-81): the distinct texture objects the entities' materials point to, in first-seen order, in
        // the layout of gi_scene_desc::tex_* (include/gi_hip.h); image pixels as QImage::pixelColor returns them; and known answers of
        // get() / getAlpha() on a uv lattice that leaves [0, 1] on both sides
        std::vector<texture*> texs;
        std::map<texture*, int> tex_id;
        std::vector<int32_t> tri_tex, tkind;
        for (Entity* e : L.ents)
            for (texture* t : {e->material.diffuse, e->material.emissive}) {
                if (!tex_id.count(t)) { tex_id[t] = (int)texs.size(); texs.push_back(t); }
                tri_tex.push_back(tex_id[t]);
            }
        std::vector<double> tpar, kat;
        std::vector<uint8_t> pix;
        const int KU = 23;
        for (texture* t : texs) {
            checkerboard* cb = dynamic_cast<checkerboard*>(t);
            imageTexture* im = dynamic_cast<imageTexture*>(t);
            double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (cb) { q[0] = cb->a.x; q[1] = cb->a.y; q[2] = cb->a.z; q[3] = cb->b.x; q[4] = cb->b.y; q[5] = cb->b.z; q[6] = cb->tiles; tkind.push_back(1); }
            else if (im) {
                q[0] = im->tile.x; q[1] = im->tile.y; q[2] = im->image.width(); q[3] = im->image.height(); q[4] = im->image.hasAlphaChannel() ? 1 : 0; q[5] = (double)pix.size();
                for (int y = 0; y < im->image.height(); y++)
                    for (int x = 0; x < im->image.width(); x++) {
                        QColor c = im->image.pixelColor(x, y);
                        pix.push_back((uint8_t)c.red()); pix.push_back((uint8_t)c.green()); pix.push_back((uint8_t)c.blue()); pix.push_back((uint8_t)c.alpha());
                    }
                tkind.push_back(2);
            } else { q[0] = t->color.x; q[1] = t->color.y; q[2] = t->color.z; tkind.push_back(0); }
            tpar.insert(tpar.end(), q, q + 8);
            for (int a = 0; a < KU; a++)
                for (int b = 0; b < KU; b++) {
                    glm::dvec2 uv(-0.75 + 0.1171875 * a, -0.6 + 0.109375 * b);
                    glm::dvec3 g = t->get(uv);
                    kat.insert(kat.end(), {uv.x, uv.y, g.x, g.y, g.z, t->getAlpha(uv)});
                }
        }
        save_i32(out + "/tri_tex.npy", tri_tex, {T, 2});
        save_i32(out + "/tex_kind.npy", tkind, {texs.size()});
        save_f64(out + "/tex_param.npy", tpar, {texs.size(), 8});
        save_u8(out + "/tex_pixels.npy", pix, {pix.size()});
        save_f64(out + "/tex_kat.npy", kat, {texs.size(), (size_t)KU * KU, 6});
    }

    std::vector<double> li;
    for (Light* l : L.scene->lights) li.insert(li.end(), {l->pos.x, l->pos.y, l->pos.z, l->col.x, l->col.y, l->col.z, l->rad, l->dir.x, l->dir.y, l->dir.z, l->angle});
    save_f64(out + "/lights.npy", li, {L.scene->lights.size(), 11});

    {   // atmosphere entities (HeightFog): parameters + the noise grid its constructor filled with drand()
        std::vector<double> fp, fg;
        std::vector<int32_t> fo = {0};
        for (AtmosphereEntity* a : L.scene->at) {
            HeightFog* hf = dynamic_cast<HeightFog*>(a);
            if (!hf) continue;
            fp.insert(fp.end(), {hf->pos.x, hf->pos.y, hf->pos.z, hf->s.x, hf->s.y, hf->s.z, hf->col.x, hf->col.y, hf->col.z, hf->d, hf->sc, (double)hf->nscale});
            fg.insert(fg.end(), hf->noiseGrid.begin(), hf->noiseGrid.end());
            fo.push_back((int32_t)fg.size());
        }
        save_f64(out + "/fog.npy", fp, {fp.size() / 12, 12});
        save_f64(out + "/fog_grid.npy", fg, {fg.size()});
        save_i32(out + "/fog_grid_off.npy", fo, {fo.size()});
    }
    std::vector<double> misc = {L.rt.ambient.x, L.rt.ambient.y, L.rt.ambient.z, (double)L.rt.min_samples, (double)L.rt.max_samples, L.rt.noise_thresh,
                                (double)L.rt.photons, (double)L.rt.photon_depth,
                                L.rt._camera.pos.x, L.rt._camera.pos.y, L.rt._camera.pos.z,
                                L.rt._camera.up.x, L.rt._camera.up.y, L.rt._camera.up.z,
                                L.rt._camera.forward.x, L.rt._camera.forward.y, L.rt._camera.forward.z,
                                L.rt._camera.right.x, L.rt._camera.right.y, L.rt._camera.right.z,
                                L.rt._camera.sensorDiag, L.rt._camera.focalDist};
    save_f64(out + "/settings.npy", misc, {misc.size()});
}

static void walk_octree(const Octree::Node* n, Loaded& L, std::vector<double>& box, std::vector<int32_t>& child, std::vector<int32_t>& eoff, std::vector<int32_t>& eidx)
{
    int me = (int)(box.size() / 6);
    push3(box, n->_bbox.min); push3(box, n->_bbox.max);
    for (int i = 0; i < 8; i++) child.push_back(-1);
    for (Entity* e : n->_entities) eidx.push_back(L.ent_id.at(e));
    eoff.push_back((int)eidx.size());
    for (int i = 0; i < 8; i++)
        if (n->_children[i]) {
            child[(size_t)me * 8 + i] = (int)(box.size() / 6);
            walk_octree(n->_children[i].get(), L, box, child, eoff, eidx);
        }
}

static void dump_octree(Loaded& L, const std::string& out)
{
    std::vector<double> box;
    std::vector<int32_t> child, eoff = {0}, eidx;
    walk_octree(&L.scene->_root, L, box, child, eoff, eidx);
    save_f64(out + "/oct_bbox.npy", box, {box.size() / 6, 6});
    save_i32(out + "/oct_child.npy", child, {child.size() / 8, 8});
    save_i32(out + "/oct_ent_off.npy", eoff, {eoff.size()});
    save_i32(out + "/oct_ent_idx.npy", eidx, {eidx.size()});
}

static void walk_pmap(const PhotonMap::Node* n, std::map<const Photon*, int>& pid, std::vector<double>& box, std::vector<int32_t>& firstchild, std::vector<int32_t>& poff, std::vector<int32_t>& pidx)
{
    int me = (int)(box.size() / 6);
    push3(box, n->_bbox.min); push3(box, n->_bbox.max);
    firstchild.push_back(-1);
    for (Photon* p : n->_entities) pidx.push_back(pid.at(p));
    poff.push_back((int)pidx.size());
    if (!n->is_leaf()) {
        // children are dumped consecutively in pre-order; record each child's node index
        std::vector<int32_t> ch;
        for (int i = 0; i < 8; i++) {
            ch.push_back((int)(box.size() / 6));
            walk_pmap(n->_children[i].get(), pid, box, firstchild, poff, pidx);
        }
        firstchild[me] = ch[0];
        (void)ch;
    }
}

struct Cam {
    glm::dvec3 screenCenter, right, up, pos;
    double sw, sh;
    int w, h;
};
// camera set-up as in RayTracer::run (include/raytracer.h:74-78)
static Cam make_cam(const Camera& c, int w, int h)
{
    Cam m;
    m.sw = (c.sensorDiag * w) / (sqrt((double)w * w + h * h));
    m.sh = m.sw * ((double)h / w);
    m.screenCenter = c.pos + c.focalDist * c.forward;
    m.right = glm::normalize(glm::cross(c.forward, c.up));
    m.up = c.up;
    m.pos = c.pos;
    m.w = w; m.h = h;
    return m;
}
// primary ray as in RayTracer::run (include/raytracer.h:112-129), FOCAL_BLUR == 0
static Ray primary(const Cam& m, const Halton_sampler& sampler, const Halton_enum& he, int s, int x, int y, int& idx)
{
    idx = he.get_index(s, x, y);
    double xr = sampler.sample(0, idx);
    double yr = sampler.sample(1, idx);
    double dx = he.scale_x(xr);
    double dy = he.scale_y(yr);
    glm::dvec3 pixelPos = m.screenCenter + (m.sw * (dx / m.w - .5)) * m.right - (m.sh * (dy / m.h - .5)) * m.up;
    glm::dvec3 eyePos = m.pos + FOCAL_BLUR * (xr - .5) * m.right + FOCAL_BLUR * (yr - .5) * m.up;
    return Ray(eyePos, glm::normalize(pixelPos - eyePos));
}

static int cmd_scene(const char* scn, const std::string& out, int W, int H, int nphotons)
{
    omp_set_num_threads(1);
    Loaded L;
    load(L, scn);
    L.rt.photons = nphotons;
    L.scene->rebuild();
    dump_scene_tables(L, out);
    dump_octree(L, out);

    Halton_sampler sampler;
    sampler.init_faure();
    Halton_enum he(W, H);
    Cam cam = make_cam(L.rt._camera, W, H);

    // ---- rays: all primary rays (s = 0) then random rays through the scene box
    std::vector<double> rays;     // origin(3) dir(3)
    std::vector<int32_t> ridx;    // halton index for primaries, -1 otherwise
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int idx;
            Ray r = primary(cam, sampler, he, 0, x, y, idx);
            push3(rays, r.origin); push3(rays, r.dir); ridx.push_back(idx);
        }
    Lcg lcg(4242);
    glm::dvec3 bmin = L.scene->_root._bbox.min, bmax = L.scene->_root._bbox.max, ext = bmax - bmin;
    for (int t = 0; t < 6000; t++) {
        glm::dvec3 o = bmin + glm::dvec3(lcg.next() * 1.2 - .1, lcg.next() * 1.2 - .1, lcg.next() * 1.2 - .1) * ext;
        glm::dvec3 d(lcg.next() * 2 - 1, lcg.next() * 2 - 1, lcg.next() * 2 - 1);
        if (t % 50 == 0) d = glm::dvec3(0, -1, 0);           // axis parallel: invDir has +-inf components
        if (t % 50 == 1) d = glm::dvec3(1, 0, 0);
        if (t % 50 == 2) d = glm::dvec3(0, 0, -1);
        Ray r(o, d);
        push3(rays, r.origin); push3(rays, r.dir); ridx.push_back(-1);
    }
    size_t NR = ridx.size();
    std::vector<int32_t> thit(NR), tent(NR), tleaves(NR);
    std::vector<double> tres(NR * 8, 0.0);  // hit(3) norm(3) uv(2)
    for (size_t i = 0; i < NR; i++) {
        Ray r(glm::dvec3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), glm::dvec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
        // keep the exact stored direction (already normalised): re-normalising may move it by an ulp
        r.dir = glm::dvec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        r.invDir = glm::dvec3(1.0 / r.dir);
        glm::dvec3 hit(0), norm(0);
        glm::dvec2 uv(0);
        Entity* obj = nullptr;
        bool ok = L.rt.trace(r, hit, norm, uv, obj);
        thit[i] = ok;
        tent[i] = ok ? L.ent_id.at(obj) : -1;
        if (ok) { tres[i * 8 + 0] = hit.x; tres[i * 8 + 1] = hit.y; tres[i * 8 + 2] = hit.z; tres[i * 8 + 3] = norm.x; tres[i * 8 + 4] = norm.y; tres[i * 8 + 5] = norm.z; tres[i * 8 + 6] = uv.x; tres[i * 8 + 7] = uv.y; }
        tleaves[i] = (int)L.scene->intersectSorted(r, 0, INFINITY).size();
    }
    save_f64(out + "/rays.npy", rays, {NR, 6});
    save_i32(out + "/rays_halton_idx.npy", ridx, {NR});
    save_i32(out + "/trace_hit.npy", thit, {NR});
    save_i32(out + "/trace_ent.npy", tent, {NR});
    save_f64(out + "/trace_res.npy", tres, {NR, 8});
    save_i32(out + "/trace_nleaves.npy", tleaves, {NR});

    // sorted-leaf order for a subset: leaf node pre-order index + t0
    {
        // map Node* -> pre-order index
        std::map<const Octree::Node*, int> nid;
        std::vector<const Octree::Node*> st = {&L.scene->_root};
        // pre-order numbering identical to walk_octree
        struct W { static void go(const Octree::Node* n, std::map<const Octree::Node*, int>& m) { int id = (int)m.size(); m[n] = id; for (int i = 0; i < 8; i++) if (n->_children[i]) go(n->_children[i].get(), m); } };
        W::go(&L.scene->_root, nid);
        std::vector<int32_t> lo = {0}, ln;
        std::vector<double> lt;
        size_t step = std::max<size_t>(1, NR / 600);
        std::vector<int32_t> lray;
        for (size_t i = 0; i < NR; i += step) {
            Ray r(glm::dvec3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), glm::dvec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
            r.dir = glm::dvec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
            r.invDir = glm::dvec3(1.0 / r.dir);
            auto v = L.scene->intersectSorted(r, 0, INFINITY);
            for (auto& p : v) { ln.push_back(nid.at(p.first)); lt.push_back(p.second); }
            lo.push_back((int)ln.size());
            lray.push_back((int)i);
        }
        save_i32(out + "/leaforder_ray.npy", lray, {lray.size()});
        save_i32(out + "/leaforder_off.npy", lo, {lo.size()});
        save_i32(out + "/leaforder_node.npy", ln, {ln.size()});
        save_f64(out + "/leaforder_t0.npy", lt, {lt.size()});
    }

    // ---- shadow queries: from traced hit points (biased along normal) toward each light centre / random points
    {
        std::vector<double> sq;    // origin(3) target(3)
        std::vector<int32_t> vis, ncand;
        glm::dvec3 lpos = L.scene->lights.empty() ? glm::dvec3(0, 5, 0) : L.scene->lights[0]->pos;
        size_t step = std::max<size_t>(1, NR / 5000);
        for (size_t i = 0; i < NR; i += step) {
            if (!thit[i]) continue;
            glm::dvec3 hit(tres[i * 8], tres[i * 8 + 1], tres[i * 8 + 2]), n(tres[i * 8 + 3], tres[i * 8 + 4], tres[i * 8 + 5]);
            glm::dvec3 d(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
            if (glm::dot(n, d) > 0) n *= -1.0;
            glm::dvec3 o = hit + SHADOW_BIAS * n;
            glm::dvec3 target = (i % 3 == 0) ? bmin + glm::dvec3(lcg.next(), lcg.next(), lcg.next()) * ext : lpos + 0.05 * randomUnitVec(lcg.next(), lcg.next());
            glm::dvec3 ld = target - o;
            double maxt = vecLengthSquared(ld);
            Ray sr(o, ld);
            bool v = L.rt.visible(sr, maxt);
            push3(sq, o); push3(sq, target);
            vis.push_back(v);
            ncand.push_back((int)L.scene->intersect(sr, 0, sqrt(maxt) - SHADOW_BIAS).size());
        }
        save_f64(out + "/shadow_q.npy", sq, {vis.size(), 6});
        save_i32(out + "/shadow_vis.npy", vis, {vis.size()});
        save_i32(out + "/shadow_ncand.npy", ncand, {ncand.size()});
    }

    // ---- photons (pinned chain; this process has consumed drand() draws above, so the photon SET is a fixture input, not a chain pin)
    if (nphotons > 0 && !L.scene->lights.empty()) {
        L.rt.tracePhotons(5, nphotons, sampler, he);
        PhotonMap* pm = L.rt._photon_map;
        std::vector<Photon*> ph = pm->_root._entities;
        std::map<const Photon*, int> pid;
        std::vector<double> pv;
        for (size_t i = 0; i < ph.size(); i++) { pid[ph[i]] = (int)i; push3(pv, ph[i]->origin); push3(pv, ph[i]->dir); push3(pv, ph[i]->col); }
        save_f64(out + "/photons.npy", pv, {ph.size(), 9});
        pm->rebuild();
        std::vector<double> box;
        std::vector<int32_t> fc, poff = {0}, pidx;
        walk_pmap(&pm->_root, pid, box, fc, poff, pidx);
        save_f64(out + "/pm_bbox.npy", box, {box.size() / 6, 6});
        save_i32(out + "/pm_firstchild.npy", fc, {fc.size()});
        save_i32(out + "/pm_off.npy", poff, {poff.size()});
        save_i32(out + "/pm_idx.npy", pidx, {pidx.size()});

        // gather queries: primary hit points, photon positions (jittered), random points incl. outside the root box
        std::vector<double> gq;   // pos(3) dir(3)
        std::vector<double> gr;   // result(3)
        std::vector<int32_t> gn;  // candidate count
        auto query = [&](glm::dvec3 p, glm::dvec3 d) {
            double scale = 0;
            glm::dvec3 pp = p;
            int n = (int)pm->getInRange(pp, scale, 0).size();
            glm::dvec3 r = L.rt.samplePhotons(p, d, 32);
            push3(gq, p); push3(gq, d); push3(gr, r); gn.push_back(n);
        };
        size_t step = std::max<size_t>(1, NR / 4000);
        for (size_t i = 0; i < NR; i += step) {
            if (!thit[i]) continue;
            glm::dvec3 hit(tres[i * 8], tres[i * 8 + 1], tres[i * 8 + 2]);
            query(hit, randomUnitVec(lcg.next(), lcg.next()));
        }
        for (size_t i = 0; i < ph.size(); i += std::max<size_t>(1, ph.size() / 3000))
            query(ph[i]->origin + 0.01 * glm::dvec3(lcg.next() - .5, lcg.next() - .5, lcg.next() - .5), randomUnitVec(lcg.next(), lcg.next()));
        for (int t = 0; t < 500; t++)
            query(bmin + glm::dvec3(lcg.next() * 1.4 - .2, lcg.next() * 1.4 - .2, lcg.next() * 1.4 - .2) * ext, randomUnitVec(lcg.next(), lcg.next()));
        query(bmax, glm::dvec3(0, 1, 0));   // exactly on the half-open upper boundary
        query(bmin, glm::dvec3(0, 1, 0));
        save_f64(out + "/gather_q.npy", gq, {gn.size(), 6});
        save_f64(out + "/gather_res.npy", gr, {gn.size(), 3});
        save_i32(out + "/gather_ncand.npy", gn, {gn.size()});
    }
    return 0;
}

// ---------------------------------------------------------------- whole-frame chain renders
static int cmd_chain(const char* scn, const std::string& out, int W, int H, int spp, int nphotons, const std::string& mode)
{
    omp_set_num_threads(1);
    Loaded L;
    load(L, scn);
    L.rt.photons = nphotons;
    L.rt.min_samples = spp;
    L.rt.max_samples = spp;
    dump_scene_tables(L, out);   // before rebuild: light dir/angle not yet computed here (lights.npy of `scene` has them)
    L.rt.start();
    if (mode == "run") {
        L.rt.run(W, H);          // the reference's own frame loop, 8-bit result
        std::vector<uint8_t> im;
        auto img = L.rt.getImage();
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                glm::dvec3 p = img->getPixel(x, y);
                im.push_back((uint8_t)lround(p.x * 255)); im.push_back((uint8_t)lround(p.y * 255)); im.push_back((uint8_t)lround(p.z * 255));
            }
        save_u8(out + "/chain_run_u8.npy", im, {(size_t)H, (size_t)W, 3});
    } else {
        // same preparation order as RayTracer::run (include/raytracer.h:46-72) without the two subrand() draws,
        // then a row-major pixel loop that keeps the running mean of radiance() in double
        Halton_sampler sampler;
        sampler.init_faure();
        Halton_enum he(W, H);
        if (!L.scene->valid) L.scene->rebuild();
        if (!L.rt._photon_map->valid) { L.rt.tracePhotons(5, L.rt.photons, sampler, he); L.rt._photon_map->rebuild(); }
        Cam cam = make_cam(L.rt._camera, W, H);
        std::vector<double> lin;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                glm::dvec3 color(0.5, 0.5, 0.5);
                for (int s = 0; s < spp; s++) {
                    int idx;
                    Ray ray = primary(cam, sampler, he, s, x, y, idx);
                    glm::dvec3 Lr = L.rt.radiance(ray, 0, sampler, he, idx, glm::dvec3(1, 1, 1));
                    if (s == 0) color = Lr;
                    else color = (1.0 * s * color + Lr) * (1.0 / (s + 1));
                }
                push3(lin, color);
            }
        save_f64(out + "/chain_lin.npy", lin, {(size_t)H, (size_t)W, 3});
    }
    // photon set of this chain (root list is cleared by rebuild only when it partitions; dump what is reachable)
    {
        std::vector<double> pv;
        std::vector<const PhotonMap::Node*> st = {&L.rt._photon_map->_root};
        size_t n = 0;
        while (!st.empty()) {
            const PhotonMap::Node* nd = st.back(); st.pop_back();
            for (Photon* p : nd->_entities) { push3(pv, p->origin); push3(pv, p->dir); push3(pv, p->col); n++; }
            if (!nd->is_leaf()) for (int i = 7; i >= 0; i--) st.push_back(nd->_children[i].get());
        }
        save_f64(out + "/chain_photons_leaforder.npy", pv, {n, 9});
    }
    std::vector<double> meta = {(double)W, (double)H, (double)spp, (double)nphotons, (double)PINNED_TIME, mode == "run" ? 1.0 : 0.0};
    save_f64(out + "/chain_meta.npy", meta, {meta.size()});
    return 0;
}

// ---------------------------------------------------------------- timing of the reference's own pixel loop (calibration of the CPU baseline)
static int cmd_time(const char* scn, int W, int H, int spp, int nphotons, int threads)
{
    omp_set_num_threads(threads);
    Loaded L;
    load(L, scn);
    L.rt.photons = nphotons;
    L.rt.min_samples = spp;
    L.rt.max_samples = spp;
    Halton_sampler sampler;
    sampler.init_faure();
    Halton_enum he(W, H);
    L.scene->rebuild();
    auto t0 = std::chrono::high_resolution_clock::now();
    L.rt.tracePhotons(5, L.rt.photons, sampler, he);
    L.rt._photon_map->rebuild();
    auto t1 = std::chrono::high_resolution_clock::now();
    L.rt.start();
    L.rt.run(W, H);   // scene and photon map are valid: this is the pixel loop (plus Halton table init)
    auto t2 = std::chrono::high_resolution_clock::now();
    double tp = std::chrono::duration<double>(t1 - t0).count(), tr = std::chrono::duration<double>(t2 - t1).count();
    fprintf(stderr, "REFTIME threads %d frame %dx%dx%d photons %d: photon_s %.3f render_s %.3f Msamples/s %.4f\n", threads, W, H, spp, nphotons, tp, tr, (double)W * H * spp / tr / 1e6);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: see header comment\n"); return 2; }
    std::string cmd = argv[1];
    if (cmd == "halton") return cmd_halton(argv[2]);
    if (cmd == "kat") return cmd_kat(argv[2]);
    if (cmd == "scene" && argc >= 7) return cmd_scene(argv[2], argv[3], atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
    if (cmd == "time" && argc >= 8) return cmd_time(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]));
    if (cmd == "chain" && argc >= 9) return cmd_chain(argv[2], argv[3], atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), argv[8]);
    fprintf(stderr, "bad arguments\n");
    return 2;
}
