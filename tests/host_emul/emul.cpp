// tests/host_emul/emul.cpp -- CPU build of the per-lane device functions (gi_device.h) for unit tests and sanitizers.
//
// TEST INFRASTRUCTURE ONLY.  The product library never contains this file: it exists so that the kernel *logic*
// (stackless ordered traversal, LDS heap selection, iterative radiance, tile/stripe maps) can be checked against the
// oracle in the CPU-only container and under -fsanitize=address,undefined.  Each "lane" is run sequentially; the heap
// that lives in LDS on the GPU is a plain array here.
#include <cstdlib>
#define GI_HD static inline
#define GI_HDM inline
#include "../../gi_raytracer_amd/csrc/gi_layout.h"

using namespace gi;

struct Emul {
    HostScene hs;
    HostPhotons hp;
    std::vector<HaltonDim> hdims;
    std::vector<uint16_t> htable;
    Scene S{};
    std::string err;
    bool wide = true, cull = true;
    void bind()
    {
        S.tnodes = hs.tnodes.data(); S.leaf_refs = hs.refs.data(); S.leaf_tris = hs.leaf_tris.data(); S.tris = hs.tris.data(); S.shade = hs.shade.data();
        S.mats = hs.mats.data(); S.lights = hs.lights.data();
        auto off = [](const char* name) { const char* e = getenv(name); return e && atoi(e) == 0; };   // the product's knobs (gi_kernels.hip: set_walk_shortcuts)
        S.leaf_boxes = off("GI_ENTITY_BOXES") ? nullptr : hs.leaf_boxes.data();
        S.trace_boxes = !S.leaf_boxes ? nullptr : ((off("GI_CLIP_BOXES") || !hs.clipped) ? hs.leaf_boxes.data() : hs.trace_boxes.data());
        S.cut_margin = (S.leaf_boxes && !off("GI_WALK_CUT")) ? hs.cut_margin : -1.0;
        S.n_node = hs.n_node; S.n_tri = hs.n_tri; S.n_light = hs.n_light; S.has_spheres = 1;
        if (!hs.tnodes.empty()) for (int k = 0; k < 3; k++) { S.root_bmin[k] = hs.tnodes[0].bmin[k]; S.root_bmax[k] = hs.tnodes[0].bmax[k]; }
        S.n_wnode = (int32_t)hs.wnodes.size();
        S.wnodes = (wide && S.n_wnode > 0) ? hs.wnodes.data() : nullptr;
        S.wleaf_id = hs.wleaf_id.data();
        S.cboxes = (cull && S.wnodes && hs.cboxes.size() > 1) ? hs.cboxes.data() : nullptr;
        S.cuse = hs.cuse.data();
        const bool cut_to_leaves = S.cboxes && S.trace_boxes == hs.trace_boxes.data() && hs.tcboxes.size() == hs.cboxes.size();   // (gi_kernels.hip: set_walk_shortcuts)
        S.tcboxes = cut_to_leaves ? hs.tcboxes.data() : S.cboxes;
        S.tcuse = cut_to_leaves ? hs.tcuse.data() : S.cuse;
        const bool sh = hs.lights_clear && S.trace_boxes == hs.trace_boxes.data();
        S.shadow_boxes = sh ? S.trace_boxes : S.leaf_boxes;
        S.scboxes = sh ? S.tcboxes : S.cboxes;
        S.scuse = sh ? S.tcuse : S.cuse;
        S.tri_uv = hs.tri_uv.data(); S.texs = hs.texs.data(); S.tex_pixels = hs.tex_pixels.data(); S.tex_lut = hs.tex_lut.data(); S.n_tex = hs.n_tex();
        S.fogs = hs.fogs.data(); S.fog_grid = hs.fog_grid.data(); S.n_fog = hs.n_fog();
        for (int k = 0; k < 3; k++) S.ambient[k] = hs.ambient[k];
        S.pnodes = hp.nodes.data(); S.pranges = hp.ranges.data(); S.ph_pos = hp.pos.data(); S.ph_dircol = hp.dircol.data();
        S.n_pnode = hp.n_node; S.n_photon = hp.n_photon;
        S.pn_planes = (wide && hp.planes_ok) ? 1 : 0;
        S.hdims = hdims.data(); S.htable = htable.data();
    }
};

extern "C" {

Emul* emul_create() { Emul* e = new Emul(); build_halton_tables(e->hdims, e->htable); e->bind(); return e; }
void emul_destroy(Emul* e) { delete e; }
int emul_set_wide(Emul* e, int on) { e->wide = on != 0; e->bind(); return (e->S.wnodes != nullptr ? 1 : 0) | (e->S.pn_planes ? 2 : 0); }
int emul_set_cull(Emul* e, int on) { e->cull = on != 0; e->bind(); return e->S.cboxes != nullptr ? 1 : 0; }
const char* emul_error(Emul* e) { return e->err.c_str(); }
int emul_upload_scene(Emul* e, const gi_scene_desc* d)
{
    if (!layout_scene(d, e->hs, e->err)) return GI_E_INVALID;
    e->hp = HostPhotons();
    e->bind();
    return 0;
}
int emul_upload_photons(Emul* e, const gi_photon_map_desc* d)
{
    if (!layout_photons(d, e->hp, e->err)) return GI_E_INVALID;
    e->bind();
    return 0;
}
int emul_trace(Emul* e, int n, const double* rays, int32_t* hit, int32_t* ent, double* res)
{
    for (int i = 0; i < n; i++) {
        const double* r = rays + (size_t)i * 6;
        Ray ray = make_ray_exact(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]));
        Rng rng = rng_make(0, (uint32_t)i);
        HitRec h;
        bool ok = trace(e->S, ray, rng, P_TRACE_ALPHA, h, nullptr);
        hit[i] = ok; ent[i] = ok ? h.tri : -1;
        double* o = res + (size_t)i * 8;
        if (ok) { V3 nn = shading_normal(e->S, h); o[0] = h.pos.x; o[1] = h.pos.y; o[2] = h.pos.z; o[3] = nn.x; o[4] = nn.y; o[5] = nn.z; o[6] = h.u; o[7] = h.v; }
        else for (int k = 0; k < 8; k++) o[k] = 0;
    }
    return 0;
}
int emul_visible(Emul* e, int n, const double* q, int32_t* vis)
{
    for (int i = 0; i < n; i++) {
        const double* p = q + (size_t)i * 6;
        V3 o = v3(p[0], p[1], p[2]), t = v3(p[3], p[4], p[5]);
        V3 ld = t - o;
        Ray sr = make_ray(o, ld);
        Rng rng = rng_make(0, (uint32_t)i);
        vis[i] = visible(e->S, sr, len2(ld), rng, 0, nullptr);
    }
    return 0;
}
// RayTracer::visible the way k_st_shadow walks it: one turn at a time (wwalk_turn), a leaf's triangles when the walk stands on one
// (visible_leaf_blocks), the medium at the end -- must answer as visible() does.  -1: the scene has no wide records.
// light_bound: the segments end at a light (what k_st_shadow is given): the boxes and content boxes it uses for them
int emul_visible_turns(Emul* e, int n, const double* q, int32_t* vis, int light_bound)
{
    if (!e->S.wnodes) return -1;
    GlobalWide W;
    W.g = e->S.wnodes; W.cboxes = light_bound ? e->S.scboxes : e->S.cboxes; W.cuse = light_bound ? e->S.scuse : e->S.cuse;
    for (int i = 0; i < n; i++) {
        const double* p = q + (size_t)i * 6;
        V3 o = v3(p[0], p[1], p[2]), t = v3(p[3], p[4], p[5]);
        V3 ld = t - o;
        Ray sr = make_ray(o, ld);
        const double mt = len2(ld);
        Rng rng = rng_make(0, (uint32_t)i);
        VisWalk v;
        bool blocked = false;
        if (visible_wide_begin<7>(e->S, W, sr, mt, v)) {
            for (;;) {
                int32_t lnode = 0, first = 0, cnt = 0;
                int lslot = 0;
                const int r = wwalk_turn(W, v.k, sr, v.wr, 0.0, v.tmax, lnode, lslot, first, cnt);
                if (r == WALK_END) break;
                if (r == WALK_LEAF && visible_leaf_blocks<7>(e->S, W, sr, mt, rng, 0u, lnode, lslot, first, cnt, light_bound ? e->S.shadow_boxes : e->S.leaf_boxes)) { blocked = true; break; }
            }
        }
        vis[i] = !blocked && visible_through_fog<7>(e->S, sr, mt, rng, 0u);
    }
    return 0;
}
int emul_gather(Emul* e, int n, const double* q, double* res3, int32_t* n_cand)
{
    float heap[GI_GATHER_K * 4];
    for (int i = 0; i < n; i++) {
        const double* p = q + (size_t)i * 6;
        int nc = 0;
        V3 r = gather(e->S, v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), heap + (i & 3), 4, &nc, nullptr);
        res3[i * 3] = r.x; res3[i * 3 + 1] = r.y; res3[i * 3 + 2] = r.z;
        if (n_cand) n_cand[i] = nc;
    }
    return 0;
}
int emul_radiance(Emul* e, int n, const double* rays, const uint32_t* stream, uint64_t seed, double* out3)
{
    float heap[GI_GATHER_K];
    for (int i = 0; i < n; i++) {
        const double* r = rays + (size_t)i * 6;
        Ray ray = make_ray_exact(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]));
        V3 L = radiance_path(e->S, ray, stream[i], seed, heap, 1, nullptr);
        out3[i * 3] = L.x; out3[i * 3 + 1] = L.y; out3[i * 3 + 2] = L.z;
    }
    return 0;
}
// per-path work profile of one frame: for every path that shades more than `min_shaded` vertices, its 8 counters (profiling aid)
int emul_deep_paths(Emul* e, const gi_render_params* p, int min_shaded, int cap, int64_t* out8, int32_t* n_out, double* rays6, uint32_t* streams)
{
    Frame F;
    if (!make_frame(p, F, e->err)) return GI_E_INVALID;
    float heap[GI_GATHER_K];
    int n = 0;
    for (int y = 0; y < F.local_rows; y++)
        for (int x = 0; x < F.w; x++)
            for (int s = 0; s < F.max_samples; s++) {
                uint32_t idx;
                Ray ray = primary_ray(e->S, F, s, x, global_row(F, y), idx);
                Counters c;
                memset(&c, 0, sizeof c);
                radiance_path(e->S, ray, idx, F.seed, heap, 1, &c);
                if ((int)c.shaded > min_shaded && n < cap) {
                    memcpy(out8 + (size_t)n * 8, &c, sizeof c);
                    if (rays6) { double* r = rays6 + (size_t)n * 6; r[0] = ray.o.x; r[1] = ray.o.y; r[2] = ray.o.z; r[3] = ray.d.x; r[4] = ray.d.y; r[5] = ray.d.z; }
                    if (streams) streams[n] = idx;
                    n++;
                }
            }
    *n_out = n;
    return 0;
}
int emul_render(Emul* e, const gi_render_params* p, double* out, int32_t* out_spp, int64_t* counters8)
{
    Frame F;
    if (!make_frame(p, F, e->err)) return GI_E_INVALID;
    float heap[GI_GATHER_K];
    Counters c;
    memset(&c, 0, sizeof c);
    const int tiles_x = (F.w + 7) >> 3, tiles_y = (F.local_rows + 7) >> 3;
    for (int tile = 0; tile < tiles_x * tiles_y; tile++)
        for (int lane = 0; lane < 64; lane++) {
            const int tx = tile % tiles_x, ty = tile / tiles_x;
            const int x = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
            if (!(x < F.w && ly < F.local_rows)) continue;
            const int y = global_row(F, ly);
            PixelState ps;
            pixel_begin(ps);
            while (pixel_wants_sample(ps, F)) {
                uint32_t idx;
                Ray ray = primary_ray(e->S, F, ps.s, x, y, idx);
                V3 L = radiance_path(e->S, ray, idx, F.seed, heap, 1, counters8 ? &c : nullptr);
                pixel_add_sample(ps, F, L);
            }
            const size_t o = ((size_t)ly * F.w + x);
            out[o * 3] = ps.color.x; out[o * 3 + 1] = ps.color.y; out[o * 3 + 2] = ps.color.z;
            if (out_spp) out_spp[o] = ps.s;
        }
    if (counters8) {
        counters8[0] = c.v_trace; counters8[1] = c.v_shadow; counters8[2] = c.tri; counters8[3] = c.shaded;
        counters8[4] = c.pcand; counters8[5] = c.traces; counters8[6] = c.shadows; counters8[7] = c.gathers;
    }
    return 0;
}
int emul_emit(Emul* e, int count, int max_depth, uint64_t seed, double* out, int cap, int64_t* tries_out)
{
    int stored = 0;
    int64_t tries = 0;
    for (int i = 0; i < count; i++)
        for (int li = 0; li < e->S.n_light; li++) {
            PhotonOut po;
            int32_t t = 0;
            bool ok = emit_photon(e->S, i, li, count, max_depth, seed, po, t);
            tries += t;
            if (ok) { if (stored < cap) memcpy(out + (size_t)stored * 9, po.v, 72); stored++; }
        }
    if (tries_out) *tries_out = tries;
    return stored;
}
int emul_leaf_order(Emul* e, int n, const double* rays, int cap, int32_t* leaf_out, int32_t* n_out)
{
    for (int i = 0; i < n; i++) {
        const double* r = rays + (size_t)i * 6;
        n_out[i] = leaf_order(e->S, make_ray_exact(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])), cap, leaf_out + (size_t)i * cap);
    }
    return 0;
}
int emul_kat(int what, int n, const double* in, int in_stride, double* out3)
{
    for (int i = 0; i < n; i++) {
        double a[9];
        for (int k = 0; k < 9; k++) a[k] = k < in_stride ? in[(size_t)i * in_stride + k] : 0.0;
        kat_eval(what, a, out3 + (size_t)i * 3);
    }
    return 0;
}
float emul_halton_sample(Emul* e, uint32_t dim, uint32_t index) { return halton_sample(e->S, dim, index); }
uint32_t emul_halton_index(int w, int h, uint32_t s, uint32_t x, uint32_t y) { return halton_index(make_halton_enum(w, h), s, x, y); }
double emul_rng(uint64_t seed, uint32_t stream, uint32_t depth, uint32_t purpose, uint32_t a, uint32_t b)
{
    Rng r = rng_make(seed, stream);
    r.depth = depth;
    return rng_draw(r, purpose, a, b);
}
}
