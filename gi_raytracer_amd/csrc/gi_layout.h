// gi_layout.h -- host-side construction of the device tables (pure C++, no HIP calls).
//
// Turns the flat C-ABI descriptions (include/gi_hip.h) into the arrays the kernels read (gi_device.h):
//   * 8 direction-ordered pre-order copies of the scene octree with skip links (stackless front-to-back traversal),
//   * per-triangle test record {p0, e1, e2, material, flags} and shading record,
//   * photon octree in pre-order with skip links and photons re-ordered leaf by leaf,
//   * Halton/Faure digit-group tables and the per-frame Halton enumeration constants.
// Used by gi_kernels.hip (which uploads the vectors) and by tests/host_emul (which points a gi::Scene at them).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gi_hip.h"
#include "gi_device.h"

namespace gi {

struct HostScene {
    std::vector<TNode> tnodes;
    std::vector<WNode> wnodes;        // empty when the tree is not made of exact octants (layout_wide)
    std::vector<int32_t> wleaf_id;
    std::vector<float> cboxes;        // [n_wnode][8][6] content boxes of the children (gi_device.h: content_cull); one dummy entry without wide records
    std::vector<uint32_t> cuse;       // [n_wnode] children worth testing
    std::vector<int32_t> refs;
    std::vector<LeafTri> leaf_tris;
    std::vector<double> leaf_boxes;   // [n_refs][6] the entity's own box, widened (gi_device.h: entity_box_missed), parallel to leaf_tris
    std::vector<double> trace_boxes;  // [n_refs][6] for the closest-hit walk: the box of the part of an opaque entity inside its leaf (gi_device.h: trace_wide_step)
    double cut_margin = -1;           // what the closest-hit walk's short cuts allow for rounding (1e-5 of the scene)
    bool clipped = false;             // trace_boxes differs from leaf_boxes
    bool lights_clear = false;        // no entity's box within a light's radius + twice the shadow bias (+ 4 margins) of the light: nothing can block a shadow segment in its last stretch
    std::vector<float> tcboxes;       // content boxes made of the trace boxes: what the closest-hit walk culls by (like cboxes; empty when !clipped)
    std::vector<uint32_t> tcuse;
    std::vector<int32_t> worder;      // canonical node of every wide record
    std::vector<TriGeom> tris;
    std::vector<TriShade> shade;
    std::vector<TriUV> tri_uv;
    std::vector<Mat> mats;
    std::vector<LightD> lights;
    std::vector<FogD> fogs;
    std::vector<double> fog_grid;
    std::vector<TexD> texs;           // only when the scene has a non-constant texture (n_tex() > 0)
    std::vector<unsigned char> tex_pixels;
    std::vector<double> tex_lut;
    int32_t n_tex() const { return (int32_t)texs.size(); }
    int32_t n_node = 0, n_tri = 0, n_light = 0;
    double ambient[3] = {0, 0, 0};
    int32_t n_fog() const { return (fogs.size() == 1 && fogs[0].grid_n == 0) ? 0 : (int32_t)fogs.size(); }
};
struct HostPhotons {
    std::vector<PNode> nodes;
    std::vector<PRange> ranges;
    std::vector<double> pos, dircol;
    int32_t n_node = 0, n_photon = 0;
    bool planes_ok = true;            // inner records' planes reproduce every child box (gather_find_leaf's one-record-per-level walk)
};

// Faure permutations and per-base digit-group tables of Halton_sampler (include/halton_sampler.h:573-603,890-1414).
inline void build_halton_tables(std::vector<HaltonDim>& dims, std::vector<uint16_t>& table)
{
    const unsigned max_base = 1619u;
    std::vector<std::vector<uint16_t>> perm(max_base + 1);
    for (unsigned k = 1; k <= 3; ++k) { perm[k].resize(k); for (unsigned i = 0; i < k; ++i) perm[k][i] = (uint16_t)i; }
    for (unsigned base = 4; base <= max_base; ++base) {
        std::vector<uint16_t>& pb = perm[base];
        pb.resize(base);
        const unsigned half = base / 2;
        if (base & 1) {
            const std::vector<uint16_t>& prev = perm[base - 1];
            for (unsigned i = 0; i + 1 < base; ++i) pb[i + (i >= half)] = (uint16_t)(prev[i] + (prev[i] >= half));
            pb[half] = (uint16_t)half;
        } else {
            const std::vector<uint16_t>& hp = perm[half];
            for (unsigned i = 0; i < half; ++i) { pb[i] = (uint16_t)(2 * hp[i]); pb[half + i] = (uint16_t)(2 * hp[i] + 1); }
        }
    }
    std::vector<unsigned> primes;
    for (unsigned c = 2; primes.size() < 256; c++) {
        bool is_prime = true;
        for (unsigned p : primes) { if (p * p > c) break; if (c % p == 0) { is_prime = false; break; } }
        if (is_prime) primes.push_back(c);
    }
    dims.resize(256);
    table.clear();
    for (unsigned d = 0; d < 256; d++) {
        const unsigned b = primes[d];
        unsigned digits = 1, P = b;
        while ((unsigned long long)P * b <= 500ull) { P *= b; digits++; }      // table over the largest b^g <= 500
        unsigned groups = 1;
        unsigned long long full = P;
        while (full * P < 4294967296ull) { full *= P; groups++; }               // groups with (b^g)^n < 2^32
        dims[d].P = P; dims[d].n = groups; dims[d].off = (uint32_t)table.size();
        dims[d].scale = float(0.9999998807907104 / (double)full);
        // groups <= 7: the longest chain is base 23 (23^7 < 2^32 <= 23^8); halton_sample unrolls 7 (all 256 dims are checked bit for bit by the tests)
        const unsigned long long magic = 0xffffffffffffffffull / P + 1ull;
        dims[d].mlo = (uint32_t)magic; dims[d].mhi = (uint32_t)(magic >> 32); dims[d].pad[0] = dims[d].pad[1] = 0;
        for (unsigned i = 0; i < P; i++) {
            unsigned rest = i;
            uint16_t inv = 0;
            for (unsigned k = 0; k < digits; ++k) { inv = (uint16_t)(inv * b + perm[b][rest % b]); rest /= b; }
            table.push_back(inv);
        }
    }
}

inline HaltonEnumD make_halton_enum(unsigned width, unsigned height)  // Halton_enum ctor, include/halton_enum.h:69-104
{
    HaltonEnumD e;
    unsigned w = 1, h = 1;
    e.p2 = 0; while (w < width) { ++e.p2; w *= 2; }
    e.p3 = 0; while (h < height) { ++e.p3; h *= 3; }
    e.scale_x = float(w); e.scale_y = float(h);
    e.inc = w * h;
    // extended Euclid on (h, w), iterative: s0*h + t0*w == gcd == 1
    long long a = (long long)h, b = (long long)w, s0 = 1, s1 = 0, t0 = 0, t1 = 1;
    while (b) {
        long long q = a / b, r = a % b;
        a = b; b = r;
        long long s2 = s0 - q * s1; s0 = s1; s1 = s2;
        long long t2 = t0 - q * t1; t0 = t1; t1 = t2;
    }
    unsigned inv2 = (s0 < 0) ? (unsigned)(s0 + (long long)w) : (unsigned)(s0 % (long long)w);
    unsigned inv3 = (t0 < 0) ? (unsigned)(t0 + (long long)h) : (unsigned)(t0 % (long long)h);
    e.m_x = h * inv2;
    e.m_y = w * inv3;
    return e;
}

inline bool validate_scene(const gi_scene_desc* d, std::string& err)
{
    if (!d || d->n_tri < 0 || d->n_mat <= 0 || d->n_light < 0 || d->n_node <= 0) { err = "scene: bad counts"; return false; }
    if (d->n_tri && (!d->tri_pos || !d->tri_nrm || !d->tri_mat)) { err = "scene: null triangle tables"; return false; }
    if (!d->mats || !d->node_bbox || !d->node_child || !d->node_ent_off || (d->n_light && !d->lights)) { err = "scene: null tables"; return false; }
    for (int i = 0; i < d->n_tri; i++) {
        if (d->tri_mat[i] < 0 || d->tri_mat[i] >= d->n_mat) { err = "scene: material index out of range"; return false; }
        if (d->ent_kind && d->ent_kind[i] != 0 && d->ent_kind[i] != 1) { err = "scene: unknown entity kind"; return false; }
    }
    if (d->n_mat >= (1 << 28)) { err = "scene: too many materials"; return false; }
    if (d->n_tex < 0 || (d->n_tex > 0 && (!d->tex_kind || !d->tex_param || !d->mat_tex))) { err = "scene: null texture tables"; return false; }
    for (int t = 0; t < d->n_tex; t++) {
        const double* q = d->tex_param + (size_t)t * 8;
        if (d->tex_kind[t] < 0 || d->tex_kind[t] > 2) { err = "scene: unknown texture kind"; return false; }
        if (d->tex_kind[t] == 2) {
            const double w = q[2], h = q[3], off = q[5];
            if (!(w >= 1 && h >= 1 && w <= 32768 && h <= 32768 && off >= 0) || !d->tex_pixels || off + w * h * 4 > (double)d->n_tex_pixel_bytes) { err = "scene: image texture outside the pixel table"; return false; }
        }
    }
    if (d->n_tex > 0)
        for (int m = 0; m < d->n_mat * 2; m++)
            if (d->mat_tex[m] < -1 || d->mat_tex[m] >= d->n_tex) { err = "scene: material texture index out of range"; return false; }
    const int nref = d->node_ent_off[d->n_node];
    if (d->node_ent_off[0] != 0 || nref < 0 || (nref && !d->node_ent_idx)) { err = "scene: bad leaf reference table"; return false; }
    for (int n = 0; n < d->n_node; n++) {
        if (d->node_ent_off[n + 1] < d->node_ent_off[n]) { err = "scene: leaf offsets not monotone"; return false; }
        for (int k = 0; k < 8; k++) {
            int ch = d->node_child[(size_t)n * 8 + k];
            if (ch != -1 && (ch <= n || ch >= d->n_node)) { err = "scene: child index is not a later pre-order node"; return false; }
        }
    }
    for (int r = 0; r < nref; r++)
        if (d->node_ent_idx[r] < 0 || d->node_ent_idx[r] >= d->n_tri) { err = "scene: leaf reference out of range"; return false; }
    return true;
}

// per-octant links of the canonical tree: visiting the children of every node in the order k ^ a (k = 0..7) is front to back for
// every ray whose direction signs are `a` (child numbering x = bit0, z = bit1, y = bit2 as in include/octree.cpp:321-328).
// hit = first child in that order; skip = next sibling in that order, or the parent's skip after the last child.
inline void link_octant(const gi_scene_desc* d, int a, int node, int32_t skip_to, const std::vector<int32_t>& rec_of, std::vector<TNode>& out)
{
    TNode& t = out[(size_t)rec_of[node]];
    int kids[8], nk = 0;
    for (int k = 0; k < 8; k++) { int ch = d->node_child[(size_t)node * 8 + (k ^ a)]; if (ch >= 0) kids[nk++] = ch; }
    t.link[a].skip = skip_to;
    t.link[a].hit = nk ? rec_of[kids[0]] : skip_to;
    for (int i = 0; i < nk; i++) link_octant(d, a, kids[i], i + 1 < nk ? rec_of[kids[i + 1]] : skip_to, rec_of, out);
}

// Content boxes: for every child of every wide record the union of the boxes of all entities referenced in its sub-tree, rounded outwards to
// float, and (cuse) which children's boxes are clearly smaller than their octants.  ref_boxes = null: the entities' whole boxes, widened here by
// 1e-7 of the scene; else one (already widened) box per leaf reference -- an entry with b[0] = 1e300 stands for "nothing of it near the leaf".
inline void layout_cboxes(const gi_scene_desc* d, const HostScene& H, const double* ref_boxes, std::vector<float>& cboxes, std::vector<uint32_t>& cuse)
{
    const int N = d->n_node;
    const std::vector<int32_t>& order = H.worder;
    std::vector<double> cb((size_t)N * 6);
    for (int n = N - 1; n >= 0; n--) {      // children come later in pre-order
        double* b = &cb[(size_t)n * 6];
        b[0] = b[1] = b[2] = INFINITY; b[3] = b[4] = b[5] = -INFINITY;
        for (int r = d->node_ent_off[n]; r < d->node_ent_off[n + 1]; r++) {
            if (ref_boxes) {
                const double* q = ref_boxes + (size_t)r * 6;
                if (q[0] == 1e300) continue;
                for (int ax = 0; ax < 3; ax++) { b[ax] = std::min(b[ax], q[ax]); b[3 + ax] = std::max(b[3 + ax], q[3 + ax]); }
                continue;
            }
            const int e = d->node_ent_idx[r];
            const double* P = d->tri_pos + (size_t)e * 9;
            if (d->ent_kind && d->ent_kind[e] == 1) {          // sphere: centre P[0..2], radius P[3]
                for (int ax = 0; ax < 3; ax++) { b[ax] = std::min(b[ax], P[ax] - P[3]); b[3 + ax] = std::max(b[3 + ax], P[ax] + P[3]); }
            } else
                for (int v = 0; v < 3; v++) for (int ax = 0; ax < 3; ax++) { b[ax] = std::min(b[ax], P[v * 3 + ax]); b[3 + ax] = std::max(b[3 + ax], P[v * 3 + ax]); }
        }
        for (int k = 0; k < 8; k++) {
            const int ch = d->node_child[(size_t)n * 8 + k];
            if (ch < 0) continue;
            for (int ax = 0; ax < 3; ax++) { b[ax] = std::min(b[ax], cb[(size_t)ch * 6 + ax]); b[3 + ax] = std::max(b[3 + ax], cb[(size_t)ch * 6 + 3 + ax]); }
        }
    }
    double extent = 0;
    for (int ax = 0; ax < 3; ax++) extent = std::max(extent, d->node_bbox[3 + ax] - d->node_bbox[ax]);
    const double margin = ref_boxes ? 0.0 : 1e-7 * std::max(extent, 1e-3);     // a hit point is off its entity by ~1e-16 of its coordinates: eight orders of slack
    cboxes.assign(order.size() * 48, 0.f);
    cuse.assign(order.size(), 0u);
    for (size_t r = 0; r < order.size(); r++) {
        const int n = order[r];
        for (int c = 0; c < 8; c++) {
            const int ch = d->node_child[(size_t)n * 8 + c];
            if (ch < 0 || !(H.wnodes[r].exists & (1u << c))) continue;
            const double* b = &cb[(size_t)ch * 6];
            float* out = &cboxes[(r * 8 + (size_t)c) * 6];
            double share = 1.0;                                // how much of the octant the content (clipped to it) fills
            for (int ax = 0; ax < 3; ax++) {
                // (a sub-tree none of whose entities comes near its leaves: an empty box, which no ray touches)
                out[ax] = b[ax] > b[3 + ax] ? INFINITY : std::nextafterf((float)(b[ax] - margin), -INFINITY);
                out[3 + ax] = b[ax] > b[3 + ax] ? -INFINITY : std::nextafterf((float)(b[3 + ax] + margin), INFINITY);
                const double olo = d->node_bbox[(size_t)ch * 6 + ax], ohi = d->node_bbox[(size_t)ch * 6 + 3 + ax];
                const double len = std::min((double)out[3 + ax], ohi) - std::max((double)out[ax], olo);
                share *= ohi > olo ? std::max(0.0, std::min(1.0, len / (ohi - olo))) : 1.0;
            }
            if (share < 0.6) cuse[r] |= 1u << c;             // the extra test is only made where it can reject something
        }
    }
}
// Wide records (gi_device.h: WNode) for the inner nodes, breadth first.  Every existing child's box must be, bit for bit, the
// octant Octree::Node::partition gives it (include/octree.cpp:318-328): low side [min, mid], high side [lo2, hi2], child 7
// [mid, max], with one value of mid / lo2 / hi2 per axis and node.  Returns false (and leaves H.wnodes empty) for any other tree.
inline bool layout_wide(const gi_scene_desc* d, HostScene& H)
{
    H.wnodes.clear(); H.wleaf_id.clear();
    const int N = d->n_node;
    auto inner = [&](int n) { for (int k = 0; k < 8; k++) if (d->node_child[(size_t)n * 8 + k] >= 0) return true; return false; };
    if (N < 1 || !inner(0)) return false;
    std::vector<int32_t> wrec((size_t)N, -1), order, depth((size_t)N, 0);
    wrec[0] = 0; order.push_back(0);
    for (size_t head = 0; head < order.size(); head++) {
        const int n = order[head];
        for (int k = 0; k < 8; k++) {
            const int ch = d->node_child[(size_t)n * 8 + k];
            if (ch < 0 || !inner(ch)) continue;
            depth[ch] = depth[n] + 1;
            if (depth[ch] > 15) return false;          // the walk keeps one mask byte per level in 128 bits
            wrec[ch] = (int32_t)order.size();
            order.push_back(ch);
        }
    }
    std::vector<WNode> W(order.size());
    std::vector<int32_t> L(order.size() * 8, -1);
    for (WNode& w : W) { memset(&w, 0, sizeof w); w.parent = -1; }
    for (size_t r = 0; r < order.size(); r++) {
        const int n = order[r];
        WNode& w = W[r];
        const double* pb = d->node_bbox + (size_t)n * 6;
        for (int ax = 0; ax < 3; ax++) {
            const double mn = pb[ax], mx = pb[3 + ax];
            // the values partition() computes; replaced below by what the children's boxes actually hold
            double mid = mn * 0.5 + mx * 0.5, lo2 = mn + .5 * (mx - mn), hi2 = mid + .5 * (mx - mn);
            bool have_mid = false, have_hi = false;
            const int bitpos = ax == 0 ? 0 : (ax == 1 ? 2 : 1);   // slot bit of this axis: x = bit 0, z = bit 1, y = bit 2
            for (int c = 0; c < 8; c++) {
                const int ch = d->node_child[(size_t)n * 8 + c];
                if (ch < 0) continue;
                const double clo = d->node_bbox[(size_t)ch * 6 + ax], chi = d->node_bbox[(size_t)ch * 6 + 3 + ax];
                const int b = (c >> bitpos) & 1;
                if (c == 7) {
                    if (chi != mx) return false;
                    if (have_mid && clo != mid) return false;
                    mid = clo; have_mid = true;
                } else if (b == 0) {
                    if (clo != mn) return false;
                    if (have_mid && chi != mid) return false;
                    mid = chi; have_mid = true;
                } else {
                    if (have_hi && (clo != lo2 || chi != hi2)) return false;
                    lo2 = clo; hi2 = chi; have_hi = true;
                }
            }
            w.pl[ax][0] = mn; w.pl[ax][1] = mid; w.pl[ax][2] = lo2; w.pl[ax][3] = hi2; w.pl[ax][4] = mx; w.pl[ax][5] = mid;
        }
        for (int c = 0; c < 8; c++) {
            const int ch = d->node_child[(size_t)n * 8 + c];
            if (ch < 0) continue;
            if (inner(ch)) { w.ca[c] = wrec[ch]; w.cb[c] = -1; w.exists |= 1u << c; W[(size_t)wrec[ch]].parent = (int32_t)r; }
            else {
                const int cnt = d->node_ent_off[ch + 1] - d->node_ent_off[ch];
                w.ca[c] = d->node_ent_off[ch]; w.cb[c] = cnt;
                if (cnt > 0) w.exists |= 1u << c;
                L[r * 8 + c] = ch;
            }
        }
    }
    H.worder.assign(order.begin(), order.end());
    H.wnodes.swap(W);
    H.wleaf_id.swap(L);
    layout_cboxes(d, H, nullptr, H.cboxes, H.cuse);
    return true;
}

inline bool layout_scene(const gi_scene_desc* d, HostScene& H, std::string& err)
{
    if (!validate_scene(d, err)) return false;
    // breadth-first record order: the nodes every ray visits (top levels) come first and are what the kernels keep in LDS
    const int N = d->n_node;
    std::vector<int32_t> rec_of((size_t)N, -1), order;
    order.reserve((size_t)N);
    rec_of[0] = 0;
    order.push_back(0);
    for (size_t head = 0; head < order.size(); head++) {
        const int n = order[head];
        for (int k = 0; k < 8; k++) {
            const int ch = d->node_child[(size_t)n * 8 + k];
            if (ch < 0) continue;
            if (rec_of[ch] != -1) { err = "scene: octree node has two parents"; return false; }
            rec_of[ch] = (int32_t)order.size();
            order.push_back(ch);
        }
    }
    if ((int)order.size() != N) { err = "scene: octree is not one connected tree"; return false; }
    H.tnodes.assign((size_t)N, TNode());
    for (int n = 0; n < N; n++) {
        TNode& t = H.tnodes[(size_t)rec_of[n]];
        memset(&t, 0, sizeof t);
        for (int k = 0; k < 3; k++) { t.bmin[k] = d->node_bbox[(size_t)n * 6 + k]; t.bmax[k] = d->node_bbox[(size_t)n * 6 + 3 + k]; }
        bool inner = false;
        for (int k = 0; k < 8; k++) if (d->node_child[(size_t)n * 8 + k] >= 0) inner = true;
        t.first_ref = d->node_ent_off[n];
        t.n_ref = inner ? -1 : d->node_ent_off[n + 1] - d->node_ent_off[n];
        t.leaf_id = n;
    }
    for (int a = 0; a < 8; a++) link_octant(d, a, 0, N, rec_of, H.tnodes);
    if (!layout_wide(d, H)) { H.wnodes.clear(); H.wleaf_id.clear(); H.cboxes.clear(); H.cuse.clear(); }
    if (H.cboxes.empty()) H.cboxes.assign(1, 0.f);
    if (H.cuse.empty()) H.cuse.assign(1, 0u);
    H.refs.assign(d->node_ent_idx, d->node_ent_idx + d->node_ent_off[d->n_node]);
    H.tris.resize((size_t)d->n_tri);
    H.shade.resize((size_t)d->n_tri);
    H.tri_uv.resize((size_t)d->n_tri);
    for (int i = 0; i < d->n_tri; i++) {
        const double* P = d->tri_pos + (size_t)i * 9;
        const double* N = d->tri_nrm + (size_t)i * 9;
        V3 p0 = ld3(P), p1 = ld3(P + 3), p2 = ld3(P + 6);
        V3 e1 = p1 - p0, e2 = p2 - p0;
        TriGeom& g = H.tris[i];
        g.p0[0] = p0.x; g.p0[1] = p0.y; g.p0[2] = p0.z;
        g.e1[0] = e1.x; g.e1[1] = e1.y; g.e1[2] = e1.z;
        g.e2[0] = e2.x; g.e2[1] = e2.y; g.e2[2] = e2.z;
        g.mat = d->tri_mat[i];
        const double* m = d->mats + (size_t)g.mat * 9;
        const bool sphere = d->ent_kind && d->ent_kind[i] == 1;
        const bool smooth = !sphere && len2(ld3(N)) > 0 && len2(ld3(N + 3)) > 0 && len2(ld3(N + 6)) > 0;   // include/entities.h:478
        bool tex_has_alpha = false;   // Material::getAlpha = opacity * diffuse->getAlpha(uv): below 1 wherever an image's alpha channel says so
        if (d->n_tex > 0) { const int dt = d->mat_tex[(size_t)g.mat * 2]; tex_has_alpha = dt >= 0 && d->tex_kind[dt] == 2 && d->tex_param[(size_t)dt * 8 + 4] != 0; }
        const bool always = (!tex_has_alpha && m[1] * 1.0 >= 1.0) || (m[2] != 1);                // include/raytracer.h:455
        g.flags = (smooth ? 1u : 0u) | (always ? 2u : 0u) | (sphere ? 4u : 0u);
        if (sphere) { g.e1[0] = P[3]; g.e1[1] = 0; g.e1[2] = 0; g.e2[0] = 0; g.e2[1] = 0; g.e2[2] = 0; }   // centre in p0, radius in e1[0]
        TriShade& s = H.shade[i];
        for (int k = 0; k < 3; k++) { s.n0[k] = N[k]; s.n1[k] = N[3 + k]; s.n2[k] = N[6 + k]; }
        V3 fn = sphere ? v3(0, 0, 0) : normalize(cross((p1 - p0), (p2 - p0)));                   // include/entities.h:339
        s.fnorm[0] = fn.x; s.fnorm[1] = fn.y; s.fnorm[2] = fn.z;
        if (sphere) { s.n0[0] = P[0]; s.n0[1] = P[1]; s.n0[2] = P[2]; }                         // centre, for the shading normal
        TriUV& tu = H.tri_uv[i];
        for (int k = 0; k < 2; k++) { tu.t0[k] = d->tri_uv ? d->tri_uv[(size_t)i * 6 + k] : 0; tu.t1[k] = d->tri_uv ? d->tri_uv[(size_t)i * 6 + 2 + k] : 0; tu.t2[k] = d->tri_uv ? d->tri_uv[(size_t)i * 6 + 4 + k] : 0; }
    }
    H.leaf_tris.resize(H.refs.size());
    for (size_t r = 0; r < H.refs.size(); r++) {
        const TriGeom& g = H.tris[(size_t)H.refs[r]];
        LeafTri& lt = H.leaf_tris[r];
        for (int k = 0; k < 3; k++) { lt.p0[k] = g.p0[k]; lt.e1[k] = g.e1[k]; lt.e2[k] = g.e2[k]; }
        lt.tri = H.refs[r];
        lt.matflags = ((uint32_t)g.mat << 3) | g.flags;
    }
    if (H.leaf_tris.empty()) H.leaf_tris.resize(1);
    {   // the entities' own boxes, leaf-reference order: what a ray has to touch before Entity::intersect can succeed
        double extent = 0, reach = 0;
        for (int ax = 0; ax < 3; ax++) {
            extent = std::max(extent, d->node_bbox[3 + ax] - d->node_bbox[ax]);
            reach = std::max(reach, std::max(std::fabs(d->node_bbox[ax]), std::fabs(d->node_bbox[3 + ax])));
        }
        const double margin = 1e-7 * std::max(extent, 1e-3);     // as for the content boxes: a hit point is off its entity by ~1e-16 of its coordinates
        H.leaf_boxes.assign(std::max<size_t>(H.refs.size(), 1) * 6, 0.0);
        for (size_t r = 0; r < H.refs.size(); r++) {
            const int e = H.refs[r];
            const double* P = d->tri_pos + (size_t)e * 9;
            double* b = &H.leaf_boxes[r * 6];
            if (d->ent_kind && d->ent_kind[e] == 1)
                for (int ax = 0; ax < 3; ax++) { b[ax] = P[ax] - P[3] - margin; b[3 + ax] = P[ax] + P[3] + margin; }
            else
                for (int ax = 0; ax < 3; ax++) {
                    b[ax] = std::min(P[ax], std::min(P[3 + ax], P[6 + ax])) - margin;
                    b[3 + ax] = std::max(P[ax], std::max(P[3 + ax], P[6 + ax])) + margin;
                }
        }
        // The closest-hit walk's boxes: the part of the entity inside the leaf that refers to it.  The tree builder decides in float arithmetic which
        // leaves an entity overlaps (relative error 1e-7 of the coordinates), so the leaf is taken 1e-5 of the scene larger before the entity
        // is cut to it, and the box of what is left is widened by as much again: a hit point within rounding of a leaf is never refused there.
        // Entities with an alpha test keep their whole box (the test draws per leaf: include/raytracer.h:455), and so does everything in a
        // scene with textures (a hit leaves its uv behind for the next entity's alpha look-up).
        const double wide = 1e-5 * std::max(std::max(extent, reach), 1e-3);
        H.cut_margin = wide;
        H.trace_boxes = H.leaf_boxes;
        H.clipped = false;
        const bool textured = [&] { for (int t = 0; t < d->n_tex; t++) if (d->tex_kind[t] != 0) return true; return false; }();
        for (int n = 0; n < d->n_node && !textured; n++) {
            bool inner = false;
            for (int k = 0; k < 8; k++) if (d->node_child[(size_t)n * 8 + k] >= 0) inner = true;
            if (inner) continue;
            double lo[3], hi[3];
            for (int ax = 0; ax < 3; ax++) { lo[ax] = d->node_bbox[(size_t)n * 6 + ax] - wide; hi[ax] = d->node_bbox[(size_t)n * 6 + 3 + ax] + wide; }
            for (int r = d->node_ent_off[n]; r < d->node_ent_off[n + 1]; r++) {
                const int e = H.refs[(size_t)r];
                if (!(H.tris[(size_t)e].flags & 2u)) continue;
                const double* P = d->tri_pos + (size_t)e * 9;
                double* b = &H.trace_boxes[(size_t)r * 6];
                H.clipped = true;
                double cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
                if (d->ent_kind && d->ent_kind[e] == 1) {
                    for (int ax = 0; ax < 3; ax++) { cmin[ax] = std::max(P[ax] - P[3], lo[ax]); cmax[ax] = std::min(P[ax] + P[3], hi[ax]); }
                } else {
                    // Sutherland-Hodgman: the triangle against the six planes of the (enlarged) leaf, in doubles
                    double poly[2][10][3];
                    int np = 3, cur = 0;
                    for (int v = 0; v < 3; v++) for (int ax = 0; ax < 3; ax++) poly[0][v][ax] = P[v * 3 + ax];
                    for (int pl = 0; pl < 6 && np > 0; pl++) {
                        const int ax = pl >> 1;
                        const bool upper = pl & 1;
                        const double lim = upper ? hi[ax] : lo[ax];
                        int nn = 0;
                        for (int v = 0; v < np; v++) {
                            const double* A = poly[cur][v];
                            const double* B = poly[cur][(v + 1) % np];
                            const bool ain = upper ? A[ax] <= lim : A[ax] >= lim, bin = upper ? B[ax] <= lim : B[ax] >= lim;
                            if (ain) { for (int k = 0; k < 3; k++) poly[cur ^ 1][nn][k] = A[k]; nn++; }
                            if (ain != bin) {
                                const double f = (lim - A[ax]) / (B[ax] - A[ax]);
                                for (int k = 0; k < 3; k++) poly[cur ^ 1][nn][k] = k == ax ? lim : A[k] + f * (B[k] - A[k]);
                                nn++;
                            }
                        }
                        cur ^= 1;
                        np = nn;
                    }
                    for (int v = 0; v < np; v++) for (int ax = 0; ax < 3; ax++) { cmin[ax] = std::min(cmin[ax], poly[cur][v][ax]); cmax[ax] = std::max(cmax[ax], poly[cur][v][ax]); }
                }
                const bool empty = !(cmin[0] <= cmax[0] && cmin[1] <= cmax[1] && cmin[2] <= cmax[2]);
                // nothing of the entity near this leaf: a point no ray of the scene reaches (entity_box_missed has no empty box)
                for (int ax = 0; ax < 3; ax++) { b[ax] = empty ? 1e300 : std::max(b[ax], cmin[ax] - wide); b[3 + ax] = empty ? 1e300 : std::min(b[3 + ax], cmax[ax] + wide); }
            }
        }
        H.tcboxes.clear(); H.tcuse.clear();
        if (H.clipped && !H.wnodes.empty()) layout_cboxes(d, H, H.trace_boxes.data(), H.tcboxes, H.tcuse);
        // RayTracer::visible walks a segment up to GI_SHADOW_BIAS before the light but lets anything up to the light itself block it
        // (include/raytracer.h:290-305): a hit in that last stretch is only found from leaves the walk meets earlier.  Where no entity comes that
        // near a light (every scene of the reference: lights hang in free space) the stretch is empty and the leaf-cut boxes serve shadow segments too.
        H.lights_clear = d->n_light > 0;
        for (int li = 0; li < d->n_light; li++) {
            const double* l = d->lights + (size_t)li * 11;
            const double R = l[6] + 2 * GI_SHADOW_BIAS + 4 * wide;
            for (int e = 0; e < d->n_tri && H.lights_clear; e++) {
                const double* P = d->tri_pos + (size_t)e * 9;
                double d2 = 0;
                for (int ax = 0; ax < 3; ax++) {
                    const bool sph = d->ent_kind && d->ent_kind[e] == 1;
                    const double lo = sph ? P[ax] - P[3] : std::min(P[ax], std::min(P[3 + ax], P[6 + ax]));
                    const double hi = sph ? P[ax] + P[3] : std::max(P[ax], std::max(P[3 + ax], P[6 + ax]));
                    const double g = l[ax] < lo ? lo - l[ax] : (l[ax] > hi ? l[ax] - hi : 0.0);
                    d2 += g * g;
                }
                if (!(d2 > R * R)) H.lights_clear = false;
            }
        }
    }
    H.mats.resize((size_t)d->n_mat);
    for (int i = 0; i < d->n_mat; i++) {
        const double* m = d->mats + (size_t)i * 9;
        H.mats[i].roughness = m[0]; H.mats[i].opacity = m[1]; H.mats[i].ior = m[2];
        for (int k = 0; k < 3; k++) { H.mats[i].diffuse[k] = m[3 + k]; H.mats[i].emissive[k] = m[6 + k]; }
        H.mats[i].dtex = d->n_tex > 0 ? d->mat_tex[(size_t)i * 2] : -1;
        H.mats[i].etex = d->n_tex > 0 ? d->mat_tex[(size_t)i * 2 + 1] : -1;
    }
    H.lights.resize((size_t)d->n_light);
    for (int i = 0; i < d->n_light; i++) {
        const double* l = d->lights + (size_t)i * 11;
        for (int k = 0; k < 3; k++) { H.lights[i].pos[k] = l[k]; H.lights[i].col[k] = l[3 + k]; H.lights[i].dir[k] = l[7 + k]; }
        H.lights[i].rad = l[6]; H.lights[i].angle = l[10];
    }
    // textures: device records only when something is not a constant colour (then the kernels take the GI_FEAT_TEX instances)
    H.texs.clear(); H.tex_pixels.clear(); H.tex_lut.clear();
    bool textured = false;
    for (int t = 0; t < d->n_tex; t++) if (d->tex_kind[t] != 0) textured = true;
    if (textured) {
        H.texs.resize((size_t)d->n_tex);
        for (int t = 0; t < d->n_tex; t++) {
            const double* q = d->tex_param + (size_t)t * 8;
            TexD& x = H.texs[t];
            memset(&x, 0, sizeof x);
            x.kind = d->tex_kind[t];
            if (x.kind == 2) { x.tile_u = q[0]; x.tile_v = q[1]; x.w = (int32_t)q[2]; x.h = (int32_t)q[3]; x.has_alpha = q[4] != 0; x.pix = (unsigned long long)q[5]; }
            else { for (int k = 0; k < 3; k++) { x.a[k] = q[k]; x.b[k] = q[3 + k]; } x.tiles = q[6]; }
        }
        if (d->n_tex_pixel_bytes > 0) H.tex_pixels.assign(d->tex_pixels, d->tex_pixels + d->n_tex_pixel_bytes);
        // imageTexture::get: gamma({p/255}, 1.0/GAMMA) = pow(p / 255.0, 1.0 / (1.0 / 2.2)) (include/material.h:67, include/util.h:94-97),
        // evaluated once per 8-bit value with the host's libm -- the library the reference itself calls
        H.tex_lut.resize(256);
        const double g = 1.0 / 2.2;
        for (int k = 0; k < 256; k++) H.tex_lut[k] = std::pow(k / 255.0, 1.0 / g);
    } else {
        for (Mat& m : H.mats) { m.dtex = -1; m.etex = -1; }
    }
    H.fogs.clear(); H.fog_grid.clear();
    if (d->n_fog < 0 || (d->n_fog > 0 && (!d->fog || !d->fog_grid_off || !d->fog_grid))) { err = "scene: bad atmosphere tables"; return false; }
    for (int i = 0; i < d->n_fog; i++) {
        const double* q = d->fog + (size_t)i * 12;
        FogD f;
        for (int k = 0; k < 3; k++) {
            f.pos[k] = q[k]; f.size[k] = q[3 + k]; f.col[k] = q[6 + k];
            f.bmin[k] = q[k] - .5 * q[3 + k]; f.bmax[k] = q[k] + .5 * q[3 + k];   // AtmosphereEntity ctor, include/atmosphere.h:14
        }
        f.d = q[9]; f.sc = q[10];
        f.grid_off = d->fog_grid_off[i];
        f.grid_n = d->fog_grid_off[i + 1] - d->fog_grid_off[i];
        if (f.grid_n <= 0 || f.grid_off < 0) { err = "scene: empty fog noise grid"; return false; }
        H.fogs.push_back(f);
    }
    if (d->n_fog > 0) H.fog_grid.assign(d->fog_grid, d->fog_grid + d->fog_grid_off[d->n_fog]);
    if (H.fogs.empty()) { FogD z; memset(&z, 0, sizeof z); H.fogs.push_back(z); H.fog_grid.push_back(0); }   // non-null tables; n_fog stays 0
    H.n_node = d->n_node; H.n_tri = d->n_tri; H.n_light = d->n_light;
    for (int k = 0; k < 3; k++) H.ambient[k] = d->ambient[k];
    return true;
}

// photon octree.  Input: the reference's tree in pre-order (first child = n+1, each next child where the previous sub-tree
// ends).  Output: (1) photons re-ordered leaf by leaf in that DFS order; (2) node records where the 8 children of a node are
// consecutive, so the descent of PhotonMap::Node::getBounds indexes the child directly; (3) for every leaf the result of
// PhotonMap::Node::get(leaf box +- EPSILON) (include/photonMap.cpp:50-92) as a list of photon ranges -- the candidate set of a
// gather depends only on the leaf that contains the query point, so it is computed once per leaf here instead of per query.
inline bool layout_photons(const gi_photon_map_desc* d, HostPhotons& H, std::string& err)
{
    H.nodes.clear(); H.ranges.clear(); H.pos.clear(); H.dircol.clear(); H.n_node = 0; H.n_photon = 0; H.planes_ok = true;
    if (d->n_node <= 0 || d->n_photon <= 0) return true;
    if (!d->photons || !d->node_bbox || !d->node_child || !d->node_off) { err = "photons: null tables"; return false; }
    const int N = d->n_node;
    const int nref = d->node_off[N];
    if (d->node_off[0] != 0 || nref < 0 || (nref && !d->node_idx)) { err = "photons: bad leaf table"; return false; }
    for (int r = 0; r < nref; r++)
        if (d->node_idx[r] < 0 || d->node_idx[r] >= d->n_photon) { err = "photons: leaf reference out of range"; return false; }
    std::vector<int32_t> skip((size_t)N, 0);
    for (int n = N - 1; n >= 0; n--) {   // sub-tree end = max over children, children come later in pre-order
        int32_t end = n + 1;
        for (int k = 0; k < 8; k++) {
            int ch = d->node_child[(size_t)n * 8 + k];
            if (ch == -1) continue;
            if (ch <= n || ch >= N) { err = "photons: child index is not a later pre-order node"; return false; }
            end = std::max(end, skip[ch]);
        }
        skip[n] = end;
    }
    for (int n = 0; n < N; n++) {
        int32_t expect = n + 1;
        const bool leaf = d->node_child[(size_t)n * 8] == -1;
        if (d->node_off[n + 1] < d->node_off[n]) { err = "photons: leaf offsets not monotone"; return false; }
        for (int k = 0; k < 8; k++) {
            int ch = d->node_child[(size_t)n * 8 + k];
            if (leaf) { if (ch != -1) { err = "photons: node with partial children"; return false; } continue; }
            if (ch != expect) { err = "photons: nodes are not in pre-order"; return false; }
            expect = skip[ch];
        }
    }
    // (1) photons in leaf (DFS) order
    H.pos.resize((size_t)nref * 3);
    H.dircol.resize((size_t)nref * 6);
    for (int r = 0; r < nref; r++) {
        const double* p = d->photons + (size_t)d->node_idx[r] * 9;
        for (int k = 0; k < 3; k++) H.pos[(size_t)r * 3 + k] = p[k];
        for (int k = 0; k < 6; k++) H.dircol[(size_t)r * 6 + k] = p[3 + k];
    }
    // (2) records with consecutive children: breadth-first numbering
    std::vector<int32_t> rec_of((size_t)N, -1), order;
    order.reserve((size_t)N);
    rec_of[0] = 0;
    order.push_back(0);
    for (size_t head = 0; head < order.size(); head++) {
        const int n = order[head];
        if (d->node_child[(size_t)n * 8] == -1) continue;
        for (int k = 0; k < 8; k++) {
            const int ch = d->node_child[(size_t)n * 8 + k];
            rec_of[ch] = (int32_t)order.size();
            order.push_back(ch);
        }
    }
    if ((int)order.size() != N) { err = "photons: tree is not connected"; return false; }
    H.nodes.assign((size_t)N, PNode());
    auto box_of = [&](int n, double* lo, double* hi) { for (int k = 0; k < 3; k++) { lo[k] = d->node_bbox[(size_t)n * 6 + k]; hi[k] = d->node_bbox[(size_t)n * 6 + 3 + k]; } };
    // (3) candidate ranges per leaf: the reference's get() on the canonical tree, sub-trees skipped with the pre-order links
    for (int n = 0; n < N; n++) {
        PNode& t = H.nodes[(size_t)rec_of[n]];
        memset(&t, 0, sizeof t);
        box_of(n, t.bmin, t.bmax);
        const bool leaf = d->node_child[(size_t)n * 8] == -1;
        t.first_child = leaf ? -1 : rec_of[d->node_child[(size_t)n * 8]];
        if (!leaf) {
            const int c7 = d->node_child[(size_t)n * 8 + 7];
            for (int k = 0; k < 3; k++) t.mid[k] = d->node_bbox[(size_t)c7 * 6 + k];
            // the planes the 8 children's boxes are made of (include/photonMap.cpp:139-149); H.planes_ok stays true only while every
            // child's stored box equals, bit for bit, the box gather_find_leaf derives from them
            for (int ax = 0; ax < 3; ax++) {
                const int bitpos = ax == 0 ? 0 : (ax == 1 ? 2 : 1);   // x = bit 0, z = bit 1, y = bit 2
                const int hs = 1 << bitpos;                           // a child on the high side of this axis only
                const int ch = d->node_child[(size_t)n * 8 + hs];
                t.u.in.lo2[ax] = d->node_bbox[(size_t)ch * 6 + ax]; t.u.in.hi2[ax] = d->node_bbox[(size_t)ch * 6 + 3 + ax];
            }
            for (int c = 0; c < 8; c++) {
                const int ch = d->node_child[(size_t)n * 8 + c];
                for (int ax = 0; ax < 3; ax++) {
                    const int bitpos = ax == 0 ? 0 : (ax == 1 ? 2 : 1);
                    const double clo = d->node_bbox[(size_t)ch * 6 + ax], chi = d->node_bbox[(size_t)ch * 6 + 3 + ax];
                    const double wlo = c == 7 ? t.mid[ax] : (((c >> bitpos) & 1) ? t.u.in.lo2[ax] : t.bmin[ax]);
                    const double whi = c == 7 ? t.bmax[ax] : (((c >> bitpos) & 1) ? t.u.in.hi2[ax] : t.mid[ax]);
                    if (clo != wlo || chi != whi) H.planes_ok = false;
                }
            }
            continue;
        }
        double qlo[3], qhi[3];
        for (int k = 0; k < 3; k++) { qlo[k] = t.bmin[k] - GI_EPSILON; qhi[k] = t.bmax[k] + GI_EPSILON; }
        t.nb_off = (int32_t)H.ranges.size();
        int total = 0;
        if (!(qhi[0] - qlo[0] <= 0)) {   // include/photonMap.cpp:73-74
            int m = 0;
            while (m < N) {
                bool take = true;
                if (m != 0) {   // the root itself is not tested (PhotonMap::getInRange calls _root.get directly)
                    double lo[3], hi[3];
                    box_of(m, lo, hi);
                    take = (lo[0] <= qhi[0] && hi[0] >= qlo[0]) && (lo[1] <= qhi[1] && hi[1] >= qlo[1]) && (lo[2] <= qhi[2] && hi[2] >= qlo[2]);   // include/bbox.h:33-38
                }
                if (!take) { m = skip[m]; continue; }
                if (d->node_child[(size_t)m * 8] != -1) { m = m + 1; continue; }
                const int first = d->node_off[m], count = d->node_off[m + 1] - d->node_off[m];
                if (count > 0) {
                    if ((int32_t)H.ranges.size() > t.nb_off && H.ranges.back().first + H.ranges.back().count == first) H.ranges.back().count += count;
                    else { PRange rg; rg.first = first; rg.count = count; H.ranges.push_back(rg); }
                    total += count;
                }
                m = skip[m];
            }
        }
        t.u.lf.nb_cnt = (int32_t)H.ranges.size() - t.nb_off;
        t.u.lf.nb_photons = total;
    }
    if (H.ranges.empty()) { PRange z; z.first = 0; z.count = 0; H.ranges.push_back(z); }
    H.n_node = N; H.n_photon = nref;
    return true;
}

inline int local_rows(const gi_render_params* p)
{
    if (!p || p->stripe_h <= 0 || p->stripe_world <= 0 || p->height <= 0) return 0;
    int rows = 0;
    const int n_stripes = (p->height + p->stripe_h - 1) / p->stripe_h;
    for (int k = p->stripe_rank; k < n_stripes; k += p->stripe_world) rows += std::min(p->stripe_h, p->height - k * p->stripe_h);
    return rows;
}

// camera basis and frame constants, include/raytracer.h:74-78
inline bool make_frame(const gi_render_params* p, Frame& F, std::string& err)
{
    if (!p || p->width <= 0 || p->height <= 0 || p->stripe_h <= 0 || p->stripe_world <= 0 || p->stripe_rank < 0 || p->stripe_rank >= p->stripe_world ||
        p->min_samples < 0 || p->max_samples < 0) { err = "render: bad parameters"; return false; }
    V3 pos = ld3(p->cam_pos), up = ld3(p->cam_up), fwd = ld3(p->cam_forward);
    const int w = p->width, h = p->height;
    F.sw = (p->sensor_diag * w) / (std::sqrt((double)w * w + h * h));
    F.sh = F.sw * ((double)h / w);
    F.screen_center = pos + p->focal_dist * fwd;
    F.right = normalize(cross(fwd, up));
    F.cam_pos = pos; F.cam_up = up;
    F.w = w; F.h = h;
    F.he = make_halton_enum((unsigned)w, (unsigned)h);
    F.min_samples = p->min_samples; F.max_samples = p->max_samples; F.noise_thresh = p->noise_thresh;
    F.seed = p->seed;
    F.stripe_h = p->stripe_h; F.stripe_rank = p->stripe_rank; F.stripe_world = p->stripe_world;
    F.local_rows = local_rows(p);
    return true;
}

}  // namespace gi
