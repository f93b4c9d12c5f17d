// gi_device.h -- per-lane device functions of the MI355X render hot path (one ray / one pixel per lane).
//
// Everything here is plain scalar-per-lane code marked GI_HD so that the very same functions can also be compiled
// for the host by tests/host_emul (a CPU build used ONLY by unit tests and sanitizers to check the kernel logic
// without a GPU; the product library never contains it).  Wave-level code (ballots, atomics, LDS carving, launches)
// lives in gi_kernels.hip.
//
// Reference semantics followed (file:line relative to moepforfreedom/GI_Raytracer):
//   trace            include/raytracer.h:382-478 + include/octree.cpp:188-211,285-313 + include/bbox.h:47-73
//   visible          include/raytracer.h:280-319 + include/octree.cpp:150-185,256-282 + include/bbox.h:117-138
//   triangle test    include/entities.h:443-490
//   secondaryRay     include/raytracer.h:321-379, rayType :481-506
//   radiance         include/raytracer.h:167-276 (recursion turned into a loop carrying the path throughput)
//   samplePhotons    include/raytracer.h:532-579 + include/photonMap.cpp:50-134
//   tracePhotons     include/raytracer.h:582-715
//   samplers         include/util.cpp:27-107, include/util.h:100-188
//   Halton           include/halton_enum.h:106-155, include/halton_sampler.h:626-888,1417-3286
// Arithmetic keeps the reference's operand order (glm 0.9.8 conventions) and is compiled with -ffp-contract=off so that
// hit/miss and traversal decisions agree with the CPU path; see DESIGN.md "Numerics".
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef GI_HD
#define GI_HD __host__ __device__ __forceinline__
#endif
// keep a value alive without using it (a load issued early only to pull its cache line in)
#if defined(__HIP_DEVICE_COMPILE__)
#define GI_TOUCH(x) asm volatile("" ::"v"(x))
#else
#define GI_TOUCH(x) (void)(x)
#endif
#ifndef GI_HDM   // member functions
#define GI_HDM __host__ __device__ __forceinline__
#endif

namespace gi {

// ------------------------------------------------------------------------------------------------ constants (include/util.h:14-31)
#if defined(__clang__)
#define GI_UNROLL _Pragma("unroll")
#else
#define GI_UNROLL
#endif
#define GI_EPSILON 0.00001
#define GI_SHADOW_BIAS 0.0001
#define GI_MIN_DEPTH 2
#define GI_MAX_DEPTH 64
#define GI_PI 3.14159265358979323846
#define GI_GATHER_K 32
#define GI_RAYMARCH_STEPSIZE 0.04
#define GI_FEAT_SPHERES 1   // template feature bits: code for entity kinds / media a scene does not contain is not compiled in
#define GI_FEAT_FOG 2
#define GI_FEAT_TEX 4       // checkerboard / image textures (include/material.h:32-81): uv carried through the walk, texture alpha in the alpha test

// ------------------------------------------------------------------------------------------------ device tables (HBM layout)
struct NodeLink { int32_t hit, skip; };
struct alignas(128) TNode { // 128 B = one L2 line: a scene-octree node, stored once, nodes in breadth-first order (top levels first,
                            // so that a prefix of the array is what a kernel stages in LDS)
    double bmin[3], bmax[3];
    int32_t first_ref;      // leaves: first entry in leaf_refs
    int32_t n_ref;          // leaves: number of entries; inner nodes: -1
    int32_t leaf_id;        // canonical pre-order index (RNG key, same numbering as the CPU side)
    int32_t pad;
    NodeLink link[8];       // per direction octant: next node when the box is hit (inner nodes) / when the sub-tree is left;
                            // following them visits the children of every node front to back for that octant; END = n_node
};
// Wide node: one record per INNER node holding what the box tests of all its (up to 8) children need.  Octree::Node::partition
// (include/octree.cpp:318-328) builds the children's boxes from five planes per axis: children on the low side span [min, mid],
// children on the high side [min + .5 d, mid + .5 d], except child 7 which spans [mid, max] (mid = mix(min, max, .5); the three
// forms of "the middle" and of "the far side" differ in the last bit).  The record keeps those planes exactly as the children's
// boxes hold them (checked bit for bit when the scene is laid out; a tree of another shape keeps the per-node walk), so one record
// and 36 multiply-subtracts replace eight records and 96.  pl[axis] = {min, mid, lo2, hi2, max, mid}: the pairs (0,1), (2,3),
// (5,4) are the low-side, high-side and child-7 intervals, and a ray that runs backwards along the axis reads every pair in the
// opposite order (entry plane first) by xor-ing 8 into the byte offset -- the swap of include/bbox.h:52-57 costs no instruction.
struct alignas(32) WNode {  // 224 B
    double pl[3][6];        // x, y, z
    int32_t ca[8], cb[8];   // child slot c (x = bit 0, z = bit 1, y = bit 2): cb < 0: inner, wide record ca; cb > 0: leaf, its references
                            // are leaf_tris[ca .. ca + cb); cb == 0: no such child (or an empty leaf)
    int32_t parent;         // wide record of the parent, -1 at the root
    uint32_t exists;        // bit c: child slot c is an inner node or a non-empty leaf
    int32_t pad[2];
};
struct TriGeom {            // 80 B: what a ray-triangle test needs
    double p0[3], e1[3], e2[3];
    int32_t mat;
    uint32_t flags;         // bit0: interpolate vertex normals; bit1: alpha test always passes (opacity >= 1 or IOR != 1);
                            // bit2: analytic sphere (p0 = centre, e1[0] = radius)
};
struct LeafTri {            // 80 B: the test record again, stored once per leaf reference in leaf order, so that a leaf's triangles
    double p0[3], e1[3], e2[3];   // are consecutive in memory and no index has to be chased before the vertex data can be fetched
    int32_t tri;            // triangle index (shading record, RNG key)
    uint32_t matflags;      // material << 3 | flags
};
struct TriShade { double n0[3], n1[3], n2[3], fnorm[3]; };  // 96 B, read once per shaded hit
struct TriUV { double t0[2], t1[2], t2[2]; };                // 48 B: vertex texCoords, read only by textured scenes (GI_FEAT_TEX)
struct Mat { double roughness, opacity, ior, diffuse[3], emissive[3]; int32_t dtex, etex; };   // dtex / etex: texture record or -1 = the constant colour
struct TexD {               // include/material.h:10-81
    int32_t kind;           // 0 texture(col), 1 checkerboard, 2 imageTexture
    int32_t w, h, has_alpha;
    double a[3], b[3];      // colour; checkerboard colours a, b
    double tiles, tile_u, tile_v;
    unsigned long long pix; // first byte of the image in Scene::tex_pixels (RGBA8, rows top to bottom)
};
struct LightD { double pos[3], col[3], rad, dir[3], angle; };
struct alignas(128) PNode { // 128 B: photon octree node; the 8 children of a node are 8 consecutive records
    double bmin[3], bmax[3];// node box: PhotonMap::Node::getBounds checks contains() at every level
    double mid[3];          // inner nodes: lower corner of child 7 = the split point the reference computes
    int32_t first_child;    // record index of child 0, or -1 for a leaf
    int32_t nb_off;         // leaves: candidate photon ranges [nb_off, nb_off + nb_cnt) in Scene::pranges
    union {
        struct { int32_t nb_cnt, nb_photons, pad[10]; } lf;   // leaves: nb_photons = total photons in those ranges (= what getInRange returns here)
        struct { double lo2[3], hi2[3]; } in;                 // inner nodes: the high-side children span [lo2, hi2] (= [min + .5 d, mid + .5 d],
    } u;                                                      // include/photonMap.cpp:139-149), child 7 [mid, max], low-side children [min, mid]
};
struct PDescent { double mid[3]; int32_t first_child, pad; };   // what the descent of getBounds decides by; the boxes it checks stay in PNode
struct PRange { int32_t first, count; };   // photons are stored leaf by leaf in the reference's DFS order
struct HaltonDim { uint32_t P, n, off; float scale; uint32_t mlo, mhi, pad[2]; };   // (mhi:mlo) = floor((2^64 - 1) / P) + 1: x / P = hi64(x * m) for every 32-bit x
struct FogD { double pos[3], size[3], col[3], d, sc, bmin[3], bmax[3]; int32_t grid_off, grid_n; };   // HeightFog, include/atmosphere.h:30-83

struct Scene {
    const TNode* tnodes;      // [n_node]
    const WNode* wnodes;      // [n_wnode] inner nodes, breadth first; null when the tree is not made of exact octants
    const int32_t* wleaf_id;  // [n_wnode][8] canonical node index of a leaf child (RNG key of the alpha test)
    const float* cboxes;      // [n_wnode][8][6] content box of every child's sub-tree (all entities referenced below it), rounded outwards; null = no culling
    const uint32_t* cuse;     // [n_wnode] children (slot bits) whose content box is worth testing (clearly smaller than their octant)
    const double* shadow_boxes;   // boxes for segments that END AT A LIGHT (k_st_shadow): trace_boxes when no entity comes near a light (visible_leaf_blocks), else leaf_boxes
    const float* scboxes;     // and the content boxes that go with them (= tcboxes / tcuse, or cboxes / cuse)
    const uint32_t* scuse;
    const float* tcboxes;     // the same two tables made of trace_boxes, for the streaming closest-hit walk (k_st_trace); = cboxes / cuse where nothing is cut to its leaf
    const uint32_t* tcuse;
    const int32_t* leaf_refs;
    const LeafTri* leaf_tris; // [n_refs], parallel to leaf_refs
    const double* leaf_boxes; // [n_refs][6] every reference's own box (min xyz, max xyz, widened): a ray that misses it cannot hit the entity; null = not used
    const double* trace_boxes;// [n_refs][6] the same for the closest-hit walk: where the rules of trace_wide_step allow it, the part of the entity inside its leaf; null with leaf_boxes
    const TriGeom* tris;
    const TriShade* shade;
    const TriUV* tri_uv;      // [n_tri], only when n_tex > 0
    const Mat* mats;
    const LightD* lights;
    const PNode* pnodes;
    const struct PDescent* pdescent;   // [n_pnode] split point and first child only (32 B: the whole table stays in L2), for gather_find_leaf_fast; null = not used
    const int32_t* pleaf_rank;   // [n_pnode] rank of a leaf among the leaves with candidate photons (the gather queries' sort key), -1 for every other node
    const int32_t* prank_leaf;   // [n_pleaf] its inverse
    int32_t n_pleaf;             // leaves with candidates
    const uint32_t* pcand_off;   // [n_pleaf + 1] where the candidate list of the leaf of that rank begins in pcand; null = not used
    const double* pcand;         // [total][3] the candidates' POSITIONS: the ranges of every such leaf written out back to back, in the order gather_in_leaf visits them (k_st_gather)
    const double* pcand_dc;      // [total][6] and their direction and colour (read by the second pass)
    const PRange* pranges;
    const double* ph_pos;     // [n_photon][3] leaf order
    const double* ph_dircol;  // [n_photon][6] leaf order
    const FogD* fogs;
    const double* fog_grid;
    const TexD* texs;
    const unsigned char* tex_pixels;
    const double* tex_lut;    // [256] pow(k / 255.0, 1.0 / (1.0 / GAMMA)) as the host libm computes it = imageTexture::get's gamma()
    int32_t n_tex;            // > 0: the scene has a non-constant texture (kernels instantiated with GI_FEAT_TEX)
    const HaltonDim* hdims;   // [256]
    const uint16_t* htable;
    int32_t n_node, n_tri, n_light, n_pnode, n_photon;
    int32_t n_wnode;
    int32_t pn_planes;        // photon octree: every inner record carries its children's planes (gather_find_leaf)
    int32_t has_spheres;      // 0: triangles only
    int32_t n_fog;
    double ambient[3];
    double root_bmin[3], root_bmax[3];   // box of octree node 0 (kernel argument: no memory round trip before a walk starts)
    double pmap_bmin[3], pmap_bmax[3];   // box of the photon map's root (gather_find_leaf_fast)
    const int32_t* pjump;     // [N][N][N] (y, z, x; N = GI_PJUMP_N) the node the descent of a position in that cell of the map's box has reached after GI_PJUMP_BITS levels (or its leaf); null = not used
    double pjump_cell[3], pjump_inv[3];   // a cell's size per axis and its inverse
    double cut_margin;        // >= 0: a closest-hit walk looks no further than its best hit plus this (trace_wide_step); < 0: it walks on as the reference does
};

#ifndef GI_PJUMP_BITS
#define GI_PJUMP_BITS 7          // levels of the photon octree the jump table of gather_find_leaf_fast covers: a grid of 128^3 cells, 8 MB (5: 22.3 ms of queue compaction + gather keys on the benchmark, 6: 21.1, 7: 19.9)
#endif
#define GI_PJUMP_N (1 << GI_PJUMP_BITS)
struct Counters { unsigned long long v_trace, v_shadow, tri, shaded, pcand, traces, shadows, gathers; };

// ------------------------------------------------------------------------------------------------ vec3 in glm operand order
struct V3 { double x, y, z; };
GI_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
GI_HD V3 ld3(const double* p) { return v3(p[0], p[1], p[2]); }
GI_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
GI_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
GI_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
GI_HD V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
GI_HD V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
GI_HD V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
GI_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GI_HD V3 cross(V3 x, V3 y) { return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
GI_HD V3 normalize(V3 v) { return v * (1.0 / sqrt(dot(v, v))); }
GI_HD double length(V3 v) { return sqrt(dot(v, v)); }
GI_HD V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0; }
GI_HD V3 mix(V3 x, V3 y, double a) { return x + a * (y - x); }
GI_HD double len2(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
GI_HD double comp_max(V3 v) { return fmax(fmax(v.x, v.y), v.z); }

struct Ray { V3 o, d, inv; };
GI_HD Ray make_ray(V3 o, V3 d)  // Ray ctor / setDir, include/ray.h:7-17
{
    Ray r;
    r.o = o;
    r.d = normalize(d);
    r.inv = v3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    return r;
}
GI_HD Ray make_ray_exact(V3 o, V3 d)
{
    Ray r;
    r.o = o; r.d = d;
    r.inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    return r;
}

// ------------------------------------------------------------------------------------------------ counter RNG (DESIGN.md "RNG contract")
GI_HD uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27; z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
enum {
    P_TRACE_ALPHA = 0, P_SHADOW_ALPHA = 1, P_LIGHT_X = 2, P_LIGHT_Y = 3, P_TYPE_OPACITY = 4, P_TYPE_FRESNEL = 5,
    P_RR = 6, P_FOG = 7, P_TRACE_GUARD = 8,
    P_PH_DIR_U = 16, P_PH_DIR_V = 17, P_PH_SEC_U = 18, P_PH_SEC_V = 19, P_PH_TRACE0_ALPHA = 20, P_PH_FOG_U = 21, P_PH_FOG_V = 22
};
enum { P_FOG_CAMERA = 0, P_FOG_SHADOW = 1, P_FOG_PHOTON = 2 };   // `b` key of P_FOG draws (shadow: + 16 * light index); `a` = step
#define GI_PHOTON_SEED_XOR 0x5048544f4e5eed00ull
struct Rng {
    uint64_t hs;      // hash of (seed, stream)
    uint32_t depth;
};
GI_HD Rng rng_make(uint64_t seed, uint32_t stream)
{
    Rng r;
    r.hs = mix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)stream + 1));
    r.depth = 0;
    return r;
}
GI_HD double rng_draw(const Rng& r, uint32_t purpose, uint32_t a = 0, uint32_t b = 0)
{
    uint64_t h = mix64(r.hs ^ (((uint64_t)r.depth << 32) | purpose));
    h = mix64(h ^ (((uint64_t)a << 32) | b));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------------------------------------ Halton
GI_HD uint32_t rev_bits32(uint32_t index)
{
    index = (index << 16) | (index >> 16);
    index = ((index & 0x00ff00ffu) << 8) | ((index & 0xff00ff00u) >> 8);
    index = ((index & 0x0f0f0f0fu) << 4) | ((index & 0xf0f0f0f0u) >> 4);
    index = ((index & 0x33333333u) << 2) | ((index & 0xccccccccu) >> 2);
    index = ((index & 0x55555555u) << 1) | ((index & 0xaaaaaaaau) >> 1);
    return index;
}
// Halton_sampler::sample: base 2 by bit reversal into the mantissa, other bases by table digit groups (Horner form of the
// reference's sum_k table[(index / P^k) % P] * P^(n-1-k)), then one float multiply.
GI_HD float halton_sample(const Scene& S, uint32_t dim, uint32_t index)
{
    if (dim == 0) {
        union { uint32_t u; float f; } r;
        r.u = 0x3f800000u | (rev_bits32(index) >> 9);
        return r.f - 1.f;
    }
    const HaltonDim D = S.hdims[dim];
    // digit groups first (division by the table size as a multiplication: exact for every 32-bit index and P <= 1619), then all table
    // reads at once -- they do not depend on each other, and up to 7 dependent L2 round trips per sample were what this cost
    uint32_t dg[7], idx = index;
    GI_UNROLL
    for (int k = 0; k < 7; k++) {
        unsigned long long t = ((unsigned long long)idx * D.mlo) >> 32;
        t += (unsigned long long)idx * D.mhi;
        const uint32_t q = (uint32_t)(t >> 32);
        dg[k] = idx - q * D.P;
        idx = q;
    }
    uint32_t tv[7];
    GI_UNROLL
    for (int k = 0; k < 7; k++) tv[k] = (uint32_t)k < D.n ? (uint32_t)S.htable[D.off + dg[k]] : 0u;
    uint32_t sum = 0;
    GI_UNROLL
    for (int k = 0; k < 7; k++) if ((uint32_t)k < D.n) sum = sum * D.P + tv[k];
    return (float)sum * D.scale;
}
struct HaltonEnumD { uint32_t p2, p3, m_x, m_y, inc; float scale_x, scale_y; };
// Halton_enum::get_index, include/halton_enum.h:106-114,136-155
GI_HD uint32_t halton_index(const HaltonEnumD& e, uint32_t i, uint32_t x, uint32_t y)
{
    const unsigned long long hx = e.p2 ? (rev_bits32(x) >> (32 - e.p2)) : 0u;
    uint32_t r3 = 0, yy = y;
    for (uint32_t d = 0; d < e.p3; ++d) { r3 = r3 * 3 + yy % 3; yy /= 3; }
    const unsigned long long hy = r3;
    const uint32_t offset = (uint32_t)((hx * e.m_x + hy * e.m_y) % e.inc);
    return offset + i * e.inc;
}

// ------------------------------------------------------------------------------------------------ samplers (include/util.h, util.cpp)
GI_HD double fast_precise_pow(double a, double b)  // include/util.h:113-136: bit hack on the high word + exact integer power
{
    int e = (int)b;
    union { double d; int32_t x[2]; } u;
    u.d = a;
    u.x[1] = (int32_t)((b - e) * (u.x[1] - 1072632447) + 1072632447);
    u.x[0] = 0;
    double r = 1.0;
    while (e) {
        if (e & 1) r *= a;
        a *= a;
        e >>= 1;
    }
    return r * u.d;
}
GI_HD V3 frame_mul(V3 n, double z, V3 r)  // the explicit 3x3 of include/util.cpp:37-42 applied to r (glm column-major)
{
    double k = (1.0 / (1 + z));
    V3 c0 = v3(z + k * -n.y * -n.y, k * (n.x * -n.y), -n.x);
    V3 c1 = v3(k * (n.x * -n.y), z + k * -n.x * -n.x, -n.y);
    V3 c2 = v3(n.x, n.y, z);
    return v3(c0.x * r.x + c1.x * r.y + c2.x * r.z, c0.y * r.x + c1.y * r.y + c2.y * r.z, c0.z * r.x + c1.z * r.y + c2.z * r.z);
}
GI_HD V3 lobe_local(float u, float v, double power, double frac, bool cap)
{
    float phi = (float)(v * 2.0f * GI_PI);
    float cosTheta = cap ? (float)(frac * fast_precise_pow(1.0f - u, (1.0f / power)) + (1 - frac)) : (float)fast_precise_pow(1.0f - u, (1.0f / power));
    float sinTheta = (float)sqrt((double)(1.0f - cosTheta * cosTheta));
    return v3(cos((double)phi) * sinTheta, sin((double)phi) * sinTheta, cosTheta);
}
GI_HD V3 hemi_cos_n(V3 n, float u, float v, double power)  // include/util.cpp:35-58
{
    V3 res = frame_mul(n, fabs(n.z), lobe_local(u, v, power, 0, false));
    if (n.z < 0) res.z *= -1.0;
    return res;
}
GI_HD V3 sphere_cap_cos(V3 n, float u, float v, double power, double frac)  // include/util.cpp:60-83
{
    V3 res = frame_mul(n, fabs(n.z), lobe_local(u, v, power, frac, true));
    if (n.z < 0) res.z *= -1.0;
    return res;
}
GI_HD V3 sample_phong(V3 outdir, double power, double sx, double sy)  // include/util.cpp:91-107
{
    V3 out = frame_mul(outdir, fabs(outdir.z), lobe_local((float)sx, (float)sy, power, 0, false));
    if (outdir.z < 0) out.z *= -1.0;
    return out;
}
GI_HD V3 random_unit_vec(double x, double y)  // include/util.h:183-188
{
    double theta = acos(2 * y - 1);
    return v3(sin(theta) * cos(2 * x * GI_PI), sin(theta) * sin(2 * x * GI_PI), cos(theta));
}
GI_HD V3 refr(V3 inc, V3 norm, double eta)  // include/util.h:173-181
{
    double d = dot(norm, inc);
    double k = 1.0 - eta * eta * (1.0 - d * d);
    if (k < GI_EPSILON) return reflect(inc, norm);
    return eta * inc - (eta * d + sqrt(k)) * norm;
}

// ------------------------------------------------------------------------------------------------ boxes
// BoundingBox::intersect(ray, 0, inf) reduced to the hit flag (t0 is only used for ordering, which the direction-ordered
// tree provides) -- same slab arithmetic and early-outs as include/bbox.h:47-73 / :117-138.
GI_HD bool box_hit(const double* bmin, const double* bmax, const Ray& r, double tmin, double tmax)
{
    {
        double t0 = (bmin[0] - r.o.x) * r.inv.x, t1 = (bmax[0] - r.o.x) * r.inv.x;
        if (r.inv.x < 0.0) { double t = t0; t0 = t1; t1 = t; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    {
        double t0 = (bmin[1] - r.o.y) * r.inv.y, t1 = (bmax[1] - r.o.y) * r.inv.y;
        if (r.inv.y < 0.0) { double t = t0; t0 = t1; t1 = t; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    {
        double t0 = (bmin[2] - r.o.z) * r.inv.z, t1 = (bmax[2] - r.o.z) * r.inv.z;
        if (r.inv.z < 0.0) { double t = t0; t0 = t1; t1 = t; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    return true;
}
GI_HD bool box_contains(const double* bmin, const double* bmax, V3 p)  // include/bbox.h:41-44 (half open)
{
    return p.x >= bmin[0] && p.y >= bmin[1] && p.z >= bmin[2] && p.x < bmax[0] && p.y < bmax[1] && p.z < bmax[2];
}
// BoundingBox::intersect(ray, tmin, tmax, toutmin, toutmax), include/bbox.h:47-73
GI_HD bool box_range(const double* bmin, const double* bmax, const Ray& r, double tmin, double tmax, double& t0o, double& t1o)
{
    const double o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
    for (int i = 0; i < 3; i++) {
        double t0 = (bmin[i] - o[i]) * inv[i], t1 = (bmax[i] - o[i]) * inv[i];
        if (inv[i] < 0.0) { double t = t0; t0 = t1; t1 = t; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    t0o = tmin; t1o = tmax;
    return true;
}
GI_HD double fast_pow(double a, double b)   // include/util.h:100-111
{
    union { double d; int32_t x[2]; } u;
    u.d = a;
    u.x[1] = (int32_t)(b * (u.x[1] - 1072632447) + 1072632447);
    u.x[0] = 0;
    return u.d;
}
// HeightFog::density, include/atmosphere.h:50-81 (nscale is 1 after construction; indices are computed in double as there)
GI_HD double fog_density(const Scene& S, const FogD& f, V3 p)
{
    const double ymax = f.pos[1] + .5 * f.size[1];
    const V3 rel = v3(p.x - f.bmin[0], p.y - f.bmin[1], p.z - f.bmin[2]);
    const int rx = (int)rel.x, ry = (int)rel.y, rz = (int)rel.z;
    const double dx = (rel.x - rx), dy = (rel.y - ry), dz = (rel.z - rz);
    const double* G = S.fog_grid + f.grid_off;
    const int last = f.grid_n - 1;
#define GI_FOG_G(expr) G[((int)(expr)) < last ? (((int)(expr)) < 0 ? 0 : (int)(expr)) : last]
    const double sx = f.size[0], sz = f.size[2];
    double c00 = (1 - dx) * GI_FOG_G((rx * sx + ry) * sz + rz) + dx * GI_FOG_G(((rx + 1) * sx + ry) * sz + rz);
    double c01 = (1 - dx) * GI_FOG_G((rx * sx + ry) * sz + rz + 1) + dx * GI_FOG_G(((rx + 1) * sx + ry) * sz + rz + 1);
    double c10 = (1 - dx) * GI_FOG_G((rx * sx + (ry + 1)) * sz + rz) + dx * GI_FOG_G(((rx + 1) * sx + (ry + 1)) * sz + rz);
    double c11 = (1 - dx) * GI_FOG_G((rx * sx + (ry + 1)) * sz + rz + 1) + dx * GI_FOG_G(((rx + 1) * sx + (ry + 1)) * sz + rz + 1);
#undef GI_FOG_G
    double c0 = c00 * (1 - dy) + c10 * dy;
    double c1 = c01 * (1 - dy) + c11 * dy;
    double noise = fast_pow((1 - dz) * c0 + dz * c1, 7);
    return f.d * noise * fast_pow((ymax - p.y) / f.size[1], 2);
}
GI_HD double atmosphere_density(const Scene& S, V3 pos, V3& col)   // Octree::atmosphereDensity, include/octree.cpp:214-226
{
    double d = 0;
    for (int i = 0; i < S.n_fog; i++) {
        const FogD& f = S.fogs[i];
        if (box_contains(f.bmin, f.bmax, pos)) { col = ld3(f.col); d += GI_RAYMARCH_STEPSIZE * fog_density(S, f, pos); }
    }
    return d;
}
GI_HD bool atmosphere_bounds(const Scene& S, const Ray& r, double& mint, double& maxt)   // Octree::atmosphereBounds, include/octree.cpp:229-251
{
    double mn = 0, mx = 0;
    bool intersected = false;
    for (int i = 0; i < S.n_fog; i++) {
        double a, b;
        if (box_range(S.fogs[i].bmin, S.fogs[i].bmax, r, mint, maxt, a, b)) { mn = a < mn ? a : mn; mx = b > mx ? b : mx; intersected = true; }
    }
    mint = mint > mn ? mint : mn;
    maxt = maxt < mx ? maxt : mx;
    return intersected;
}
// RayTracer::raymarch, include/raytracer.h:509-529: absorb with probability density*step per 0.04 step
GI_HD bool raymarch(const Scene& S, const Ray& r, V3& hit, V3& col, double mint, double maxt, const Rng& rng, uint32_t which)
{
    double t = mint + GI_SHADOW_BIAS;
    V3 current = r.o + mint * r.d;
    uint32_t step = 0;
    while (t < maxt) {
        if (rng_draw(rng, P_FOG, step, which) < atmosphere_density(S, current, col)) { hit = current; return true; }
        current = current + GI_RAYMARCH_STEPSIZE * r.d;
        t += GI_RAYMARCH_STEPSIZE;
        step++;
    }
    return false;
}
// child visiting order: the reference numbers children x = bit0, z = bit1, y = bit2 (include/octree.cpp:321-328)
GI_HD int dir_octant(const Ray& r) { return (r.d.x < 0.0 ? 1 : 0) | (r.d.z < 0.0 ? 2 : 0) | (r.d.y < 0.0 ? 4 : 0); }

// ------------------------------------------------------------------------------------------------ triangle (include/entities.h:443-490)
template <class Tri>
GI_HD bool tri_hit(const Tri& g, const Ray& ray, double& u, double& v, double& t)
{
    V3 edge1 = ld3(g.e1), edge2 = ld3(g.e2);
    V3 p = cross(ray.d, edge2);
    double det = dot(edge1, p);
    if (det < GI_EPSILON && det > -GI_EPSILON) return false;
    double inv_det = 1.0 / det;
    V3 tvec = ray.o - ld3(g.p0);
    u = dot(tvec, p) * inv_det;
    if (u < 0 || u > 1) return false;
    V3 q = cross(tvec, edge1);
    v = dot(ray.d, q) * inv_det;
    if (v < 0 || u + v > 1) return false;
    t = dot(edge2, q) * inv_det;
    if (t <= 0) return false;
    return true;
}

struct HitRec { V3 pos; double u, v; int32_t tri; uint32_t mf; double tu, tv; };   // mf = LeafTri::matflags of the hit (material << 3 | flags)   // u, v barycentric; tu, tv = the reference's `uv` (GI_FEAT_TEX only)

// Wave-uniform leaves.  When every active lane of a wave stands on the SAME leaf (primary rays of one 8x8 tile, shadow rays towards one
// light) the leaf's records are fetched once per wave through the scalar cache (s_load into SGPRs -- a load from the constant address
// space at a wave-uniform address) instead of once per lane through the vector memory path: the 64 identical copies of an 80-byte
// record were what kept the L1 -> VGPR path as busy as the VALU.  The tests then read the record as SGPR operands; same arithmetic,
// same order, same bits.  Device code only (the CPU build of these functions has no waves).
#if defined(__HIP_DEVICE_COMPILE__)
#define GI_WAVE_UNIFORM_LEAVES 1
GI_HD bool leaf_is_wave_uniform(int32_t first, int32_t cnt, int32_t& first_u, int32_t& cnt_u)
{
    first_u = __builtin_amdgcn_readfirstlane(first);
    cnt_u = __builtin_amdgcn_readfirstlane(cnt);
    return __ballot(first != first_u || cnt != cnt_u) == 0ull;
}
GI_HD LeafTri leaf_tri_scalar(const LeafTri* rec)   // rec is wave-uniform
{
    typedef const __attribute__((address_space(4))) double* KD;
    typedef const __attribute__((address_space(4))) int32_t* KI;
    const KD d = (KD)(reinterpret_cast<const double*>(rec));
    const KI w = (KI)(reinterpret_cast<const int32_t*>(rec));
    LeafTri g;
    g.p0[0] = d[0]; g.p0[1] = d[1]; g.p0[2] = d[2];
    g.e1[0] = d[3]; g.e1[1] = d[4]; g.e1[2] = d[5];
    g.e2[0] = d[6]; g.e2[1] = d[7]; g.e2[2] = d[8];
    g.tri = w[18]; g.matflags = (uint32_t)w[19];
    return g;
}
struct Box6 { double b[6]; };
GI_HD Box6 leaf_box_scalar(const double* rec)   // rec is wave-uniform: the entity's box through the scalar cache
{
    typedef const __attribute__((address_space(4))) double* KD;
    const KD d = (KD)rec;
    Box6 r;
    for (int k = 0; k < 6; k++) r.b[k] = d[k];
    return r;
}
#endif

// ------------------------------------------------------------------------------------------------ textures (include/material.h:10-81)
GI_HD const unsigned char* tex_pixel(const Scene& S, const TexD& x, double tu, double tv)   // image.pixelColor(...), include/material.h:65
{
    const int px = abs((int)(tu * x.w * x.tile_u) % x.w);
    const int py = x.h - abs((int)(tv * x.h * x.tile_v) % x.h) - 1;
    return S.tex_pixels + x.pix + ((size_t)py * x.w + px) * 4;
}
GI_HD V3 tex_get(const Scene& S, int32_t t, V3 constant, double tu, double tv)   // texture::get
{
    if (t < 0) return constant;
    const TexD& x = S.texs[t];
    if (x.kind == 0) return ld3(x.a);
    if (x.kind == 1) {
        const int tiles = (int)x.tiles;
        if (((int)(tu * tiles) % 2 == 0) ^ ((int)(tv * tiles) % 2 == 0)) return ld3(x.a);
        return ld3(x.b);
    }
    const unsigned char* px = tex_pixel(S, x, tu, tv);
    return v3(S.tex_lut[px[0]], S.tex_lut[px[1]], S.tex_lut[px[2]]);
}
GI_HD double tex_alpha(const Scene& S, int32_t t, double tu, double tv)   // texture::getAlpha
{
    if (t < 0) return 1;
    const TexD& x = S.texs[t];
    if (x.kind != 2 || !x.has_alpha) return 1;
    return tex_pixel(S, x, tu, tv)[3] / 255.0;
}
// the reference's `uv` after a successful Entity::intersect: interpolated texCoords of a smooth triangle (include/entities.h:480-482),
// asin / atan2 of a sphere (include/entities.h:93-96); a flat-shaded triangle leaves the caller's variable as it was
template <class Tri>
GI_HD void ent_uv(const Scene& S, const Tri& g, uint32_t flags, int32_t ti, double u, double v, V3 hp, double& tu, double& tv)
{
    if (flags & 4u) {
        const V3 d = (ld3(g.p0) - hp) / g.e1[0];
        tv = .5 + asin(d.y) / GI_PI;
        tu = .5 + atan2(d.z, d.x) / (2 * GI_PI);
    } else if (flags & 1u) {
        const TriUV& sh = S.tri_uv[ti];
        const double w = (1 - u - v);
        tu = w * sh.t0[0] + u * sh.t1[0] + v * sh.t2[0];
        tv = w * sh.t0[1] + u * sh.t1[1] + v * sh.t2[1];
    }
}
// Material::getAlpha = opacity * diffuse->getAlpha(uv), include/material.h:90-93
GI_HD double mat_alpha(const Scene& S, const Mat& m, double tu, double tv) { return m.opacity * tex_alpha(S, m.dtex, tu, tv); }

// Entity::intersect for the two kinds on this path: triangle (include/entities.h:443-490, barycentric u, v) and analytic sphere
// (include/entities.h:60-101).  hp = hit point.
template <int FEAT, class Tri>
GI_HD bool ent_hit(const Tri& g, uint32_t flags, const Ray& ray, double& u, double& v, V3& hp)
{
    if (!(FEAT & GI_FEAT_SPHERES) || !(flags & 4u)) {   // no sphere bit: the scene holds triangles only (checked at upload)
        double t;
        if (!tri_hit(g, ray, u, v, t)) return false;
        hp = ray.o + t * ray.d;
        return true;
    }
    const V3 pos = ld3(g.p0);
    const double rad = g.e1[0];
    const V3 oc = ray.o - pos;
    const double dt = dot(ray.d, oc);
    const double r = (dt * dt - len2(oc) + rad * rad);
    if (r < 0) return false;
    const double sr = sqrt(r);
    const double t_1 = -1 * dt - sr, t_2 = -1 * dt + sr;
    if (t_1 < 0 && t_2 < 0) return false;
    if ((t_1 < t_2 && t_1 > 0) || t_2 < 0) hp = ray.o + ray.d * t_1;
    else hp = ray.o + ray.d * t_2;
    u = 0; v = 0;   // the reference derives texture coordinates here (asin / atan2); constant textures never read them
    return true;
}

// Where node records come from: plain global memory here; gi_kernels.hip adds a source that serves the first nodes from LDS.
struct NodeView { double bmin[3], bmax[3]; int32_t first_ref, n_ref, hit, skip; };
struct GlobalNodes {
    static constexpr bool kWide = false;
    const TNode* g;
    GI_HDM void fetch(int32_t i, int oct, NodeView& v) const
    {
        const TNode& n = g[i];
        v.bmin[0] = n.bmin[0]; v.bmin[1] = n.bmin[1]; v.bmin[2] = n.bmin[2];
        v.bmax[0] = n.bmax[0]; v.bmax[1] = n.bmax[1]; v.bmax[2] = n.bmax[2];
        v.first_ref = n.first_ref; v.n_ref = n.n_ref;
        v.hit = n.link[oct].hit; v.skip = n.link[oct].skip;
    }
    GI_HDM int32_t leaf_id(int32_t i) const { return g[i].leaf_id; }
};

// ------------------------------------------------------------------------------------------------ wide-node walk
// The same walk as trace_nodes / visible_nodes below -- children of every node in the order k ^ a, every child's box tested with
// the reference's slab arithmetic, leaves met in the same order -- but a node's children are tested together from the WNode of
// their parent.  BoundingBox::intersect updates tmin upwards and tmax downwards axis by axis and leaves at the first
// tmax <= tmin (include/bbox.h:47-73); both are monotone, so it returns false exactly when the final tmax <= tmin, and
// `t0 > tmin ? t0 : tmin` is maxNum (a NaN from 0 * inf is ignored), which is what fmax / v_max_f64 compute.
struct WRay { int a; double tc; bool plain; };   // a = direction octant; tc = parameter beyond which no hit can count (content-box culling, the closest-hit walk's cut);
                                                  // plain: a closest-hit walk that asks every entity the reference asks (trace_wide_step): no content culling either
// 8 when the ray runs backwards along an axis (invDir < 0), else 0: the byte offset between a plane pair's near and far side in a wide record.  Read off
// the sign bit of 1/d each time (1/d is never a zero or a NaN) -- three registers a walk does not have to hold
GI_HD int back_off(double inv)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(((uint32_t)__double2hiint(inv) >> 31) << 3);
#else
    return inv < 0.0 ? 8 : 0;
#endif
}
GI_HD WRay wray_make(const Ray& r)
{
    WRay w;
    w.a = (r.d.x < 0.0 ? 1 : 0) | (r.d.z < 0.0 ? 2 : 0) | (r.d.y < 0.0 ? 4 : 0);
    w.tc = INFINITY;
    w.plain = false;
    return w;
}
// Content-box culling.  The reference's octree cuts space, not content: a leaf that holds a piece of the floor spans its whole octant, and
// a ray passing over the floor has to test every triangle in it.  Next to every child's octant box (the reference's test, unchanged) the
// wide records carry the box of everything REFERENCED in that child's sub-tree -- the union of the entities' own boxes, widened by a margin
// a billion times the rounding of a hit point and rounded outwards to float.  A child whose octant the ray enters is skipped when the ray
// misses that box inside [0, tc]: no entity below it can then report a hit (a hit lies on its entity, at t >= 0 and, for a shadow ray,
// before the light), so every hit the reference records is still recorded, in the same order, and the walk stops after the same leaf --
// same results by construction (frames with and without culling are compared bit for bit, gi_set_content_culling).  What it saves is the
// visits that could not have mattered: in the benchmark 96 % of the reflected rays leave the scene, and their walks shrink to a few nodes.
// m: candidate children in the ray's order (bit k = slot k ^ a); returns m without the children the ray cannot hit anything in.
GI_HD uint32_t content_cull(const float* cboxes, const uint32_t* cuse, int32_t node, uint32_t m, const Ray& r, const WRay& wr, uint32_t* n_tested = nullptr)
{
    uint32_t u = cuse[node];
    if (wr.a & 1) u = ((u & 0x55u) << 1) | ((u >> 1) & 0x55u);      // slot bits -> the ray's order, as wide_hits does
    if (wr.a & 2) u = ((u & 0x33u) << 2) | ((u >> 2) & 0x33u);
    if (wr.a & 4) u = ((u & 0x0fu) << 4) | ((u >> 4) & 0x0fu);
    uint32_t todo = m & u;
    if (n_tested) *n_tested = (uint32_t)__builtin_popcount(todo);
    const double o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
    while (todo) {
        const int kk = __builtin_ctz(todo);
        todo &= todo - 1;
        const float* b = cboxes + ((size_t)node * 8 + (size_t)(kk ^ wr.a)) * 6;
        double tn = 0.0, tf = wr.tc;
        for (int ax = 0; ax < 3; ax++) {
            // entry plane first: a NaN (origin exactly on a plane of an axis the ray does not move along: 0 * inf) is ignored by fmax / fmin,
            // which is the right answer -- the origin is inside that closed slab
            const int back = back_off(inv[ax]) ? 3 : 0;
            tn = fmax(tn, ((double)b[ax + back] - o[ax]) * inv[ax]);
            tf = fmin(tf, ((double)b[ax + 3 - back] - o[ax]) * inv[ax]);
        }
        if (!(tf >= tn)) m &= ~(1u << kk);
    }
    return m;
}
// Entity boxes.  A leaf of the reference's octree holds up to a few dozen entities and RayTracer::trace / visible test every one of them
// (include/raytracer.h:290-305,446-472); a ray through the leaf touches the boxes of one or two.  An entity whose own box -- widened by a margin
// a billion times the rounding of a hit point -- the ray misses inside [0, tc] cannot report a hit (the hit point lies on the entity, at t > 0
// and, for a shadow ray, before the light), so its test is skipped: the same hits in the same order, the same draws.  The lanes of a wave first
// sort their leaf's references into a bit mask (a slab test of 24 flops per reference), then run Entity::intersect (~50 flops and a division)
// on the few that are left -- the wave's loop is as long as its longest lane's, and that is now the lane with the most SURVIVORS.
GI_HD bool entity_box_missed(const double* b, const Ray& r, double tc)
{
    const double o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
    double tn = 0.0, tf = tc;
    for (int ax = 0; ax < 3; ax++) {
        const double t0 = (b[ax] - o[ax]) * inv[ax], t1 = (b[3 + ax] - o[ax]) * inv[ax];
        // a NaN (origin on a plane of an axis the ray does not move along: 0 * inf) is ignored by fmin / fmax: the origin is inside that slab
        tn = fmax(tn, fmin(t0, t1));
        tf = fmin(tf, fmax(t0, t1));
    }
    return !(tf >= tn);
}
// the references [first, first + cnt) of a leaf (cnt <= 32) whose boxes the ray touches, as a bit mask; all of them without the table
GI_HD uint32_t entity_survivors(const double* boxes, int32_t first, int32_t cnt, const Ray& r, double tc)
{
    uint32_t m = cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u);
    if (!boxes) return m;
    for (int32_t j = 0; j < cnt; j++)
        if (entity_box_missed(boxes + (size_t)(first + j) * 6, r, tc)) m &= ~(1u << j);
    return m;
}
// bit k of the result: the k-th child in this ray's front-to-back order (slot k ^ a) exists and its box is hit in (tmin0, tmax0)
GI_HD uint32_t wide_hits(const WNode* w, const Ray& r, const WRay& wr, double tmin0, double tmax0)
{
    const char* b = reinterpret_cast<const char*>(w);
    const double o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
    double n[3][3], f[3][3];   // [axis][low side, high side, child 7]: entry and exit parameter
    for (int ax = 0; ax < 3; ax++) {
        const int A = ax * 48, x = back_off(inv[ax]), y = x ^ 8;
        n[ax][0] = (*reinterpret_cast<const double*>(b + A + x) - o[ax]) * inv[ax];
        f[ax][0] = (*reinterpret_cast<const double*>(b + A + y) - o[ax]) * inv[ax];
        n[ax][1] = (*reinterpret_cast<const double*>(b + A + 16 + x) - o[ax]) * inv[ax];
        f[ax][1] = (*reinterpret_cast<const double*>(b + A + 16 + y) - o[ax]) * inv[ax];
        n[ax][2] = (*reinterpret_cast<const double*>(b + A + 32 + y) - o[ax]) * inv[ax];
        f[ax][2] = (*reinterpret_cast<const double*>(b + A + 32 + x) - o[ax]) * inv[ax];
    }
    const double nx[2] = {fmax(n[0][0], tmin0), fmax(n[0][1], tmin0)}, fx[2] = {fmin(f[0][0], tmax0), fmin(f[0][1], tmax0)};
    uint32_t m = 0;
    for (int bz = 0; bz < 2; bz++)
        for (int bx = 0; bx < 2; bx++) {
            const double nA = fmax(nx[bx], n[2][bz]), fA = fmin(fx[bx], f[2][bz]);
            for (int by = 0; by < 2; by++) {
                const double t0 = fmax(nA, n[1][by]), t1 = fmin(fA, f[1][by]);
                if (t1 > t0) m |= 1u << (bx | (bz << 1) | (by << 2));
            }
        }
    {
        const double t0 = fmax(fmax(fmax(n[0][2], tmin0), n[1][2]), n[2][2]), t1 = fmin(fmin(fmin(f[0][2], tmax0), f[1][2]), f[2][2]);
        m = (m & 0x7fu) | (t1 > t0 ? 0x80u : 0u);
    }
    m &= w->exists;
    if (wr.a & 1) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    if (wr.a & 2) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (wr.a & 4) m = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    return m;
}
GI_HD void wide_leaf_box(const WNode* w, int slot, double* lmin, double* lmax)   // the box of child `slot` as partition() built it
{
    const int bit[3] = {slot & 1, (slot >> 2) & 1, (slot >> 1) & 1};   // x, y, z
    for (int ax = 0; ax < 3; ax++) {
        if (slot == 7) { lmin[ax] = w->pl[ax][5]; lmax[ax] = w->pl[ax][4]; }
        else { lmin[ax] = w->pl[ax][bit[ax] * 2]; lmax[ax] = w->pl[ax][bit[ax] * 2 + 1]; }
    }
}

// measurement aid (never defined in the product build): lane-level and wave-level step counts of the wide walk, to tell how much of a
// wave's work is lanes waiting for the slowest ray (tools/divergence_probe.py)
#if defined(GI_EXP_DIV) && defined(__HIP_DEVICE_COMPILE__)
#define GI_DIV(W, k) do { (W).ds[(k)]++; if ((uint32_t)__builtin_ctzll(__ballot(1)) == (threadIdx.x & 63u)) (W).ds[(k) + 1]++; } while (0)
#define GI_DIVN(W, k, n) do { (W).ds[(k)] += (n); if ((uint32_t)__builtin_ctzll(__ballot(1)) == (threadIdx.x & 63u)) (W).ds[(k) + 1] += (n); } while (0)
#else
#define GI_DIV(W, k) do { } while (0)
#define GI_DIVN(W, k, n) do { } while (0)
#endif
// Work a walk actually executes (gi_set_counters(ctx, 2): the streaming kernels count it per lane): walks begun (= tests of the root box), wide
// records visited, child boxes tested from them (one per existing child: together with the root tests these are the reference's
// BoundingBox::intersect calls when nothing is culled), content boxes tested, non-empty leaves met, entity tests.  A record source that does
// not count compiles the ticks to nothing.
struct WalkCnt { uint32_t walks, nodes, child_boxes, cull_tests, leaves, tris, ent_boxes; };
struct NoWalkCnt {
    GI_HDM void tick_walk() const {}
    GI_HDM void tick_node(uint32_t) const {}
    GI_HDM void tick_cull(uint32_t) const {}
    GI_HDM void tick_leaf() const {}
    GI_HDM void tick_tri() const {}
    GI_HDM void tick_ebox(uint32_t) const {}
};
struct GlobalWide : NoWalkCnt {
    static constexpr bool kWide = true;
    static constexpr bool kCoop = false;
    const WNode* g;
    const float* cboxes = nullptr;      // content boxes (null: the walk visits every child whose octant the ray enters)
    const uint32_t* cuse = nullptr;
#ifdef GI_EXP_DIV
    mutable uint32_t ds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    template <class F> GI_HDM auto with(int32_t i, F&& f) const { return f(g + i); }
    GI_HDM uint32_t cull(int32_t node, uint32_t m, const Ray& r, const WRay& wr) const { return (cboxes && !wr.plain) ? content_cull(cboxes, cuse, node, m, r, wr) : m; }
};
// walk state: the node, the children of it still to visit (bit k = k-th in order), and the same masks of its ancestors, one
// byte per level, in a 128-bit shift register (16 levels; deeper trees keep the per-node walk)
struct WWalk {
    int32_t node;
    uint32_t m;
    unsigned long long lo, hi;
};
GI_HD void wwalk_push(WWalk& k, int32_t child) { k.hi = (k.hi << 8) | (k.lo >> 56); k.lo = (k.lo << 8) | k.m; k.node = child; }
GI_HD void wwalk_pop(WWalk& k, int32_t parent) { k.m = (uint32_t)(k.lo & 0xffull); k.lo = (k.lo >> 8) | (k.hi << 56); k.hi >>= 8; k.node = parent; }
// The children of a record the ray enters, in its order (wide_hits), without those whose contents it misses (content_cull).  A walk that carries ONE
// ray on a group of lanes (WN::kCoop: the finisher's lone paths, where a bounce is a chain of dependent steps and every step's length counts)
// tests the eight children side by side, a lane each -- the same arithmetic per child (max / min of the same numbers: their order does not matter), a
// ballot instead of eight turns of a loop: ~30 instructions a step instead of ~150.
#if defined(__HIP_DEVICE_COMPILE__)
template <int G> __device__ __forceinline__ unsigned long long group_ballot(bool pred);
template <class WN>
__device__ __forceinline__ uint32_t walk_hits_coop(const WN& W, int32_t node, const Ray& r, const WRay& wr, double tmin0, double tmax0)
{
    constexpr int G = WN::kGroup;
    const int c = (int)(threadIdx.x & 7u);                        // this lane's child slot (lanes 8 .. G-1 of a group repeat the work of 0 .. 7; their votes are not looked at)
    const double o[3] = {r.o.x, r.o.y, r.o.z}, inv[3] = {r.inv.x, r.inv.y, r.inv.z};
    uint32_t exists = 0;
    const bool hit = W.with(node, [&](const WNode* w) {
        exists = w->exists;
        const char* b = reinterpret_cast<const char*>(w);
        const int bit[3] = {c & 1, (c >> 2) & 1, (c >> 1) & 1};   // x = bit 0, z = bit 1, y = bit 2 of a slot
        double t0 = tmin0, t1 = tmax0;
        for (int ax = 0; ax < 3; ax++) {
            const int A = ax * 48, x = back_off(inv[ax]), y = x ^ 8;
            // child 7 has planes of its own (wide_hits: n[ax][2] from the far-side entry, f[ax][2] from the near-side one)
            const int en = c == 7 ? A + 32 + y : A + 16 * bit[ax] + x, ex = c == 7 ? A + 32 + x : A + 16 * bit[ax] + y;
            t0 = fmax(t0, (*reinterpret_cast<const double*>(b + en) - o[ax]) * inv[ax]);
            t1 = fmin(t1, (*reinterpret_cast<const double*>(b + ex) - o[ax]) * inv[ax]);
        }
        return t1 > t0;
    });
    uint32_t m = (uint32_t)(group_ballot<G>(hit) & 0xffull) & exists;
    if (wr.a & 1) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    if (wr.a & 2) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (wr.a & 4) m = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    const float* cb = W.cbox_table(node);
    if (m == 0u || !cb || wr.plain) return m;
    // content boxes: lane k looks at the k-th child in the ray's order (content_cull's loop, a turn per lane)
    uint32_t u = W.cuse_of(node);
    if (wr.a & 1) u = ((u & 0x55u) << 1) | ((u >> 1) & 0x55u);
    if (wr.a & 2) u = ((u & 0x33u) << 2) | ((u >> 2) & 0x33u);
    if (wr.a & 4) u = ((u & 0x0fu) << 4) | ((u >> 4) & 0x0fu);
    const uint32_t todo = m & u;
    if (todo == 0u) return m;
    bool miss = false;
    if ((todo >> c) & 1u) {
        const float* b = cb + ((size_t)node * 8 + (size_t)(c ^ wr.a)) * 6;
        double tn = 0.0, tf = wr.tc;
        for (int ax = 0; ax < 3; ax++) {
            const int back = back_off(inv[ax]) ? 3 : 0;
            tn = fmax(tn, ((double)b[ax + back] - o[ax]) * inv[ax]);
            tf = fmin(tf, ((double)b[ax + 3 - back] - o[ax]) * inv[ax]);
        }
        miss = !(tf >= tn);
    }
    return m & ~(uint32_t)(group_ballot<G>(miss) & 0xffull);
}
#endif
template <class WN>
GI_HD uint32_t walk_hits(const WN& W, int32_t node, const Ray& ray, const WRay& wr, double tmin0, double tmax0)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (WN::kCoop) return walk_hits_coop(W, node, ray, wr, tmin0, tmax0);
    else
#endif
    {
        uint32_t m = W.with(node, [&](const WNode* w) { W.tick_node(w->exists); return wide_hits(w, ray, wr, tmin0, tmax0); });
        if (m) m = W.cull(node, m, ray, wr);
        return m;
    }
}
// next non-empty leaf whose box the ray touches, or false when the tree is exhausted; leaf = (its parent's record, slot)
template <class WN>
GI_HD bool wwalk_next_leaf(const WN& W, WWalk& k, const Ray& ray, const WRay& wr, double tmin0, double tmax0, int32_t& lnode, int& lslot, int32_t& first, int32_t& cnt)
{
    for (;;) {
        // leave exhausted nodes first, in a loop of its own: the lanes of a wave then reach the box tests below together instead of
        // one lane's ascent making the whole wave step through them
        while (k.m == 0) {
            if (k.node == 0) return false;
            const int32_t par = W.with(k.node, [&](const WNode* w) { return w->parent; });
            wwalk_pop(k, par);
        }
        const int kk = __builtin_ctz(k.m);
        k.m &= k.m - 1;
        const int slot = kk ^ wr.a;
        int32_t ca = 0, cb = 0;
        W.with(k.node, [&](const WNode* w) { ca = w->ca[slot]; cb = w->cb[slot]; return 0; });
        if (cb < 0) {
            GI_DIV(W, 0);
            wwalk_push(k, ca);
            k.m = walk_hits(W, ca, ray, wr, tmin0, tmax0);
            continue;
        }
        lnode = k.node; lslot = slot; first = ca; cnt = cb;
        return true;
    }
}
// One turn of the same walk for callers that interleave the turns of different rays (k_st_shadow): WALK_MOVED = went down one level (the box
// tests of the new node are done), WALK_LEAF = stands on a non-empty leaf, WALK_END = the tree is exhausted.  Ascents are folded into the turn.
enum { WALK_MOVED = 0, WALK_LEAF = 1, WALK_END = 2 };
template <class WN>
GI_HD int wwalk_turn(const WN& W, WWalk& k, const Ray& ray, const WRay& wr, double tmin0, double tmax0, int32_t& lnode, int& lslot, int32_t& first, int32_t& cnt)
{
    while (k.m == 0) {
        if (k.node == 0) return WALK_END;
        const int32_t par = W.with(k.node, [&](const WNode* w) { return w->parent; });
        wwalk_pop(k, par);
    }
    const int kk = __builtin_ctz(k.m);
    k.m &= k.m - 1;
    const int slot = kk ^ wr.a;
    int32_t ca = 0, cb = 0;
    W.with(k.node, [&](const WNode* w) { ca = w->ca[slot]; cb = w->cb[slot]; return 0; });
    if (cb < 0) {
        GI_DIV(W, 0);
        wwalk_push(k, ca);
        k.m = walk_hits(W, ca, ray, wr, tmin0, tmax0);
        return WALK_MOVED;
    }
    lnode = k.node; lslot = slot; first = ca; cnt = cb;
    return WALK_LEAF;
}
template <class WN>
GI_HD bool wwalk_begin(const Scene& S, const WN& W, WWalk& k, const Ray& ray, const WRay& wr, double tmin0, double tmax0)
{
    k.node = 0; k.m = 0; k.lo = 0; k.hi = 0;
    W.tick_walk();
    if (!box_hit(S.root_bmin, S.root_bmax, ray, tmin0, tmax0)) return false;
    k.m = walk_hits(W, 0, ray, wr, tmin0, tmax0);
    return true;
}
// RayTracer::trace over the wide records, one leaf per call: the streaming trace kernel keeps a wave's lanes on different rays and hands a
// finished lane its next ray while the others walk on (k_st_trace), everybody else runs the loop in trace_wide.
struct TraceWalk {
    WRay wr;
    WWalk k;
    bool intersected;
    bool tie;          // two entities at the very same distance were met (trace_wide_step)
    double best_d2;
    double cu, cv;   // the reference's `glm::dvec2 uv` of trace(): written by every successful intersect of a smooth triangle or sphere
};
template <int FEAT, class WN>
GI_HD bool trace_wide_begin(const Scene& S, const WN& W, const Ray& ray, TraceWalk& t)   // false: the ray misses the scene's box
{
    t.wr = wray_make(ray);
    // a ray that runs exactly along an axis plane may lie IN the face two leaves share and touch both all the way: leaves then do not follow
    // each other along it, which the short cuts of trace_wide_step take for granted -- such a ray walks the plain way
    t.intersected = false; t.tie = false;
    t.wr.plain = !(fabs(ray.inv.x) < INFINITY && fabs(ray.inv.y) < INFINITY && fabs(ray.inv.z) < INFINITY);
    t.best_d2 = 0; t.cu = 0; t.cv = 0;
    return wwalk_begin(S, W, t.k, ray, t.wr, 0.0, INFINITY);
}
// Two short cuts of the closest-hit walk, both leaving the hit RayTracer::trace returns (include/raytracer.h:446-472) as it is:
//  * the reference ends a walk only when the best hit lies inside the leaf it was found from; a hit found from an earlier leaf (a large
//    entity sticks out of it) lets it walk on to the end of the ray, testing entities that cannot win (it takes a hit only when it is
//    strictly nearer).  Here nothing that begins more than `cut_margin` behind the best hit is looked at: nodes, leaves, entities.
//  * trace_boxes: for an opaque entity of a scene without textures, the box of the part of it INSIDE the leaf (widened): a hit outside the
//    leaf is found again from the leaf it lies in, with the same arithmetic, so asking for it here only brings the answer forward.
// Both widen by 1e-5 of the scene, a hundred times what the float arithmetic of the tree builder can misplace an entity by.  What the second
// could change is WHICH of two entities at exactly the same distance is met first; such a walk (t.tie) is made again the plain way.
template <class WN>
GI_HD bool trace_wide_over(const Scene& S, const WN& W, const Ray& ray, TraceWalk& t)   // the walk ended: true when it has to be made again
{
    if (!t.tie || t.wr.plain || S.trace_boxes == S.leaf_boxes) return false;   // (whole boxes: the entities were met in the reference's order anyway)
    t.wr.tc = INFINITY;
    t.intersected = false; t.tie = false; t.wr.plain = true;
    t.best_d2 = 0; t.cu = 0; t.cv = 0;
    return wwalk_begin(S, W, t.k, ray, t.wr, 0.0, INFINITY);
}
template <int FEAT, class WN>
GI_HD bool trace_wide_step(const Scene& S, const WN& W, const Ray& ray, const Rng& rng, uint32_t alpha_purpose, TraceWalk& t, HitRec& best)   // false: the walk is over
{
    int32_t lnode = 0, first = 0, cnt = 0;
    int lslot = 0;
    bool term = false;
    if (wwalk_next_leaf(W, t.k, ray, t.wr, 0.0, t.wr.tc, lnode, lslot, first, cnt)) {
    GI_DIV(W, 2);
    W.tick_leaf();
    // an entity beyond the best hit cannot replace it (strict <): without textures its test changes nothing; with them a successful intersect
    // leaves its uv behind for the alpha look-up of the next flat entity (t.cu / t.cv), so every entity of a visited leaf is still asked
#define GI_TCE ((FEAT & GI_FEAT_TEX) ? INFINITY : t.wr.tc)
    auto test = [&](const LeafTri& g) {
        const int32_t ti = g.tri;
        double u, v;
        V3 hp;
        W.tick_tri();
        if (!ent_hit<FEAT>(g, g.matflags, ray, u, v, hp)) return;
        if (FEAT & GI_FEAT_TEX) ent_uv(S, g, g.matflags, ti, u, v, hp, t.cu, t.cv);
        if (!(g.matflags & 2u)) {
            const Mat& m = S.mats[g.matflags >> 3];
            const double alpha = (FEAT & GI_FEAT_TEX) ? mat_alpha(S, m, t.cu, t.cv) : m.opacity * 1.0;
            if (!(rng_draw(rng, alpha_purpose, (uint32_t)S.wleaf_id[lnode * 8 + lslot], (uint32_t)ti) < alpha || m.ior != 1)) return;
        }
        double d2 = len2(hp - ray.o);
        if (t.intersected && d2 == t.best_d2 && ti != best.tri) t.tie = true;
        if (!t.intersected || d2 < t.best_d2) {
            best.pos = hp; best.u = u; best.v = v; best.tri = ti; best.mf = g.matflags;
            if (FEAT & GI_FEAT_TEX) { best.tu = t.cu; best.tv = t.cv; }
            t.best_d2 = d2;
            t.intersected = true;
            if (S.cut_margin >= 0 && !t.wr.plain) t.wr.tc = sqrt(d2) + S.cut_margin;   // nothing that begins behind this can be nearer
            double lmin[3], lmax[3];
            W.with(lnode, [&](const WNode* w) { wide_leaf_box(w, lslot, lmin, lmax); return 0; });
            if (box_contains(lmin, lmax, hp)) term = true;
        }
    };
#ifdef GI_WAVE_UNIFORM_LEAVES
    int32_t first_u, cnt_u;
    if (leaf_is_wave_uniform(first, cnt, first_u, cnt_u)) {
        if (S.leaf_boxes) {
            // every lane holds its own ray against the same entity: the record is only fetched when some lane's ray touches the entity's box
            W.tick_ebox((uint32_t)cnt_u);
            for (int32_t j = 0; j < cnt_u; j++) {
                const Box6 bx = leaf_box_scalar(S.trace_boxes + (size_t)(first_u + j) * 6);
                const bool touch = t.wr.plain || !entity_box_missed(bx.b, ray, GI_TCE);
                if (__ballot(touch) == 0ull) continue;
                const LeafTri g = leaf_tri_scalar(S.leaf_tris + first_u + j);
                GI_DIV(W, 4);
                if (touch) test(g);
            }
        } else
            for (int32_t j = 0; j < cnt_u; j++) { GI_DIV(W, 4); test(leaf_tri_scalar(S.leaf_tris + first_u + j)); }
    } else
#endif
    if (S.leaf_boxes && cnt <= 32) {
        // the references whose boxes the ray touches, in leaf order (one or two per leaf: a single record in flight)
        uint32_t m = cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u);
        if (!t.wr.plain) m = entity_survivors(S.trace_boxes, first, cnt, ray, GI_TCE);
        W.tick_ebox((uint32_t)cnt);
        while (m) {
            const int j = __builtin_ctz(m);
            m &= m - 1;
            GI_DIV(W, 6);
            test(S.leaf_tris[first + j]);
        }
    } else
        for (int32_t j = 0; j < cnt; j += 2) {
            const LeafTri g0 = S.leaf_tris[first + j];
            const LeafTri g1 = S.leaf_tris[first + (j + 1 < cnt ? j + 1 : j)];
            GI_DIV(W, 6);
            test(g0);
            if (j + 1 < cnt) { GI_DIV(W, 6); test(g1); }
        }
#undef GI_TCE
    if (!term) return true;
    }
    return trace_wide_over(S, W, ray, t);
}
template <int FEAT, class WN>
GI_HD bool trace_wide(const Scene& S, const WN& W, const Ray& ray, const Rng& rng, uint32_t alpha_purpose, HitRec& best)
{
    TraceWalk t;
    if (!trace_wide_begin<FEAT>(S, W, ray, t)) return false;
    while (trace_wide_step<FEAT>(S, W, ray, rng, alpha_purpose, t, best)) { }
    return t.intersected;
}
// RayTracer::visible over the wide records, one leaf per call (k_st_shadow hands idle lanes new shadow rays; everybody else loops in visible_wide)
struct VisWalk {
    WRay wr;
    WWalk k;
    double tmax;
};
template <int FEAT, class WN>
GI_HD bool visible_wide_begin(const Scene& S, const WN& W, const Ray& ray, double mt, VisWalk& v)   // false: the segment misses the scene's box
{
    v.wr = wray_make(ray);
    v.tmax = sqrt(mt) - GI_SHADOW_BIAS;
    v.wr.tc = sqrt(mt) * (1.0 + 1e-9);   // a blocker lies before the light: 0 < |hit - o|^2 < mt
    return wwalk_begin(S, W, v.k, ray, v.wr, 0.0, v.tmax);
}
enum { VIS_DONE = 0, VIS_MORE = 1, VIS_BLOCKED = 2 };
// does anything in this leaf block the segment?  (the triangle loop of RayTracer::visible for one leaf of the walk)
template <int FEAT, class WN>
GI_HD bool visible_leaf_blocks(const Scene& S, const WN& W, const Ray& ray, double mt, const Rng& rng, uint32_t light_index, int32_t lnode, int lslot, int32_t first, int32_t cnt, const double* boxes)
{
    GI_DIV(W, 2);
    W.tick_leaf();
    auto blocks = [&](const LeafTri& g) -> bool {
        const int32_t ti = g.tri;
        double u, vv;
        V3 hp;
        W.tick_tri();
        if (!ent_hit<FEAT>(g, g.matflags, ray, u, vv, hp)) return false;
        if (!(g.matflags & 2u)) {
            const Mat& m = S.mats[g.matflags >> 3];
            double alpha = m.opacity * 1.0;
            if (FEAT & GI_FEAT_TEX) { double cu = 0, cv = 0; ent_uv(S, g, g.matflags, ti, u, vv, hp, cu, cv); alpha = mat_alpha(S, m, cu, cv); }
            if (!(rng_draw(rng, P_SHADOW_ALPHA | (light_index << 8), (uint32_t)S.wleaf_id[lnode * 8 + lslot], (uint32_t)ti) < alpha || m.ior != 1)) return false;
        }
        const double ts = len2(hp - ray.o);
        return (ts < mt) && (ts > 0);
    };
#ifdef GI_WAVE_UNIFORM_LEAVES
    int32_t first_u, cnt_u;
    if (leaf_is_wave_uniform(first, cnt, first_u, cnt_u)) {
        bool hit = false;
        const double tc = sqrt(mt) * (1.0 + 1e-9);
        if (boxes) W.tick_ebox((uint32_t)cnt_u);
        for (int32_t j = 0; j < cnt_u; j++) {     // wave-uniform trip count: the record address stays scalar
            bool touch = !hit;
            if (boxes) {                          // (wave-uniform branch) fetch the record only when a lane that still looks for a blocker touches the entity's box
                const Box6 bx = leaf_box_scalar(boxes + (size_t)(first_u + j) * 6);
                touch = touch && !entity_box_missed(bx.b, ray, tc);
                if (__ballot(touch) == 0ull) continue;
            }
            const LeafTri g = leaf_tri_scalar(S.leaf_tris + first_u + j);
            GI_DIV(W, 4);
            if (touch) hit = blocks(g);
            if (__ballot(!hit) == 0ull) break;
        }
        return hit;
    }
#endif
    if (boxes && cnt <= 32) {
        uint32_t m = entity_survivors(boxes, first, cnt, ray, sqrt(mt) * (1.0 + 1e-9));   // a blocker lies before the light: 0 < |hit - o|^2 < mt
        W.tick_ebox((uint32_t)cnt);
        while (m) {
            const int j = __builtin_ctz(m);
            m &= m - 1;
            GI_DIV(W, 6);
            if (blocks(S.leaf_tris[first + j])) return true;
        }
        return false;
    }
    for (int32_t j = 0; j < cnt; j++) {
        GI_DIV(W, 6);
        if (blocks(S.leaf_tris[first + j])) return true;
    }
    return false;
}
template <int FEAT, class WN>
GI_HD int visible_wide_step(const Scene& S, const WN& W, const Ray& ray, double mt, const Rng& rng, uint32_t light_index, VisWalk& v)
{
    int32_t lnode = 0, first = 0, cnt = 0;
    int lslot = 0;
    if (!wwalk_next_leaf(W, v.k, ray, v.wr, 0.0, v.tmax, lnode, lslot, first, cnt)) return VIS_DONE;
    return visible_leaf_blocks<FEAT>(S, W, ray, mt, rng, light_index, lnode, lslot, first, cnt, S.leaf_boxes) ? VIS_BLOCKED : VIS_MORE;
}
// what visible() asks of the medium once nothing solid blocks the segment: include/raytracer.h:308-316
template <int FEAT>
GI_HD bool visible_through_fog(const Scene& S, const Ray& ray, double mt, const Rng& rng, uint32_t light_index)
{
    if ((FEAT & GI_FEAT_FOG) && S.n_fog > 0) {
        double tmin = 0, tmx = mt;
        if (atmosphere_bounds(S, ray, tmin, tmx)) {
            V3 fh, fc;
            if (raymarch(S, ray, fh, fc, tmin, tmx, rng, P_FOG_SHADOW + 16u * light_index)) return false;
        }
    }
    return true;
}
template <int FEAT, class WN>
GI_HD bool visible_wide(const Scene& S, const WN& W, const Ray& ray, double mt, const Rng& rng, uint32_t light_index)
{
    VisWalk v;
    if (visible_wide_begin<FEAT>(S, W, ray, mt, v)) {
        for (;;) {
            const int r = visible_wide_step<FEAT>(S, W, ray, mt, rng, light_index, v);
            if (r == VIS_BLOCKED) return false;
            if (r == VIS_DONE) break;
        }
    }
    return visible_through_fog<FEAT>(S, ray, mt, rng, light_index);
}

#if defined(__HIPCC__)
// ---- one ray per WAVE (the finisher's last stages: a handful of depth-65 paths, each a chain of dependent bounces).  All 64 lanes
// carry the same ray and walk the nodes together; at a leaf every lane tests its own triangle.  The reference's sequential loop
// (include/raytracer.h:447-468: take a hit when it is the first or strictly nearer than the best so far; remember if any taken hit
// lay inside the leaf's box) is reproduced from an exclusive prefix minimum over the lanes: lane j "takes" its hit exactly when
// d2_j < min(best before the leaf, d2_i of the accepted hits i < j).  The last taking lane holds the final best.
// The lanes of a wave form 64 / G groups of G lanes (G = WN::kGroup: 64 or 16); a group carries one ray.  Groups walk different rays through
// the same loops (a group whose ray needs fewer turns sits them out), every collective stays inside the group: shuffles of width G, ballots
// masked to the group's lanes.
template <int G> __device__ __forceinline__ unsigned long long group_ballot(bool pred)
{
    const unsigned long long b = __ballot(pred);
    if constexpr (G == 64) return b;
    else return (b >> ((threadIdx.x & 63u) & ~(unsigned)(G - 1))) & ((1ull << G) - 1ull);
}
template <int FEAT, class WN>
__device__ __forceinline__ bool trace_wide_coop(const Scene& S, const WN& W, const Ray& ray, const Rng& rng, uint32_t alpha_purpose, HitRec& best)
{
    if constexpr ((FEAT & GI_FEAT_TEX) != 0) return trace_wide<FEAT>(S, W, ray, rng, alpha_purpose, best);   // the carried uv is sequential: every lane walks alone
    else {
    constexpr int G = WN::kGroup;
    const int lane = (int)(threadIdx.x & 63u), gl = lane & (G - 1), gb = lane - gl;
    WRay wr = wray_make(ray);
    bool intersected = false;
    double best_d2 = 0;
    WWalk k;
    if (!wwalk_begin(S, W, k, ray, wr, 0.0, INFINITY)) return false;
    for (;;) {
        int32_t lnode = 0, first = 0, cnt = 0;
        int lslot = 0;
        if (!wwalk_next_leaf(W, k, ray, wr, 0.0, wr.tc, lnode, lslot, first, cnt)) break;
        bool term = false;
        double lmin[3], lmax[3];
        W.with(lnode, [&](const WNode* w) { wide_leaf_box(w, lslot, lmin, lmax); return 0; });
        for (int32_t base = 0; base < cnt; base += G) {
            const int32_t j = base + gl;
            bool ok = false;
            double u = 0, v = 0, d2 = INFINITY;
            V3 hp = v3(0, 0, 0);
            int32_t ti = 0;
            uint32_t mf = 0;
            if (j < cnt) {
                const LeafTri g = S.leaf_tris[first + j];
                ti = g.tri; mf = g.matflags;
                ok = ent_hit<FEAT>(g, mf, ray, u, v, hp);
                if (ok && !(mf & 2u)) {
                    const Mat& m = S.mats[mf >> 3];
                    ok = rng_draw(rng, alpha_purpose, (uint32_t)S.wleaf_id[lnode * 8 + lslot], (uint32_t)ti) < m.opacity * 1.0 || m.ior != 1;
                }
                if (ok) d2 = len2(hp - ray.o);
            }
            double pm = ok ? d2 : INFINITY;   // inclusive prefix minimum over the group's lanes
            for (int off = 1; off < G; off <<= 1) {
                const double t = __shfl_up(pm, (unsigned)off, G);
                if (gl >= off) pm = fmin(pm, t);
            }
            double ex = __shfl_up(pm, 1u, G);
            if (gl == 0) ex = INFINITY;
            if (intersected) ex = fmin(ex, best_d2);
            const bool take = ok && d2 < ex;
            const unsigned long long tm = group_ballot<G>(take);
            if (tm != 0ull) {
                if (group_ballot<G>(take && box_contains(lmin, lmax, hp)) != 0ull) term = true;
                const int last = gb + 63 - __clzll((long long)tm);
                best.pos = v3(__shfl(hp.x, last), __shfl(hp.y, last), __shfl(hp.z, last));
                best.u = __shfl(u, last); best.v = __shfl(v, last);
                best.tri = __shfl(ti, last); best.mf = (uint32_t)__shfl((int)mf, last);
                best_d2 = __shfl(d2, last);
                intersected = true;
                if (S.cut_margin >= 0) wr.tc = sqrt(best_d2) + S.cut_margin;   // no look behind the best hit (trace_wide_step): the reference walks on to the end of the ray
                                                                              // when the hit was found from an earlier leaf -- a lone path inside a glass body does that at every bounce
            }
        }
        if (term) break;
    }
    return intersected;
    }
}
template <int FEAT, class WN>
__device__ __forceinline__ bool visible_wide_coop(const Scene& S, const WN& W, const Ray& ray, double mt, const Rng& rng, uint32_t light_index)
{
    if constexpr ((FEAT & GI_FEAT_TEX) != 0) return visible_wide<FEAT>(S, W, ray, mt, rng, light_index);
    else {
    constexpr int G = WN::kGroup;
    const int gl = (int)(threadIdx.x & (unsigned)(G - 1));
    WRay wr = wray_make(ray);
    const double tmax = sqrt(mt) - GI_SHADOW_BIAS;
    wr.tc = sqrt(mt) * (1.0 + 1e-9);
    WWalk k;
    if (wwalk_begin(S, W, k, ray, wr, 0.0, tmax)) {
        for (;;) {
            int32_t lnode = 0, first = 0, cnt = 0;
            int lslot = 0;
            if (!wwalk_next_leaf(W, k, ray, wr, 0.0, tmax, lnode, lslot, first, cnt)) break;
            for (int32_t base = 0; base < cnt; base += G) {
                const int32_t j = base + gl;
                bool blocks = false;
                if (j < cnt) {
                    const LeafTri g = S.leaf_tris[first + j];
                    double u, v;
                    V3 hp;
                    bool ok = ent_hit<FEAT>(g, g.matflags, ray, u, v, hp);
                    if (ok && !(g.matflags & 2u)) {
                        const Mat& m = S.mats[g.matflags >> 3];
                        ok = rng_draw(rng, P_SHADOW_ALPHA | (light_index << 8), (uint32_t)S.wleaf_id[lnode * 8 + lslot], (uint32_t)g.tri) < m.opacity * 1.0 || m.ior != 1;
                    }
                    if (ok) { const double ts = len2(hp - ray.o); blocks = (ts < mt) && (ts > 0); }
                }
                if (group_ballot<G>(blocks) != 0ull) return false;
            }
        }
    }
    return visible_through_fog<FEAT>(S, ray, mt, rng, light_index);
    }
}
#else
template <int FEAT, class WN> bool trace_wide_coop(const Scene&, const WN&, const Ray&, const Rng&, uint32_t, HitRec&) { return false; }        // host builds never
template <int FEAT, class WN> bool visible_wide_coop(const Scene&, const WN&, const Ray&, double, const Rng&, uint32_t) { return false; }    // select kCoop sources
#endif

// RayTracer::trace.  Leaves are met in the order the reference's t0-sorted list has them because the hit/skip links of a
// direction octant visit the children of every node front to back; the walk stops after the first leaf that contains a new
// nearest hit.  Control flow is "while-while": an inner loop walks nodes until THIS lane stands on a non-empty leaf, then the
// leaf's triangles are tested; the 64 lanes of a wave therefore do their node steps together and their triangle tests together
// instead of one lane's triangle loop stalling 63 lanes that want to take a node step.
template <int FEAT, class Nodes>
GI_HD bool trace_nodes(const Scene& S, const Nodes& N, const Ray& ray, const Rng& rng, uint32_t alpha_purpose, HitRec& best, Counters* c)
{
    if constexpr (Nodes::kWide) {
        if constexpr (Nodes::kCoop) return trace_wide_coop<FEAT>(S, N, ray, rng, alpha_purpose, best);
        else return trace_wide<FEAT>(S, N, ray, rng, alpha_purpose, best);
    }
    else {
    const int oct = dir_octant(ray);
    bool intersected = false;
    double best_d2 = 0;
    double cu = 0, cv = 0;
    int32_t node = 0;
    if (c) c->traces++;
    for (;;) {
        // ---- next non-empty leaf the ray touches
        int32_t first = 0, cnt = 0, leaf = -1;
        double lmin[3], lmax[3];
        while (node < S.n_node) {
            NodeView nd;
            N.fetch(node, oct, nd);
            if (c) c->v_trace++;
            if (!box_hit(nd.bmin, nd.bmax, ray, 0.0, INFINITY)) { node = nd.skip; continue; }
            if (nd.n_ref < 0) { node = nd.hit; continue; }
            if (nd.n_ref == 0) { node = nd.skip; continue; }
            leaf = node; first = nd.first_ref; cnt = nd.n_ref;
            lmin[0] = nd.bmin[0]; lmin[1] = nd.bmin[1]; lmin[2] = nd.bmin[2];
            lmax[0] = nd.bmax[0]; lmax[1] = nd.bmax[1]; lmax[2] = nd.bmax[2];
            node = nd.skip;
            break;
        }
        if (leaf < 0) break;
        // ---- its triangles, two records in flight at a time (the fetch of the second overlaps the test of the first)
        bool term = false;
        auto test = [&](const LeafTri& g) {
            const int32_t ti = g.tri;
            double u, v;
            V3 hp;
            if (c) c->tri++;
            if (!ent_hit<FEAT>(g, g.matflags, ray, u, v, hp)) return;
            if (FEAT & GI_FEAT_TEX) ent_uv(S, g, g.matflags, ti, u, v, hp, cu, cv);
            if (!(g.matflags & 2u)) {
                const Mat& m = S.mats[g.matflags >> 3];
                const double alpha = (FEAT & GI_FEAT_TEX) ? mat_alpha(S, m, cu, cv) : m.opacity * 1.0;
                if (!(rng_draw(rng, alpha_purpose, (uint32_t)N.leaf_id(leaf), (uint32_t)ti) < alpha || m.ior != 1)) return;
            }
            double d2 = len2(hp - ray.o);
            if (!intersected || d2 < best_d2) {
                best.pos = hp; best.u = u; best.v = v; best.tri = ti; best.mf = g.matflags;
                if (FEAT & GI_FEAT_TEX) { best.tu = cu; best.tv = cv; }
                best_d2 = d2;
                intersected = true;
                if (box_contains(lmin, lmax, hp)) term = true;
            }
        };
        for (int32_t k = 0; k < cnt; k += 2) {
            const LeafTri g0 = S.leaf_tris[first + k];
            const LeafTri g1 = S.leaf_tris[first + (k + 1 < cnt ? k + 1 : k)];
            test(g0);
            if (k + 1 < cnt) test(g1);
        }
        if (term) break;
    }
    return intersected;
    }
}
GI_HD bool trace(const Scene& S, const Ray& ray, const Rng& rng, uint32_t alpha_purpose, HitRec& best, Counters* c)
{
    best.tu = 0; best.tv = 0;
    if (S.wnodes && !c) {   // the work counters count the reference's per-node box tests: counted runs take the per-node walk
        GlobalWide W;
        W.g = S.wnodes; W.cboxes = S.tcboxes; W.cuse = S.tcuse;   // a closest-hit walk and nothing else: its own content boxes
        return S.n_tex > 0 ? trace_nodes<7>(S, W, ray, rng, alpha_purpose, best, nullptr) : trace_nodes<3>(S, W, ray, rng, alpha_purpose, best, nullptr);
    }
    GlobalNodes N;
    N.g = S.tnodes;
    return S.n_tex > 0 ? trace_nodes<7>(S, N, ray, rng, alpha_purpose, best, c) : trace_nodes<3>(S, N, ray, rng, alpha_purpose, best, c);
}

// RayTracer::visible: any accepted hit with 0 < |hit-o|^2 < mt among the entities of every leaf the segment touches.
template <int FEAT, class Nodes>
GI_HD bool visible_nodes(const Scene& S, const Nodes& N, const Ray& ray, double mt, const Rng& rng, uint32_t light_index, Counters* c)
{
    if constexpr (Nodes::kWide) {
        if constexpr (Nodes::kCoop) return visible_wide_coop<FEAT>(S, N, ray, mt, rng, light_index);
        else return visible_wide<FEAT>(S, N, ray, mt, rng, light_index);
    }
    else {
    const int oct = dir_octant(ray);
    const double tmax = sqrt(mt) - GI_SHADOW_BIAS;
    int32_t node = 0;
    if (c) c->shadows++;
    for (;;) {
        int32_t first = 0, cnt = 0, leaf = -1;
        while (node < S.n_node) {
            NodeView nd;
            N.fetch(node, oct, nd);
            if (c) c->v_shadow++;
            if (!box_hit(nd.bmin, nd.bmax, ray, 0.0, tmax)) { node = nd.skip; continue; }
            if (nd.n_ref < 0) { node = nd.hit; continue; }
            if (nd.n_ref == 0) { node = nd.skip; continue; }
            leaf = node; first = nd.first_ref; cnt = nd.n_ref;
            node = nd.skip;
            break;
        }
        if (leaf < 0) break;
        for (int32_t k = 0; k < cnt; k++) {
            const LeafTri& g = S.leaf_tris[first + k];
            const int32_t ti = g.tri;
            double u, v;
            V3 hp;
            if (c) c->tri++;
            if (!ent_hit<FEAT>(g, g.matflags, ray, u, v, hp)) continue;
            if (!(g.matflags & 2u)) {
                const Mat& m = S.mats[g.matflags >> 3];
                double alpha = m.opacity * 1.0;
                if (FEAT & GI_FEAT_TEX) { double cu = 0, cv = 0; ent_uv(S, g, g.matflags, ti, u, v, hp, cu, cv); alpha = mat_alpha(S, m, cu, cv); }
                if (!(rng_draw(rng, P_SHADOW_ALPHA | (light_index << 8), (uint32_t)N.leaf_id(leaf), (uint32_t)ti) < alpha || m.ior != 1)) continue;
            }
            double ts = len2(hp - ray.o);
            if ((ts < mt) && (ts > 0)) return false;
        }
    }
    if ((FEAT & GI_FEAT_FOG) && S.n_fog > 0) {   // include/raytracer.h:308-316 (the reference bounds this march by the SQUARED length)
        double tmin = 0, tmx = mt;
        if (atmosphere_bounds(S, ray, tmin, tmx)) {
            V3 fh, fc;
            if (raymarch(S, ray, fh, fc, tmin, tmx, rng, P_FOG_SHADOW + 16u * light_index)) return false;
        }
    }
    return true;
    }
}
GI_HD bool visible(const Scene& S, const Ray& ray, double mt, const Rng& rng, uint32_t light_index, Counters* c)
{
    if (S.wnodes && !c) {
        GlobalWide W;
        W.g = S.wnodes; W.cboxes = S.cboxes; W.cuse = S.cuse;
        return S.n_tex > 0 ? visible_nodes<7>(S, W, ray, mt, rng, light_index, nullptr) : visible_nodes<3>(S, W, ray, mt, rng, light_index, nullptr);
    }
    GlobalNodes N;
    N.g = S.tnodes;
    return S.n_tex > 0 ? visible_nodes<7>(S, N, ray, mt, rng, light_index, c) : visible_nodes<3>(S, N, ray, mt, rng, light_index, c);
}

// ------------------------------------------------------------------------------------------------ diagnostics (parity tests)
// The sequence of non-empty leaves the walk of trace() meets for a ray, as canonical (pre-order) node indices, without the early
// stop of trace: what Octree::intersectSorted (include/octree.cpp:188-211,285-313) returns as its t0-sorted list.  Runs the very
// walk the kernels run (wide records when the scene has them, else the per-node links).
GI_HD int leaf_order(const Scene& S, const Ray& ray, int cap, int32_t* out)
{
    int n = 0;
    if (S.wnodes) {
        GlobalWide W;
        W.g = S.wnodes;
        const WRay wr = wray_make(ray);
        WWalk k;
        if (!wwalk_begin(S, W, k, ray, wr, 0.0, INFINITY)) return 0;
        for (;;) {
            int32_t lnode = 0, first = 0, cnt = 0;
            int lslot = 0;
            if (!wwalk_next_leaf(W, k, ray, wr, 0.0, INFINITY, lnode, lslot, first, cnt)) break;
            if (n < cap) out[n] = S.wleaf_id[lnode * 8 + lslot];
            n++;
        }
        return n;
    }
    GlobalNodes N;
    N.g = S.tnodes;
    const int oct = dir_octant(ray);
    int32_t node = 0;
    while (node < S.n_node) {
        NodeView nd;
        N.fetch(node, oct, nd);
        if (!box_hit(nd.bmin, nd.bmax, ray, 0.0, INFINITY)) { node = nd.skip; continue; }
        if (nd.n_ref < 0) { node = nd.hit; continue; }
        if (nd.n_ref > 0) { if (n < cap) out[n] = N.leaf_id(node); n++; }
        node = nd.skip;
    }
    return n;
}
// Known-answer access to the scalar building blocks (include/util.h:100-188, include/util.cpp:27-107) and to the libm calls the path
// makes, as the device evaluates them.  in: up to 9 doubles, out: 3 doubles.
enum { KAT_FAST_POW = 0, KAT_FAST_PRECISE_POW = 1, KAT_HEMI_COS_N = 2, KAT_SAMPLE_PHONG = 3, KAT_SPHERE_CAP = 4, KAT_UNIT_VEC = 5, KAT_REFR = 6, KAT_REFLECT = 7, KAT_RNG = 8,
       KAT_SIN = 16, KAT_COS = 17, KAT_ACOS = 18, KAT_ASIN = 19, KAT_ATAN2 = 20, KAT_POW = 21, KAT_SQRT = 22 };
GI_HD void kat_eval(int what, const double* in, double* out)
{
    V3 r = v3(0, 0, 0);
    switch (what) {
    case KAT_FAST_POW: r.x = fast_pow(in[0], in[1]); break;
    case KAT_FAST_PRECISE_POW: r.x = fast_precise_pow(in[0], in[1]); break;
    case KAT_HEMI_COS_N: r = hemi_cos_n(ld3(in), (float)in[3], (float)in[4], in[5]); break;
    case KAT_SAMPLE_PHONG: r = sample_phong(ld3(in), in[3], in[4], in[5]); break;
    case KAT_SPHERE_CAP: r = sphere_cap_cos(ld3(in), (float)in[3], (float)in[4], in[5], in[6]); break;
    case KAT_UNIT_VEC: r = random_unit_vec(in[0], in[1]); break;
    case KAT_REFR: r = refr(ld3(in), ld3(in + 3), in[6]); break;
    case KAT_REFLECT: r = reflect(ld3(in), ld3(in + 3)); break;
    case KAT_RNG: {   // the counter RNG (a-16): seed hi, seed lo, stream, depth, purpose, a, b -> draw, stream key hi, stream key lo
        Rng g = rng_make(((uint64_t)(uint32_t)in[0] << 32) | (uint64_t)(uint32_t)in[1], (uint32_t)in[2]);
        g.depth = (uint32_t)in[3];
        r.x = rng_draw(g, (uint32_t)in[4], (uint32_t)in[5], (uint32_t)in[6]);
        r.y = (double)(uint32_t)(g.hs >> 32); r.z = (double)(uint32_t)g.hs;
        break;
    }
    case KAT_SIN: r.x = sin(in[0]); break;
    case KAT_COS: r.x = cos(in[0]); break;
    case KAT_ACOS: r.x = acos(in[0]); break;
    case KAT_ASIN: r.x = asin(in[0]); break;
    case KAT_ATAN2: r.x = atan2(in[0], in[1]); break;
    case KAT_POW: r.x = pow(in[0], in[1]); break;
    case KAT_SQRT: r.x = sqrt(in[0]); break;
    default: break;
    }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

// ------------------------------------------------------------------------------------------------ photon gather
// Lane-private max-heap of GI_GATHER_K *float* keys; element i of this lane lives at hp[i * stride] (LDS, bank-conflict free
// for any per-lane i because the lane index is the fastest-varying address component).  Keys are (float)d2: rounding is
// monotone, so key(a) < key(b) implies d2(a) < d2(b); only candidates whose key EQUALS the 32nd key are ambiguous and those
// are resolved with exact doubles below.  8 KB of LDS per wave instead of 16.
struct Heap { float* hp; int stride; int n; };
GI_HD void heap_push(Heap& h, float x)
{
    int i = h.n++;
    while (i > 0) {
        int p = (i - 1) >> 1;
        float pv = h.hp[p * h.stride];
        if (pv >= x) break;
        h.hp[i * h.stride] = pv;
        i = p;
    }
    h.hp[i * h.stride] = x;
}
GI_HD void heap_replace_root(Heap& h, float x)
{
    int i = 0;
    for (;;) {
        int l = 2 * i + 1;
        if (l >= h.n) break;
        int cidx = l;
        float cv = h.hp[l * h.stride];
        if (l + 1 < h.n) {
            float rv = h.hp[(l + 1) * h.stride];
            if (rv > cv) { cidx = l + 1; cv = rv; }
        }
        if (cv <= x) break;
        h.hp[i * h.stride] = cv;
        i = cidx;
    }
    h.hp[i * h.stride] = x;
}
GI_HD bool boxes_touch(const double* amin, const double* amax, const double* qmin, const double* qmax)  // include/bbox.h:33-38
{
    return (amin[0] <= qmax[0] && amax[0] >= qmin[0]) && (amin[1] <= qmax[1] && amax[1] >= qmin[1]) && (amin[2] <= qmax[2] && amax[2] >= qmin[2]);
}
// RayTracer::samplePhotons(pos, dir, 32): candidates = photons of every leaf touching the (+-EPSILON) box of the leaf that
// contains pos; the 32 nearest of them; sum col*dot(photon.dir, dir) / (pi * r32^2).
//   pass 1: 32nd smallest float key tau (heap);   pass 2: exact sums of everything with key < tau, and of the key == tau group;
//   if the tie group is larger than what is still needed (float ties straddling rank 32: rare) pass 3 picks the needed ones by
//   exact distance.
// PhotonMap::Node::getBounds (include/photonMap.cpp:115-134): the leaf whose half-open box contains pos, or -1.  The children
// are the 8 octants around `mid`, so the only child that can contain pos is the one on pos's side of mid on every axis
// (x = bit0, z = bit1, y = bit2); its contains() test is still made, because the upper children end at mid + .5*extent, which
// may fall an ulp short of the parent's box.
GI_HD int32_t gather_find_leaf(const Scene& S, V3 pos)
{
    if (S.n_pnode <= 0) return -1;
    int32_t node = 0;
    if (S.pn_planes) {
        // one record per level: the child's box is made of the parent's planes (checked bit for bit against the children's stored boxes
        // when the map was laid out), so the child's own record is only fetched to go on
        for (;;) {
            const PNode& nd = S.pnodes[node];
            if (nd.first_child < 0) return node;
            const int bx = pos.x >= nd.mid[0] ? 1 : 0, bz = pos.z >= nd.mid[2] ? 1 : 0, by = pos.y >= nd.mid[1] ? 1 : 0;
            const int k = bx | (bz << 1) | (by << 2);
            const int bit[3] = {bx, by, bz};
            double lo[3], hi[3];
            for (int ax = 0; ax < 3; ax++) {
                if (k == 7) { lo[ax] = nd.mid[ax]; hi[ax] = nd.bmax[ax]; }
                else if (bit[ax]) { lo[ax] = nd.u.in.lo2[ax]; hi[ax] = nd.u.in.hi2[ax]; }
                else { lo[ax] = nd.bmin[ax]; hi[ax] = nd.mid[ax]; }
            }
            if (!box_contains(lo, hi, pos)) return -1;   // no child contains pos: box of -inf, nothing is collected
            node = nd.first_child + k;
        }
    }
    while (S.pnodes[node].first_child >= 0) {
        const PNode& nd = S.pnodes[node];
        const int k = (pos.x >= nd.mid[0] ? 1 : 0) | (pos.z >= nd.mid[2] ? 2 : 0) | (pos.y >= nd.mid[1] ? 4 : 0);
        const int32_t ch = nd.first_child + k;
        const PNode& cn = S.pnodes[ch];
        if (!box_contains(cn.bmin, cn.bmax, pos)) return -1;   // no child contains pos: box of -inf, nothing is collected
        node = ch;
    }
    return node;
}
// The same descent for the bulk of the queries (k_st_compact: one per gather query of a pass).  PhotonMap::Node::getBounds picks the child by comparing the
// position with the split point and then asks whether that child's box contains it (include/photonMap.cpp:115-134).  The child's box is made of the
// parent's planes, so the answer can only be "no" when the position lies within rounding (an ulp of the box's size) of the split point or of the map's own
// faces -- the three forms of "the middle" the reference computes differ in the last bit.  A query that keeps clear of every split plane on its way by
// 1e-12 of the map's extent, and of the map's faces, is inside every box on the way: it needs the 32-byte split records only (2.3 MB for the benchmark's
// map: L2-resident, where the 128-byte nodes are not).  Any other query (returns -2) takes gather_find_leaf.  Same leaf by construction.
GI_HD int32_t gather_find_leaf_fast(const Scene& S, V3 pos)
{
    const double p[3] = {pos.x, pos.y, pos.z};
    double ext = 0.0;
    for (int ax = 0; ax < 3; ax++) ext = fmax(ext, S.pmap_bmax[ax] - S.pmap_bmin[ax]);
    const double eps = 1e-12 * ext;
    for (int ax = 0; ax < 3; ax++) if (!(p[ax] > S.pmap_bmin[ax] + eps && p[ax] < S.pmap_bmax[ax] - eps)) return -2;
    int32_t node = 0;
    if (S.pjump) {
        // the first GI_PJUMP_BITS levels at once: the cell of a grid over the map's box the position lies in names the node its descent reaches (the split
        // planes of those levels were checked against the grid's when the table was made: within 1e-12 of the extent); a position within 4 eps of a cell's
        // face starts at the root instead
        int cell[3];
        bool clear = true;
        for (int ax = 0; ax < 3; ax++) {
            const double f = (p[ax] - S.pmap_bmin[ax]) * S.pjump_inv[ax];
            int i = (int)f;
            i = i < 0 ? 0 : (i > GI_PJUMP_N - 1 ? GI_PJUMP_N - 1 : i);
            const double lo = S.pmap_bmin[ax] + (double)i * S.pjump_cell[ax];
            if (!(p[ax] - lo > 4.0 * eps && lo + S.pjump_cell[ax] - p[ax] > 4.0 * eps)) clear = false;
            cell[ax] = i;
        }
        if (clear) node = S.pjump[(cell[1] * GI_PJUMP_N + cell[2]) * GI_PJUMP_N + cell[0]];
    }
    for (;;) {
        const PDescent nd = S.pdescent[node];
        if (nd.first_child < 0) return node;
        int k = 0;
        for (int ax = 0; ax < 3; ax++) {
            const double dlt = p[ax] - nd.mid[ax];
            if (!(fabs(dlt) > eps)) return -2;                       // too close to a split plane to skip the box test (NaN lands here too)
            if (dlt > 0.0) k |= ax == 0 ? 1 : (ax == 2 ? 2 : 4);     // x = bit 0, z = bit 1, y = bit 2
        }
        node = nd.first_child + k;
    }
}
// Visit every candidate photon of a leaf's range list.  Positions are fetched four at a time before any of them is used, so the
// four loads are in flight together instead of one L2 round trip per photon (the loop is latency-bound otherwise).
template <class F>
GI_HD void for_each_candidate(const Scene& S, const PRange* ranges, int n_ranges, F&& f)
{
    for (int r = 0; r < n_ranges; r++) {
        const PRange rg = ranges[r];
        const double* base = S.ph_pos + (size_t)rg.first * 3;
        for (int32_t j = 0; j < rg.count; j += 4) {
            const int32_t last = rg.count - 1;
            const int32_t j1 = j + 1 < last ? j + 1 : last, j2 = j + 2 < last ? j + 2 : last, j3 = j + 3 < last ? j + 3 : last;
            const double* p0 = base + (size_t)j * 3;
            const double* p1 = base + (size_t)j1 * 3;
            const double* p2 = base + (size_t)j2 * 3;
            const double* p3 = base + (size_t)j3 * 3;
            const V3 a = v3(p0[0], p0[1], p0[2]), b = v3(p1[0], p1[1], p1[2]), cc = v3(p2[0], p2[1], p2[2]), d = v3(p3[0], p3[1], p3[2]);
            f(rg.first + j, a);
            if (j + 1 <= last) f(rg.first + j + 1, b);
            if (j + 2 <= last) f(rg.first + j + 2, cc);
            if (j + 3 <= last) f(rg.first + j + 3, d);
        }
    }
}
// The selection as three small steps, so that the per-lane walk below and the wave-cooperative kernel (candidates staged in LDS,
// gi_kernels.hip) run the very same arithmetic:  g_key per candidate (pass 1), g_acc per candidate (pass 2), g_end.
struct GatherAcc {
    V3 pos, dir;
    Heap h;
    float tau;
    int ncand;
    V3 s_lt, s_eq;
    int c_lt, c_eq;
    double r_eq;
};
GI_HD void g_begin(GatherAcc& a, V3 pos, V3 dir, float* heap_mem, int heap_stride, int ncand)
{
    a.pos = pos; a.dir = dir;
    a.h.hp = heap_mem; a.h.stride = heap_stride; a.h.n = 0;
    a.tau = 0; a.ncand = ncand;
    a.s_lt = v3(0, 0, 0); a.s_eq = v3(0, 0, 0);
    a.c_lt = 0; a.c_eq = 0; a.r_eq = 0;
}
GI_HD void g_key(GatherAcc& a, V3 pp)
{
    float key = (float)len2(pp - a.pos);
    if (a.h.n < GI_GATHER_K) { heap_push(a.h, key); a.tau = a.h.hp[0]; }
    else if (key < a.tau) { heap_replace_root(a.h, key); a.tau = a.h.hp[0]; }
}
GI_HD void g_acc(GatherAcc& a, V3 pp, const double* dc)
{
    double d2 = len2(pp - a.pos);
    float key = (float)d2;
    if (key <= a.tau) {
        V3 contrib = v3(dc[3], dc[4], dc[5]) * dot(v3(dc[0], dc[1], dc[2]), a.dir);
        if (key < a.tau) { a.s_lt = a.s_lt + contrib; a.c_lt++; }
        else { a.s_eq = a.s_eq + contrib; a.c_eq++; a.r_eq = d2 > a.r_eq ? d2 : a.r_eq; }
    }
}
// false: float keys tie across rank 32 -- the caller has to resolve the tie group with exact distances (pass 3)
GI_HD bool g_end(const GatherAcc& a, V3& res)
{
    const int K = a.ncand < GI_GATHER_K ? a.ncand : GI_GATHER_K;
    const int need = K - a.c_lt;  // >= 1: the heap root itself is a candidate with key == tau
    if (a.c_eq > need) return false;
    res = (a.s_lt + a.s_eq) / (GI_PI * a.r_eq);
    return true;
}
GI_HD V3 gather_in_leaf(const Scene& S, int32_t node, V3 pos, V3 dir, float* heap_mem, int heap_stride, int* n_cand_out, Counters* c)
{
    V3 res = v3(0, 0, 0);
    if (n_cand_out) *n_cand_out = 0;
    if (node < 0) return res;
    // PhotonMap::Node::get (include/photonMap.cpp:71-92) for this leaf's (+-EPSILON) box was run once per leaf when the map was
    // laid out (gi_layout.h): its result is the list of photon ranges [nb_off, nb_off + nb_cnt)
    const PNode& lf = S.pnodes[node];
    const PRange* ranges = S.pranges + lf.nb_off;
    const int n_ranges = lf.u.lf.nb_cnt;
    const int ncand = lf.u.lf.nb_photons;
    if (n_cand_out) *n_cand_out = ncand;
    if (c) c->pcand += (unsigned long long)ncand;
    if (ncand == 0) return res;
    GatherAcc a;
    g_begin(a, pos, dir, heap_mem, heap_stride, ncand);
    for_each_candidate(S, ranges, n_ranges, [&](int32_t, V3 pp) { g_key(a, pp); });                                       // pass 1
    for_each_candidate(S, ranges, n_ranges, [&](int32_t idx, V3 pp) { g_acc(a, pp, S.ph_dircol + (size_t)idx * 6); });    // pass 2
    if (g_end(a, res)) return res;
    // pass 3 (rare): the `need` nearest of the tie group by exact distance, one extraction per scan
    const int K = ncand < GI_GATHER_K ? ncand : GI_GATHER_K;
    const int need = K - a.c_lt;
    const float tau = a.tau;
    res = a.s_lt;
    double last = -1.0;
    for (int j = 0; j < need; j++) {
        double best = INFINITY;
        V3 bc = v3(0, 0, 0);
        for (int r = 0; r < n_ranges; r++) {
            const PRange rg = ranges[r];
            const double* pp = S.ph_pos + (size_t)rg.first * 3;
            for (int32_t k = 0; k < rg.count; k++, pp += 3) {
                double d2 = len2(v3(pp[0], pp[1], pp[2]) - pos);
                if ((float)d2 == tau && d2 > last && d2 < best) {
                    const double* dc = S.ph_dircol + (size_t)(rg.first + k) * 6;
                    best = d2;
                    bc = v3(dc[3], dc[4], dc[5]) * dot(v3(dc[0], dc[1], dc[2]), dir);
                }
            }
        }
        if (best == INFINITY) break;   // exact duplicates exhausted the group
        res = res + bc;
        last = best;
    }
    res = res / (GI_PI * last);
    return res;
}
GI_HD V3 gather(const Scene& S, V3 pos, V3 dir, float* heap_mem, int heap_stride, int* n_cand_out, Counters* c)
{
    if (c) c->gathers++;
    return gather_in_leaf(S, gather_find_leaf(S, pos), pos, dir, heap_mem, heap_stride, n_cand_out, c);
}

// ------------------------------------------------------------------------------------------------ shading
// RayTracer::rayType, include/raytracer.h:481-506
GI_HD int ray_type(const Mat& m, double tex_a, const Ray& ray, V3 norm, const Rng& rng)
{
    int type = 2;
    if (m.roughness < .001) type = 0;
    double opacity = tex_a * m.opacity;   // diffuse->getAlpha(minUV) * opacity, include/raytracer.h:486
    if (opacity < 1.0) {  // drand() in [0,1) can exceed the opacity only then
        if (rng_draw(rng, P_TYPE_OPACITY) > opacity) {
            double r0 = pow((1 - m.ior) / (1 + m.ior), 2.0);
            double fs = r0 + (1 - r0) * pow(1 - dot(reflect(ray.d, norm), norm), 5.0);
            if (rng_draw(rng, P_TYPE_FRESNEL) < fs) type = 0;
            else type = 1;
        }
    }
    return type;
}
// RayTracer::secondaryRay, include/raytracer.h:321-379
GI_HD void secondary_ray(const Ray& ray, const Mat& m, V3 color, double tex_a, V3& norm, double sx, double sy, V3& refDir, V3& f, double& roughness, V3& contrib, double& offset, const Rng& rng)
{
    // color = diffuse->get(UV), tex_a = diffuse->getAlpha(UV) (1 and the constant colour without GI_FEAT_TEX)
    bool backface = false;
    if (dot(norm, ray.d) > 0) { norm = norm * -1.0; backface = true; }
    roughness = m.roughness;
    int type = ray_type(m, tex_a, ray, norm, rng);
    if (type == 1) {
        refDir = backface ? refr(ray.d, norm, m.ior) : refr(ray.d, norm, 1.0 / m.ior);
        offset *= -1;
        contrib = v3(1, 1, 1);
        f = 1.0 * color;
    } else if (type == 0) {
        refDir = reflect(ray.d, norm);
        contrib = v3(1, 1, 1);
        f = 1.0 * color;
    } else {
        if (m.roughness < .9) {
            refDir = sample_phong(reflect(ray.d, norm), (1.0 / (m.roughness)) + 1, sx, sy);
            if (dot(refDir, norm) < 0) refDir = reflect(refDir, norm);
        } else
            refDir = hemi_cos_n(norm, (float)sx, (float)sy, 2);
        f = 1.0 * color;
        V3 inf = color;
        contrib = contrib * inf;
        contrib = mix(contrib, inf, 0.5);
    }
}
GI_HD V3 shading_normal(const Scene& S, const HitRec& h)  // include/entities.h:478-485
{
    const TriShade& sh = S.shade[h.tri];
    if (h.mf & 4u) return normalize(h.pos - ld3(sh.n0));   // sphere: normalize(intersect - pos), include/entities.h:84
    if (h.mf & 1u) return (1 - h.u - h.v) * ld3(sh.n0) + h.u * ld3(sh.n1) + h.v * ld3(sh.n2);
    return ld3(sh.fnorm);
}

// ---- one path as explicit stages (the recursion of RayTracer::radiance turned into a loop carrying the throughput).
// L = A_0 + f_0 (A_1 + f_1 (A_2 + ...)) is accumulated as sum_k (prod_{j<k} f_j) A_k with A = color*i + emissive (+ color*caustic,
// added by the gather stage) on continue, color*i on a failed roulette, ambient on a miss and 0 past MAX_DEPTH.
// The megakernel runs the stages back to back per lane; the wavefront pipeline runs each stage as its own kernel over a
// compacted queue of PathRec indices (gi_kernels.hip).
struct alignas(16) PathRec {    // 224 B, one per path in flight (HBM-resident in the wavefront pipeline)
    double o[3], d[3];          // current ray (d normalised)
    uint32_t stream;            // Halton sample index = RNG stream
    int32_t depth;              // -1: slot unused
    int32_t htri;
    uint32_t pad;               // matflags of the hit (material << 3 | flags): the shade stage needs no entity record for them
                                // -- the first 64 B are all a new path needs and all the trace stage reads
    double T[3], contrib[3];    // throughput prod f_j ; the reference's `contrib` (roulette weight).  At depth 0 they are 1 and L is 0 by
                                // definition: the stages do not read them there, so a new path does not have to write them (path_begin_lean)
    double gdir[3], gcoef[3];   // pending photon gather: direction (= refDir) and factor T*color -- bytes 0 .. 159 are what the shade stage writes
                                // (five whole 32-byte sectors of a 32-byte aligned record), 112 .. 183 what the gather stage reads
    double hpos[3], hu, hv;     // hit of the current segment
    double L[3];                // radiance so far where the path is advanced in registers (finisher, megakernel); idle in the streaming passes
};
static_assert(sizeof(PathRec) == 224, "PathRec layout");
enum { ST_CONTINUE = 1, ST_GATHER = 2 };

template <class P>
GI_HD void path_begin_lean(P& p, const Ray& ray, uint32_t sample)   // one 64-byte store: what a depth-0 path consists of (P: PathRec, or the streaming pool's PathRef)
{
    p.o[0] = ray.o.x; p.o[1] = ray.o.y; p.o[2] = ray.o.z;
    p.d[0] = ray.d.x; p.d[1] = ray.d.y; p.d[2] = ray.d.z;
    p.stream = sample; p.depth = 0; p.htri = -1; p.pad = 0;
}
GI_HD void path_begin(PathRec& p, const Ray& ray, uint32_t sample)
{
    p.o[0] = ray.o.x; p.o[1] = ray.o.y; p.o[2] = ray.o.z;
    p.d[0] = ray.d.x; p.d[1] = ray.d.y; p.d[2] = ray.d.z;
    for (int k = 0; k < 3; k++) { p.T[k] = 1; p.contrib[k] = 1; p.L[k] = 0; }
    p.stream = sample; p.depth = 0; p.htri = -1; p.pad = 0;
}
// stage 1: RayTracer::trace for the current segment.  Miss: L += T*ambient and the path is finished (returns false).
template <int FEAT, class Nodes>
GI_HD bool stage_trace_nodes(const Scene& S, const Nodes& N, PathRec& p, uint64_t seed, Counters* c)
{
    Rng rng = rng_make(seed, p.stream);
    rng.depth = (uint32_t)p.depth;
    Ray ray = make_ray_exact(ld3(p.o), ld3(p.d));
    HitRec h;
    if (!trace_nodes<FEAT>(S, N, ray, rng, P_TRACE_ALPHA, h, c)) {
        const bool first = p.depth == 0;   // L = 0, T = 1 (PathRec)
        V3 L = (first ? v3(0, 0, 0) : ld3(p.L)) + (first ? v3(1, 1, 1) : ld3(p.T)) * ld3(S.ambient);
        p.L[0] = L.x; p.L[1] = L.y; p.L[2] = L.z;
        return false;
    }
    p.hpos[0] = h.pos.x; p.hpos[1] = h.pos.y; p.hpos[2] = h.pos.z;
    p.hu = h.u; p.hv = h.v; p.htri = h.tri; p.pad = h.mf;
    if (FEAT & GI_FEAT_TEX) { p.gdir[0] = h.tu; p.gdir[1] = h.tv; }   // minUV rides in the (idle between gather and shade) gather fields
    return true;
}
GI_HD bool stage_trace(const Scene& S, PathRec& p, uint64_t seed, Counters* c)
{
    if (S.wnodes && !c) {
        GlobalWide W;
        W.g = S.wnodes; W.cboxes = S.tcboxes; W.cuse = S.tcuse;
        return S.n_tex > 0 ? stage_trace_nodes<7>(S, W, p, seed, nullptr) : stage_trace_nodes<3>(S, W, p, seed, nullptr);
    }
    GlobalNodes N;
    N.g = S.tnodes;
    return S.n_tex > 0 ? stage_trace_nodes<7>(S, N, p, seed, c) : stage_trace_nodes<3>(S, N, p, seed, c);
}
// coherence key of a continuing ray: direction octant (selects the order the children are walked in), Morton code of the origin inside the
// root box, coarse direction -- rays that are neighbours in this order walk the same nodes for about the same number of steps
GI_HD uint32_t coherence_key(const Scene& S, V3 o, V3 d)
{
    uint32_t m = 0;
    uint32_t q[3];
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (int k = 0; k < 3; k++) {
        double f = (oo[k] - S.root_bmin[k]) / (S.root_bmax[k] - S.root_bmin[k]);
        f = f < 0.0 ? 0.0 : (f > 0.999 ? 0.999 : f);
        q[k] = (uint32_t)(f * 64.0);
    }
    for (int b = 5; b >= 0; b--) m = (m << 3) | (((q[0] >> b) & 1u) << 2) | (((q[1] >> b) & 1u) << 1) | ((q[2] >> b) & 1u);
    const uint32_t oct = (dd[0] < 0.0 ? 1u : 0u) | (dd[2] < 0.0 ? 2u : 0u) | (dd[1] < 0.0 ? 4u : 0u);
    uint32_t db = 0;
    for (int k = 0; k < 3; k++) { double a = fabs(dd[k]); db = (db << 2) | (uint32_t)(a >= 0.999 ? 3.0 : a * 4.0); }
    return (oct << 24) | (m << 6) | db;
}
struct ShadeOut { uint32_t key; V3 gpos; };   // by-products of the shade stage for the queues: sort key of the next ray, position of the gather query
// A shadow query put off: with one light the vertex adds either A = T * (color * i + emissive) -- light visible -- or A0 = T * (color * 0 +
// emissive) to the path's radiance, so the shade stage can leave the walk to a kernel of its own (k_st_shadow: lanes take a new query as
// soon as theirs is answered) that adds the one or the other to the per-sample buffer.  A0 is zero unless the surface emits; then it waits
// in the record's (otherwise idle) L field.  Same bits as the inline form.
struct alignas(16) ShadowQ { double o[3], dir[3], A[3]; uint32_t idx, stream; int32_t depth; uint32_t live, slot, pad; };   // 96 B; idx = the sample's place in the radiance buffer; live: 0 no query, 1 query, 2 query + A0
// stage 2: shading of the hit: secondaryRay, direct light with shadow rays, Russian roulette, next ray.
// Lext: where the path's radiance accumulates when it does not live in the record (streaming pipeline: the per-sample radiance buffer,
// so that a path that ends -- 96 % of the benchmark's reflected rays leave the scene -- has nothing left to read or write); else p.L.
// DEFER (1: scenes with one light, 2: with several): the shadow walks are put off (sq), this function then contains no walk and touches no radiance at all.
template <int FEAT, class Nodes, int DEFER = 0, class P = PathRec>
GI_HD int stage_shade_nodes(const Scene& S, const Nodes& N, P& p, uint64_t seed, Counters* c, ShadeOut* so = nullptr, double* Lext = nullptr, ShadowQ* sq = nullptr)
{
    double* const Lp = Lext ? Lext : p.L;
    Rng rng = rng_make(seed, p.stream);
    const int depth = p.depth;
    rng.depth = (uint32_t)depth;
    if (c) c->shaded++;
    Ray ray = make_ray_exact(ld3(p.o), ld3(p.d));
    HitRec h;
    h.pos = ld3(p.hpos); h.u = p.hu; h.v = p.hv; h.tri = p.htri; h.mf = p.pad;
    float sx = halton_sample(S, 2 + 2 * depth, p.stream);
    float sy = halton_sample(S, 3 + 2 * depth, p.stream);
    const Mat& m = S.mats[h.mf >> 3];
    V3 norm = shading_normal(S, h);
    V3 color = ld3(m.diffuse), emissive = ld3(m.emissive);
    double tex_a = 1;
    if (FEAT & GI_FEAT_TEX) {   // diffuse->get(minUV), emissive->get(minUV), diffuse->getAlpha(minUV): include/raytracer.h:200,269,486
        const double tu = p.gdir[0], tv = p.gdir[1];
        color = tex_get(S, m.dtex, color, tu, tv);
        emissive = tex_get(S, m.etex, emissive, tu, tv);
        tex_a = tex_alpha(S, m.dtex, tu, tv);
    }
    V3 refDir, f = v3(1, 1, 1), i = v3(0, 0, 0), contrib = depth == 0 ? v3(1, 1, 1) : ld3(p.contrib);
    double roughness, offset = GI_SHADOW_BIAS;
    secondary_ray(ray, m, color, tex_a, norm, sx, sy, refDir, f, roughness, contrib, offset, rng);
    if ((FEAT & GI_FEAT_FOG) && S.n_fog > 0) {   // include/raytracer.h:209-228: the segment may end in the medium instead
        double tmin = 0, tmx = length(h.pos - ray.o);
        if (atmosphere_bounds(S, ray, tmin, tmx)) {
            V3 fh, col;
            if (raymarch(S, ray, fh, col, tmin, tmx, rng, P_FOG_CAMERA)) {
                h.pos = fh;
                p.hpos[0] = fh.x; p.hpos[1] = fh.y; p.hpos[2] = fh.z;   // the gather of this vertex uses the scatter point too
                refDir = random_unit_vec(sx, sy);
                f = 1.0 * col;
                color = col;
                contrib = col;
                roughness = 1;
            }
        }
    }
    constexpr bool defer = DEFER != 0, multi = DEFER == 2;   // DEFER 2: several lights, one query each (sq[li]); k_st_shadow asks them from the last light down
    const bool emits = emissive.x != 0.0 || emissive.y != 0.0 || emissive.z != 0.0;
    const int n_def = multi ? S.n_light : 1;
    if (defer) for (int li = 0; li < n_def; li++) { sq[li].live = (li == 0 && emits) ? 2u : 1u; sq[li].stream = p.stream; sq[li].depth = depth; }
    for (int li = 0; li < (defer ? n_def : S.n_light); li++) {
        const LightD& lt = S.lights[li];
        double ry = rng_draw(rng, P_LIGHT_Y | ((uint32_t)li << 8));
        double rx = rng_draw(rng, P_LIGHT_X | ((uint32_t)li << 8));
        V3 lpos = ld3(lt.pos);
        V3 so = h.pos + GI_SHADOW_BIAS * norm;
        V3 lightDir = (lpos + lt.rad * random_unit_vec(rx, ry)) - so;
        double maxt = len2(lightDir);
        double hfrac = 1 / (GI_PI * len2(lpos - h.pos));
        bool vis = true;
        if constexpr (defer) {   // the walk happens in k_st_shadow, from these two vectors
            sq[li].o[0] = so.x; sq[li].o[1] = so.y; sq[li].o[2] = so.z;
            sq[li].dir[0] = lightDir.x; sq[li].dir[1] = lightDir.y; sq[li].dir[2] = lightDir.z;
        } else {
            Ray sray = make_ray(so, lightDir);
            vis = visible_nodes<FEAT>(S, N, sray, maxt, rng, (uint32_t)li, c);
        }
        if (vis) {
            double d = dot(norm, normalize(lpos - h.pos));
            if (d < 0) d = 0;
            // pow(d, 1/roughness): exact shortcuts for the two exponents every constant-texture scene uses (x^1 = x; x^inf for a
            // mirror: 0 below 1, 1 at 1), the general case through pow()
            const double ex = (1.0 / roughness);
            double l = ex == 1.0 ? d : (ex == INFINITY ? (d < 1.0 ? 0.0 : (d == 1.0 ? 1.0 : INFINITY)) : pow(d, ex));
            i = ld3(lt.col) * l * hfrac;
        }
        if constexpr (multi) { sq[li].A[0] = i.x; sq[li].A[1] = i.y; sq[li].A[2] = i.z; }   // this light's share, until T is known below
    }
    p.contrib[0] = contrib.x; p.contrib[1] = contrib.y; p.contrib[2] = contrib.z;
    V3 T = depth == 0 ? v3(1, 1, 1) : ld3(p.T), L = (defer || depth == 0) ? v3(0, 0, 0) : ld3(Lp);
    double q = comp_max(contrib);
    if (depth <= GI_MIN_DEPTH || rng_draw(rng, P_RR) < q) {
        f = f * (depth <= GI_MIN_DEPTH ? 1.0 : (1.0 / q));
        if constexpr (defer) {
            for (int li = 0; li < n_def; li++) {
                const V3 il = multi ? ld3(sq[li].A) : i;
                const V3 A = T * (color * il + emissive);
                sq[li].A[0] = A.x; sq[li].A[1] = A.y; sq[li].A[2] = A.z;
            }
            if (emits) { const V3 A0 = T * (color * v3(0, 0, 0) + emissive); p.L[0] = A0.x; p.L[1] = A0.y; p.L[2] = A0.z; }
        } else {
            L = L + T * (color * i + emissive);
            Lp[0] = L.x; Lp[1] = L.y; Lp[2] = L.z;
        }
        int flags = 0;
        if (depth <= 10 && S.n_pnode > 0) {   // caustic = depth <= 10 ? samplePhotons(minHit, refDir, 32) : 0, include/raytracer.h:258
            V3 gc = T * color;
            p.gdir[0] = refDir.x; p.gdir[1] = refDir.y; p.gdir[2] = refDir.z;
            p.gcoef[0] = gc.x; p.gcoef[1] = gc.y; p.gcoef[2] = gc.z;
            flags |= ST_GATHER;
        }
        T = T * f;
        p.T[0] = T.x; p.T[1] = T.y; p.T[2] = T.z;
        Ray next = make_ray(h.pos + offset * norm, refDir);
        p.o[0] = next.o.x; p.o[1] = next.o.y; p.o[2] = next.o.z;
        p.d[0] = next.d.x; p.d[1] = next.d.y; p.d[2] = next.d.z;
        p.depth = depth + 1;
        if (so) { so->key = coherence_key(S, next.o, next.d); so->gpos = h.pos; }
        if (p.depth <= GI_MAX_DEPTH) flags |= ST_CONTINUE;   // radiance() returns 0 past MAX_DEPTH
        return flags;
    }
    if constexpr (defer) {
        for (int li = 0; li < n_def; li++) {
            const V3 il = multi ? ld3(sq[li].A) : i;
            const V3 A = T * (color * il);
            sq[li].A[0] = A.x; sq[li].A[1] = A.y; sq[li].A[2] = A.z;
        }
        if (emits) sq[0].live = 1u;   // the path ends without the surface's own light: the hidden case adds T * (color * 0) = 0
    } else {
        L = L + T * (color * i);
        Lp[0] = L.x; Lp[1] = L.y; Lp[2] = L.z;
    }
    return 0;
}
GI_HD int stage_shade(const Scene& S, PathRec& p, uint64_t seed, Counters* c)
{
    if (S.wnodes && !c) {
        GlobalWide W;
        W.g = S.wnodes; W.cboxes = S.cboxes; W.cuse = S.cuse;
        return S.n_tex > 0 ? stage_shade_nodes<7>(S, W, p, seed, nullptr) : stage_shade_nodes<3>(S, W, p, seed, nullptr);
    }
    GlobalNodes N;
    N.g = S.tnodes;
    return S.n_tex > 0 ? stage_shade_nodes<7>(S, N, p, seed, c) : stage_shade_nodes<3>(S, N, p, seed, c);
}
// stage 3: the caustic term of the vertex just shaded: L += (T*color) * samplePhotons(hit, refDir, 32)
template <class P>
GI_HD void stage_gather_in_leaf(const Scene& S, P& p, int32_t leaf, float* heap_mem, int heap_stride, double* Lext = nullptr)
{
    double* const Lp = Lext ? Lext : p.L;
    V3 caustic = gather_in_leaf(S, leaf, ld3(p.hpos), ld3(p.gdir), heap_mem, heap_stride, nullptr, nullptr);
    V3 L = ld3(Lp) + ld3(p.gcoef) * caustic;
    Lp[0] = L.x; Lp[1] = L.y; Lp[2] = L.z;
}
GI_HD void stage_gather(const Scene& S, PathRec& p, float* heap_mem, int heap_stride, Counters* c)
{
    V3 caustic = gather(S, ld3(p.hpos), ld3(p.gdir), heap_mem, heap_stride, nullptr, c);
    V3 L = ld3(p.L) + ld3(p.gcoef) * caustic;
    p.L[0] = L.x; p.L[1] = L.y; p.L[2] = L.z;
}
// the stages back to back for one lane (megakernel, function-level entry points)
GI_HD V3 radiance_path(const Scene& S, Ray ray, uint32_t sample, uint64_t seed, float* heap_mem, int heap_stride, Counters* c)
{
    PathRec p;
    path_begin(p, ray, sample);
    for (;;) {
        if (!stage_trace(S, p, seed, c)) break;
        const int fl = stage_shade(S, p, seed, c);
        if (fl & ST_GATHER) stage_gather(S, p, heap_mem, heap_stride, c);
        if (!(fl & ST_CONTINUE)) break;
    }
    return ld3(p.L);
}

// ------------------------------------------------------------------------------------------------ camera / pixel loop
struct Frame {
    V3 cam_pos, cam_up, screen_center, right;
    double sw, sh;
    int32_t w, h;
    HaltonEnumD he;
    int32_t min_samples, max_samples;
    double noise_thresh;
    uint64_t seed;
    int32_t stripe_h, stripe_rank, stripe_world, local_rows;
};
// primary ray of sample s of pixel (x,y): include/raytracer.h:112-129 (FOCAL_BLUR == 0)
GI_HD Ray primary_ray_at(const Scene& S, const Frame& F, uint32_t idx);
GI_HD Ray primary_ray(const Scene& S, const Frame& F, int s, int x, int y, uint32_t& idx)
{
    idx = halton_index(F.he, (uint32_t)s, (uint32_t)x, (uint32_t)y);
    return primary_ray_at(S, F, idx);
}
GI_HD Ray primary_ray_at(const Scene& S, const Frame& F, uint32_t idx)   // the ray of the sample with Halton index idx (the index names the pixel too: Halton_enum)
{
    double xr = halton_sample(S, 0, idx);
    double yr = halton_sample(S, 1, idx);
    double dx = (float)((float)xr * F.he.scale_x);
    double dy = (float)((float)yr * F.he.scale_y);
    V3 pixelPos = F.screen_center + (F.sw * (dx / F.w - .5)) * F.right - (F.sh * (dy / F.h - .5)) * F.cam_up;
    V3 eyePos = F.cam_pos;
    return make_ray(eyePos, normalize(pixelPos - eyePos));
}
// adaptive per-pixel loop state: include/raytracer.h:102-148
struct PixelState { V3 color, lastCol; double var; int samps, s; };
GI_HD void pixel_begin(PixelState& p) { p.color = v3(0.5, 0.5, 0.5); p.lastCol = v3(0, 0, 0); p.var = 0; p.samps = 0; p.s = 0; }
GI_HD bool pixel_wants_sample(const PixelState& p, const Frame& F) { return p.s < F.max_samples && p.samps < F.min_samples; }
GI_HD void pixel_add_sample(PixelState& p, const Frame& F, V3 L)
{
    p.lastCol = p.color;
    if (p.s == 0) p.color = L;
    else p.color = (1.0 * p.s * p.color + L) * (1.0 / (p.s + 1));
    if (p.s > 0) p.var = (1.0 * 5 * p.var + length(p.color - p.lastCol)) * (1.0 / (5 + 1));
    if (p.s > 0 && p.var > F.noise_thresh) p.samps -= 2;
    p.s++;
    p.samps++;
}
// local row r of this rank -> frame row (interleaved stripes)
GI_HD int global_row(const Frame& F, int r) { return ((r / F.stripe_h) * F.stripe_world + F.stripe_rank) * F.stripe_h + r % F.stripe_h; }

// ------------------------------------------------------------------------------------------------ photon emission (tracePhotons)
struct PhotonOut { double v[9]; };
// one (photon index i, light li): up to 500 emission tries; returns true and fills out when a caustic photon is stored
GI_HD bool emit_photon(const Scene& S, int32_t i, int32_t li, int32_t count, int32_t max_depth, uint64_t seed, PhotonOut& out, int32_t& tries_out)
{
    const LightD& l = S.lights[li];
    Rng rng = rng_make(seed ^ GI_PHOTON_SEED_XOR, (uint32_t)i * (uint32_t)S.n_light + (uint32_t)li);
    int tries = 0;
    bool stored = false;
    V3 lpos = ld3(l.pos);
    while (!stored && tries < 500) {
        rng.depth = (uint32_t)tries * 16u;
        float sx = halton_sample(S, 0, (uint32_t)(i * 500 + tries));
        float sy = halton_sample(S, 1, (uint32_t)(i * 500 + tries));
        // Light::getPointInRange, include/light.h:47-53
        V3 pos = l.angle < 1 ? lpos + l.rad * sphere_cap_cos(ld3(l.dir), sx, sy, 1, l.angle) : lpos + l.rad * random_unit_vec(sx, sy);
        double d13 = rng_draw(rng, P_PH_DIR_V);
        double d5 = rng_draw(rng, P_PH_DIR_U);
        V3 dir = sphere_cap_cos(normalize(pos - lpos), (float)fmod(d5 + 5 * i, 1.0), (float)fmod(d13 + 13 * i, 1.0), 2, l.angle);
        Ray r = make_ray(pos, dir);
        V3 col = (1.0 / count) * .5 * l.angle * ld3(l.col);
        HitRec h;
        int depth = 0;
        bool term = false, isCaustic = false;
        if (!trace(S, r, rng, P_PH_TRACE0_ALPHA, h, nullptr)) { tries++; continue; }
        V3 hit = h.pos;
        while (depth < max_depth && !term) {
            rng.depth = (uint32_t)tries * 16u + (uint32_t)depth + 1u;
            double roughness = S.mats[h.mf >> 3].roughness;
            if (roughness < 0.1) {
                if (!trace(S, r, rng, P_TRACE_ALPHA, h, nullptr)) { term = true; continue; }
                hit = h.pos;
                const Mat& m = S.mats[h.mf >> 3];
                V3 norm = shading_normal(S, h);
                V3 refDir, f, contrib = v3(0, 0, 0);
                double offset = GI_SHADOW_BIAS;
                double e13 = rng_draw(rng, P_PH_SEC_V);
                double e5 = rng_draw(rng, P_PH_SEC_U);
                V3 pcolor = ld3(m.diffuse);
                double tex_a = 1;
                if (S.n_tex > 0) { pcolor = tex_get(S, m.dtex, pcolor, h.tu, h.tv); tex_a = tex_alpha(S, m.dtex, h.tu, h.tv); }
                secondary_ray(r, m, pcolor, tex_a, norm, fmod(e5 + 5 * i, 1.0), fmod(e13 + 13 * i, 1.0), refDir, f, roughness, contrib, offset, rng);
                if (S.n_fog > 0) {   // include/raytracer.h:658-675
                    double tmin = 0, tmx = length(hit - r.o);
                    if (atmosphere_bounds(S, r, tmin, tmx)) {
                        V3 ahit, fcol;
                        if (raymarch(S, r, ahit, fcol, tmin, tmx, rng, P_FOG_PHOTON)) {
                            hit = ahit;
                            double g7 = rng_draw(rng, P_PH_FOG_V);
                            double g13 = rng_draw(rng, P_PH_FOG_U);
                            refDir = random_unit_vec(fmod(g13 + 13 * i, 1.0), fmod(g7 + 7 * i, 1.0));
                            f = 1.0 * fcol;
                            roughness = 1;
                        }
                    }
                }
                col = col * f;
                r = make_ray(hit + offset * norm, refDir);
                isCaustic = true;
            }
            if (depth > 0 && isCaustic && roughness >= 0.1) {
                out.v[0] = hit.x; out.v[1] = hit.y; out.v[2] = hit.z;
                out.v[3] = r.d.x; out.v[4] = r.d.y; out.v[5] = r.d.z;
                out.v[6] = col.x; out.v[7] = col.y; out.v[8] = col.z;
                term = true;
                stored = true;
            }
            depth++;
        }
        tries++;
    }
    tries_out = tries;
    return stored;
}

}  // namespace gi
