// gi_kernels.hip -- gfx950 kernels + the C ABI of include/gi_hip.h.
//
// Kernels (all one ray / one pixel per lane, 64-wide waves, scene tables read through L1/L2):
//   k_render     the pixel loop of RayTracer::run with the whole path inside (include/raytracer.h:93-160,167-579)
//   k_trace / k_visible / k_gather / k_radiance   function-level entry points (parity tests, public C++ methods)
//   k_emit       RayTracer::tracePhotons (include/raytracer.h:582-715), one (photon index, light) per lane
// Host side of this file: device layouts (8 direction-ordered copies of the octree, leaf-ordered photons, Halton tables)
// and the gi_* entry points.  No CPU fallback exists: every entry needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/gi_hip.h"
#include "gi_device.h"
#include "gi_layout.h"

using namespace gi;

#define GI_BLOCK 256
#ifndef GI_EXP_SHADE
#define GI_EXP_SHADE 0
#endif

// ================================================================================================= kernels
template <bool COUNT>
__global__ __launch_bounds__(GI_BLOCK) void k_render(Scene S, Frame F, void* out, int out_f64, int32_t* out_spp,
                                                     unsigned int* tile_counter, Counters* counters)
{
    __shared__ float heap[GI_GATHER_K * GI_BLOCK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int tiles_x = (F.w + 7) >> 3, tiles_y = (F.local_rows + 7) >> 3;
    const unsigned int n_tiles = (unsigned int)(tiles_x * tiles_y);
    Counters cnt;
    if (COUNT) memset(&cnt, 0, sizeof cnt);
    for (;;) {
        unsigned int t0 = 0;
        if (lane == 0) t0 = atomicAdd(tile_counter, 1u);   // one tile of 8x8 pixels per wave, dealt dynamically
        const unsigned int tile = (unsigned int)__builtin_amdgcn_readfirstlane((int)t0);
        if (tile >= n_tiles) break;
        const int tx = (int)(tile % (unsigned)tiles_x), ty = (int)(tile / (unsigned)tiles_x);
        const int x = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
        if (x < F.w && ly < F.local_rows) {
            const int y = global_row(F, ly);
            PixelState ps;
            pixel_begin(ps);
            while (pixel_wants_sample(ps, F)) {
                uint32_t idx;
                Ray ray = primary_ray(S, F, ps.s, x, y, idx);
                V3 L = radiance_path(S, ray, idx, F.seed, heap + tid, GI_BLOCK, COUNT ? &cnt : nullptr);
                pixel_add_sample(ps, F, L);
            }
            const size_t o = ((size_t)ly * F.w + x);
            if (out_f64) {
                double* p = (double*)out + o * 3;
                p[0] = ps.color.x; p[1] = ps.color.y; p[2] = ps.color.z;
            } else {
                float* p = (float*)out + o * 3;
                p[0] = (float)ps.color.x; p[1] = (float)ps.color.y; p[2] = (float)ps.color.z;
            }
            if (out_spp) out_spp[o] = ps.s;
        }
    }
    if (COUNT) {
        atomicAdd(&counters->v_trace, cnt.v_trace); atomicAdd(&counters->v_shadow, cnt.v_shadow);
        atomicAdd(&counters->tri, cnt.tri); atomicAdd(&counters->shaded, cnt.shaded);
        atomicAdd(&counters->pcand, cnt.pcand); atomicAdd(&counters->traces, cnt.traces);
        atomicAdd(&counters->shadows, cnt.shadows); atomicAdd(&counters->gathers, cnt.gathers);
    }
}

// ================================================================================================= octree top in LDS
// The traversal kernels of the streaming pipeline copy the first n_l node records (breadth-first order = the top levels of
// the octree, which every ray visits) into LDS once per workgroup and read them from there; deeper nodes come from L1/L2.
extern __shared__ __align__(128) unsigned char gi_dyn_lds[];
#define GI_LDS_NODES 512                      // 64 KB of 128-byte records
struct LdsNodes {
    static constexpr bool kWide = false;
    const TNode* g;
    int32_t n_l;
    __device__ __forceinline__ void fetch(int32_t i, int oct, NodeView& v) const
    {
        if (i < n_l) {
            const TNode* l = reinterpret_cast<const TNode*>(gi_dyn_lds);
            const TNode& n = l[i];
            v.bmin[0] = n.bmin[0]; v.bmin[1] = n.bmin[1]; v.bmin[2] = n.bmin[2];
            v.bmax[0] = n.bmax[0]; v.bmax[1] = n.bmax[1]; v.bmax[2] = n.bmax[2];
            v.first_ref = n.first_ref; v.n_ref = n.n_ref;
            v.hit = n.link[oct].hit; v.skip = n.link[oct].skip;
        } else {
            const TNode& n = g[i];
            v.bmin[0] = n.bmin[0]; v.bmin[1] = n.bmin[1]; v.bmin[2] = n.bmin[2];
            v.bmax[0] = n.bmax[0]; v.bmax[1] = n.bmax[1]; v.bmax[2] = n.bmax[2];
            v.first_ref = n.first_ref; v.n_ref = n.n_ref;
            v.hit = n.link[oct].hit; v.skip = n.link[oct].skip;
        }
    }
    __device__ __forceinline__ int32_t leaf_id(int32_t i) const { return g[i].leaf_id; }
};
__device__ __forceinline__ LdsNodes stage_nodes_in_lds(const Scene& S)
{
    LdsNodes N;
    N.g = S.tnodes;
    N.n_l = S.n_node < GI_LDS_NODES ? S.n_node : GI_LDS_NODES;
    const uint4* src = reinterpret_cast<const uint4*>(S.tnodes);
    uint4* dst = reinterpret_cast<uint4*>(gi_dyn_lds);
    for (int i = threadIdx.x; i < N.n_l * 8; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    return N;
}

// the same for the wide records (gi_device.h: WNode): 292 of them = every inner node of the BASELINE scenes
#define GI_LDS_WNODES 292                     // 64 KB of 224-byte records
// LDS layout "records + content boxes": cap records, 16 bytes of counters, the records' content boxes (192 bytes each), their flag words.
// The kernels that own a CU with one 1024-thread workgroup (k_st_trace, k_st_shadow) use nearly all of its 160 KB: 384 records; the one-path-per-
// group forms of the finisher 292 (and 4 KB of heaps behind them).
#define GI_LDS_WNODES_BIG 384
#define GI_LDS_CNT_OFF(cap) ((cap) * 224)
#define GI_LDS_CBOX_OFF(cap) (GI_LDS_CNT_OFF(cap) + 16)
#define GI_LDS_CUSE_OFF(cap) (GI_LDS_CBOX_OFF(cap) + (cap) * 192)
#define GI_LDS_BOXES_BYTES(cap) (GI_LDS_CUSE_OFF(cap) + (cap) * 4)
#define GI_LDS_BIG_CNT_OFF GI_LDS_CNT_OFF(GI_LDS_WNODES_BIG)
#define GI_LDS_WIDE_BOXES_BYTES GI_LDS_BOXES_BYTES(GI_LDS_WNODES_BIG)
template <bool COUNT>
struct LdsWideT {
    static constexpr bool kWide = true;
    static constexpr bool kCoop = false;
    static constexpr bool kCount = COUNT;
    // executed-work counters of this lane (gi_set_counters(ctx, 2)); an instance that does not count never touches them, and they are gone
    mutable WalkCnt wc = {0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void tick_walk() const { if constexpr (COUNT) wc.walks++; }
    __device__ __forceinline__ void tick_node(uint32_t exists) const { if constexpr (COUNT) { wc.nodes++; wc.child_boxes += (uint32_t)__builtin_popcount(exists & 0xffu); } }
    __device__ __forceinline__ void tick_cull(uint32_t n) const { if constexpr (COUNT) wc.cull_tests += n; }
    __device__ __forceinline__ void tick_leaf() const { if constexpr (COUNT) wc.leaves++; }
    __device__ __forceinline__ void tick_tri() const { if constexpr (COUNT) wc.tris++; }
    __device__ __forceinline__ void tick_ebox(uint32_t n) const { if constexpr (COUNT) wc.ent_boxes += n; }
    const WNode* g;
    const float* cboxes;      // content boxes of the children (gi_device.h: content_cull), read through L1 / L2; null = no culling
    const uint32_t* cuse;
    int32_t n_l;
    int32_t n_lc = 0;         // records whose content boxes are staged in LDS too
    uint32_t box_off = 0, use_off = 0;   // where (GI_LDS_CBOX_OFF / GI_LDS_CUSE_OFF of the kernel's record count)
#ifdef GI_EXP_DIV
    mutable uint32_t ds[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // lane / wave counts of: node steps, leaves, triangle tests (scalar path), triangle tests (per-lane path)
#endif
    template <class F> __device__ __forceinline__ auto with(int32_t i, F&& f) const
    {
        if (i < n_l) return f(reinterpret_cast<const WNode*>(gi_dyn_lds) + i);
        return f(g + i);
    }
    // the content boxes of a record's children: every turn of a walk reads them right after the record -- from L2 that is the longest wait of the turn
    __device__ __forceinline__ uint32_t cull(int32_t node, uint32_t m, const Ray& r, const WRay& wr) const
    {
        if (!cboxes || wr.plain) return m;
        uint32_t nt = 0;
        uint32_t* const ntp = COUNT ? &nt : nullptr;
        const uint32_t res = node < n_lc ? content_cull(reinterpret_cast<const float*>(gi_dyn_lds + box_off), reinterpret_cast<const uint32_t*>(gi_dyn_lds + use_off), node, m, r, wr, ntp)
                                         : content_cull(cboxes, cuse, node, m, r, wr, ntp);
        tick_cull(nt);
        return res;
    }
    // the tables cull() reads, for callers that test a record's children side by side (gi_device.h: walk_hits of the one-ray-per-group walks)
    __device__ __forceinline__ const float* cbox_table(int32_t node) const { return !cboxes ? nullptr : (node < n_lc ? reinterpret_cast<const float*>(gi_dyn_lds + box_off) : cboxes); }
    __device__ __forceinline__ uint32_t cuse_of(int32_t node) const { return node < n_lc ? reinterpret_cast<const uint32_t*>(gi_dyn_lds + use_off)[node] : cuse[node]; }
};
typedef LdsWideT<false> LdsWide;
#ifdef GI_EXP_DIV
__device__ unsigned long long g_div[64 * 16 * 2];   // [workgroup & 63][kernel][counter]
template <class LW> __device__ __forceinline__ void div_flush(const LW& N, int kernel)
{
    for (int k = 0; k < 8; k++) {
        unsigned long long v = N.ds[k];
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&g_div[((blockIdx.x & 63u) * 2 + kernel) * 16 + k], v);
    }
    unsigned long long one = 1;
    for (int o = 32; o; o >>= 1) one += __shfl_xor(one, o);
    if ((threadIdx.x & 63u) == 0) atomicAdd(&g_div[((blockIdx.x & 63u) * 2 + kernel) * 16 + 8], one);
}
extern "C" int gi_debug_div(unsigned long long* out, int reset)
{
    static unsigned long long h[64 * 16 * 2];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_div), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 32; i++) out[i] = 0;
    for (int b = 0; b < 64; b++) for (int i = 0; i < 32; i++) out[i] += h[b * 32 + i];
    if (reset) { static unsigned long long z[64 * 16 * 2]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_div), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
template <int G> struct LdsWideCoop : LdsWide { static constexpr bool kCoop = true; static constexpr int kGroup = G; };   // the same records, one ray per group of G lanes (gi_device.h: trace_wide_coop)
template <bool COUNT = false>
__device__ __forceinline__ LdsWideT<COUNT> stage_wide_in_lds(const Scene& S, bool with_boxes = false, int boxes_cap = GI_LDS_WNODES_BIG, int tables = 0)
{
    // tables: whose content boxes -- 0 the whole entities' (any walk), 1 the closest-hit walk's, 2 those for segments that end at a light
    LdsWideT<COUNT> N;
    N.g = S.wnodes;
    N.cboxes = tables == 1 ? S.tcboxes : (tables == 2 ? S.scboxes : S.cboxes); N.cuse = tables == 1 ? S.tcuse : (tables == 2 ? S.scuse : S.cuse);
    const int cap = with_boxes ? boxes_cap : GI_LDS_WNODES;
    N.box_off = GI_LDS_CBOX_OFF(cap); N.use_off = GI_LDS_CUSE_OFF(cap);
    N.n_l = S.n_wnode < cap ? S.n_wnode : cap;
    const uint4* src = reinterpret_cast<const uint4*>(S.wnodes);
    uint4* dst = reinterpret_cast<uint4*>(gi_dyn_lds);
    for (int i = threadIdx.x; i < N.n_l * (int)(sizeof(WNode) / 16); i += blockDim.x) dst[i] = src[i];
    if (with_boxes && N.cboxes) {
        N.n_lc = N.n_l;
        const uint4* bsrc = reinterpret_cast<const uint4*>(N.cboxes);
        uint4* bdst = reinterpret_cast<uint4*>(gi_dyn_lds + N.box_off);
        for (int i = threadIdx.x; i < N.n_lc * 12; i += blockDim.x) bdst[i] = bsrc[i];   // 8 children x 6 floats = 12 x 16 bytes per record
        uint32_t* udst = reinterpret_cast<uint32_t*>(gi_dyn_lds + N.use_off);
        for (int i = threadIdx.x; i < N.n_lc; i += blockDim.x) udst[i] = N.cuse[i];
    }
    __syncthreads();
    return N;
}
template <int WIDE, bool COUNT = false> struct LdsSrc;
template <bool COUNT> struct LdsSrc<0, COUNT> { typedef LdsNodes type; static __device__ __forceinline__ LdsNodes stage(const Scene& S) { return stage_nodes_in_lds(S); }
                         static __device__ __forceinline__ LdsNodes stage_with_boxes(const Scene& S) { return stage_nodes_in_lds(S); }
                         static __device__ __forceinline__ LdsNodes stage_for_trace(const Scene& S) { return stage_nodes_in_lds(S); } };
template <bool COUNT> struct LdsSrc<1, COUNT> { typedef LdsWideT<COUNT> type; static __device__ __forceinline__ type stage(const Scene& S) { return stage_wide_in_lds<COUNT>(S); }
                         static __device__ __forceinline__ type stage_with_boxes(const Scene& S) { return stage_wide_in_lds<COUNT>(S, true); }
                         static __device__ __forceinline__ type stage_for_trace(const Scene& S) { return stage_wide_in_lds<COUNT>(S, true, GI_LDS_WNODES_BIG, 1); } };   // the closest-hit walk's own content boxes
// what the streaming kernels executed in one frame (gi_get_stream_counters): per-lane WalkCnt sums of k_st_trace and k_st_shadow, the rays handed
// to each, the gather's queries and the candidates they scanned
struct StreamCounters { unsigned long long trace[7], trace_rays, shadow[7], shadow_rays, gather_queries, gather_cand; };
__device__ __forceinline__ void flush_u64(unsigned long long* dst, unsigned long long v)
{
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63u) == 0u && v) atomicAdd(dst, v);
}
__device__ __forceinline__ void flush_walk_cnt(unsigned long long* dst6, const WalkCnt& w)
{
    flush_u64(dst6 + 0, w.walks); flush_u64(dst6 + 1, w.nodes); flush_u64(dst6 + 2, w.child_boxes);
    flush_u64(dst6 + 3, w.cull_tests); flush_u64(dst6 + 4, w.leaves); flush_u64(dst6 + 5, w.tris); flush_u64(dst6 + 6, w.ent_boxes);
}

// ================================================================================================= wavefront pipeline
// The frame is rendered in rounds.  In a round every pixel that still wants samples (adaptive loop of RayTracer::run,
// include/raytracer.h:108-148) starts up to B paths (as many as it is certain to take whatever their variance turns out to
// be); the paths live as PathRec records in an HBM pool (slot = pixel * B + k) and go through
//     k_wf_trace  ->  k_wf_shade  ->  k_wf_gather          (one pass per path depth)
// each kernel working on a compacted queue of slot indices (appended with one atomic per wave from a ballot), so that all
// 64 lanes of a wave run the same stage; k_wf_accum then folds the finished paths into the per-pixel running mean in sample
// order.  Per-path arithmetic is the same stage_* code the megakernel runs, so both give the same numbers.
// The streaming pipeline's paths in flight: ONE ARRAY PER GROUP OF FIELDS that a stage reads or writes together, instead of one 224-byte PathRec per
// path.  A stage wants 40 - 160 bytes of a path; out of records it moved whole DRAM pages for them (exp/aos_bench.hip: the shade stage's
// read-104-write-172 pattern costs 48 ms per 477 M records as 224-byte structures, 24 ms as field arrays).  pool[slot] gives a PathRef -- the field
// names of PathRec as references -- so the stage functions (templates on the path type) read and write exactly the fields they did.
struct alignas(16) PoolRay  { double o[3], d[3]; uint32_t stream; int32_t depth; uint32_t pad_[2]; };   // 64 B: all a new path consists of, all the trace stage reads
struct alignas(16) PoolHit  { double hpos[3], hu, hv; int32_t htri; uint32_t mf; };                     // 48 B: what the trace stage leaves for the shade stage
struct alignas(16) PoolThru { double T[3], contrib[3]; };                                               // 48 B
struct alignas(16) PoolGath { double gdir[3], gcoef[3]; };                                              // 48 B: the pending photon gather (and minUV between trace and shade)
struct PathRef {
    double (&o)[3]; double (&d)[3]; uint32_t& stream; int32_t& depth; int32_t& htri; uint32_t& pad;
    double (&T)[3]; double (&contrib)[3]; double (&gdir)[3]; double (&gcoef)[3]; double (&hpos)[3]; double& hu; double& hv; double (&L)[3];
};
#define GI_POOL_BYTES_PER_SLOT (sizeof(PoolRay) + sizeof(PoolHit) + sizeof(PoolThru) + sizeof(PoolGath) + 24)
struct PathPool {
    PoolRay* ray; PoolHit* hit; PoolThru* thru; PoolGath* gath; double* L;
    __device__ __forceinline__ PathRef operator[](size_t s) const
    {
        return PathRef{ray[s].o, ray[s].d, ray[s].stream, ray[s].depth, hit[s].htri, hit[s].mf, thru[s].T, thru[s].contrib, gath[s].gdir, gath[s].gcoef,
                       hit[s].hpos, hit[s].hu, hit[s].hv, *reinterpret_cast<double (*)[3]>(L + s * 3)};
    }
    __device__ __forceinline__ PathRec load(size_t s) const     // a copy in registers (finisher)
    {
        PathRec p;
        const PathRef r = (*this)[s];
        for (int k = 0; k < 3; k++) { p.o[k] = r.o[k]; p.d[k] = r.d[k]; p.T[k] = r.T[k]; p.contrib[k] = r.contrib[k]; p.gdir[k] = r.gdir[k]; p.gcoef[k] = r.gcoef[k]; p.hpos[k] = r.hpos[k]; p.L[k] = r.L[k]; }
        p.stream = r.stream; p.depth = r.depth; p.htri = r.htri; p.pad = r.pad; p.hu = r.hu; p.hv = r.hv;
        return p;
    }
    __device__ __forceinline__ void store(size_t s, const PathRec& p) const
    {
        const PathRef r = (*this)[s];
        for (int k = 0; k < 3; k++) { r.o[k] = p.o[k]; r.d[k] = p.d[k]; r.T[k] = p.T[k]; r.contrib[k] = p.contrib[k]; r.gdir[k] = p.gdir[k]; r.gcoef[k] = p.gcoef[k]; r.hpos[k] = p.hpos[k]; r.L[k] = p.L[k]; }
        r.stream = p.stream; r.depth = p.depth; r.htri = p.htri; r.pad = p.pad; r.hu = p.hu; r.hv = p.hv;
    }
};
__device__ __forceinline__ void pool_begin(const PathPool& pool, size_t slot, const Ray& ray, uint32_t sample)   // a new path: one 64-byte store (path_begin_lean)
{
    PoolRay r;
    r.o[0] = ray.o.x; r.o[1] = ray.o.y; r.o[2] = ray.o.z;
    r.d[0] = ray.d.x; r.d[1] = ray.d.y; r.d[2] = ray.d.z;
    r.stream = sample; r.depth = 0; r.pad_[0] = 0u; r.pad_[1] = 0u;
    pool.ray[slot] = r;
}
static PathPool make_path_pool(void* base, size_t n)
{
    PathPool P;
    char* b = static_cast<char*>(base);
    P.ray = reinterpret_cast<PoolRay*>(b); b += n * sizeof(PoolRay);
    P.hit = reinterpret_cast<PoolHit*>(b); b += n * sizeof(PoolHit);
    P.thru = reinterpret_cast<PoolThru*>(b); b += n * sizeof(PoolThru);
    P.gath = reinterpret_cast<PoolGath*>(b); b += n * sizeof(PoolGath);
    P.L = reinterpret_cast<double*>(b);
    return P;
}
struct PixRec { double color[3], lastCol[3], var; int32_t samps, s, n, pad; };

__device__ __forceinline__ uint32_t wave_append(unsigned int* counter, bool pred)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned int)__popcll(mask));
    base = (unsigned int)__shfl((int)base, leader);
    return base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
}

// pixel index -> (x, local row): 8x8 tiles so that the 64 lanes of a wave start neighbouring pixels
__device__ __forceinline__ bool wf_pixel_xy(const Frame& F, uint32_t i, int& x, int& ly)
{
    const int tiles_x = (F.w + 7) >> 3;
    const uint32_t tile = i >> 6, t = i & 63u;
    x = (int)(tile % (uint32_t)tiles_x) * 8 + (int)(t & 7u);
    ly = (int)(tile / (uint32_t)tiles_x) * 8 + (int)(t >> 3);
    return x < F.w && ly < F.local_rows;
}

// v-th pixel of the local frame in 8x8-tile order, partial tiles at the right / bottom edge included (no padding pixels)
__device__ __forceinline__ void st_pixel_xy(const Frame& F, uint32_t v, int& x, int& ly)
{
    const uint32_t w = (uint32_t)F.w, rows = (uint32_t)F.local_rows;
    const uint32_t tiles_x = (w + 7u) >> 3, rows_full = rows >> 3;
    uint32_t r = v / (w * 8u);
    if (r > rows_full) r = rows_full;
    const uint32_t hr = r < rows_full ? 8u : (rows & 7u);
    const uint32_t rem = v - r * w * 8u;
    uint32_t tile = rem / (8u * hr);
    uint32_t tw = 8u;
    if (tile >= tiles_x - 1u) { tile = tiles_x - 1u; tw = w - 8u * (tiles_x - 1u); }
    const uint32_t t = rem - tile * 8u * hr;
    x = (int)(tile * 8u + t % tw);
    ly = (int)(r * 8u + t / tw);
}

__global__ __launch_bounds__(GI_BLOCK) void k_wf_init(PixRec* pix, uint32_t n_pix)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += gridDim.x * blockDim.x) {
        PixelState ps;
        pixel_begin(ps);
        PixRec r;
        r.color[0] = ps.color.x; r.color[1] = ps.color.y; r.color[2] = ps.color.z;
        r.lastCol[0] = 0; r.lastCol[1] = 0; r.lastCol[2] = 0;
        r.var = 0; r.samps = 0; r.s = 0; r.n = 0; r.pad = 0;
        pix[i] = r;
    }
}

__global__ __launch_bounds__(GI_BLOCK) void k_wf_gen(Scene S, Frame F, PixRec* pix, PathRec* pool, uint32_t n_pix, int B)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += gridDim.x * blockDim.x) {
        int x, ly;
        int n = 0;
        const bool inside = wf_pixel_xy(F, i, x, ly);
        int s0 = 0;
        if (inside) {
            const int s = pix[i].s, samps = pix[i].samps;
            if (s < F.max_samples && samps < F.min_samples) {
                // samples this pixel takes for certain: each one adds 1 to samps or subtracts 1 (include/raytracer.h:143-147)
                n = min(B, min(F.max_samples - s, F.min_samples - samps));
                s0 = s;
            }
            pix[i].n = n;
        }
        const int y = inside ? global_row(F, ly) : 0;
        for (int k = 0; k < B; k++) {
            PathRec& p = pool[(size_t)i * B + k];
            if (k < n) {
                uint32_t idx;
                Ray ray = primary_ray(S, F, s0 + k, x, y, idx);
                path_begin(p, ray, idx);
            } else
                p.depth = -1;
        }
    }
}

__global__ __launch_bounds__(GI_BLOCK) void k_wf_trace(Scene S, uint64_t seed, PathRec* pool, const uint32_t* q_in, uint32_t n_in,
                                                       uint32_t* q_shade, unsigned int* cnt_shade)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n_in; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + lane;
        bool hit = false;
        uint32_t slot = 0;
        if (i < n_in) {
            slot = q_in ? q_in[i] : i;
            PathRec& p = pool[slot];
            if (p.depth >= 0) hit = stage_trace(S, p, seed, nullptr);
        }
        const uint32_t at = wave_append(cnt_shade, hit);
        if (hit) q_shade[at] = slot;
    }
}

__global__ __launch_bounds__(GI_BLOCK) void k_wf_shade(Scene S, uint64_t seed, PathRec* pool, const uint32_t* q_shade, const unsigned int* cnt_shade,
                                                       uint32_t* q_next, unsigned int* cnt_next, uint32_t* q_gather, unsigned int* cnt_gather)
{
    const uint32_t n_in = *cnt_shade;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n_in; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + lane;
        int fl = 0;
        uint32_t slot = 0;
        if (i < n_in) {
            slot = q_shade[i];
            fl = stage_shade(S, pool[slot], seed, nullptr);
        }
        const uint32_t a = wave_append(cnt_next, (fl & ST_CONTINUE) != 0);
        if (fl & ST_CONTINUE) q_next[a] = slot;
        const uint32_t g = wave_append(cnt_gather, (fl & ST_GATHER) != 0);
        if (fl & ST_GATHER) q_gather[g] = slot;
    }
}

__global__ __launch_bounds__(GI_BLOCK) void k_wf_gather(Scene S, PathRec* pool, const uint32_t* q_gather, const unsigned int* cnt_gather)
{
    __shared__ float heap[GI_GATHER_K * GI_BLOCK];
    const uint32_t n_in = *cnt_gather;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += gridDim.x * blockDim.x)
        stage_gather(S, pool[q_gather[i]], heap + threadIdx.x, GI_BLOCK, nullptr);
}

__global__ __launch_bounds__(GI_BLOCK) void k_wf_accum(Frame F, PixRec* pix, const PathRec* pool, uint32_t n_pix, int B, void* out, int out_f64,
                                                       int32_t* out_spp, unsigned int* n_wanting)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_round = (n_pix + 63u) & ~63u;
    for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n_round; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + lane;
        bool wants = false;
        int x, ly;
        if (i < n_pix && wf_pixel_xy(F, i, x, ly)) {
            PixRec r = pix[i];
            PixelState ps;
            ps.color = ld3(r.color); ps.lastCol = ld3(r.lastCol); ps.var = r.var; ps.samps = r.samps; ps.s = r.s;
            for (int k = 0; k < r.n; k++) pixel_add_sample(ps, F, ld3(pool[(size_t)i * B + k].L));
            r.color[0] = ps.color.x; r.color[1] = ps.color.y; r.color[2] = ps.color.z;
            r.lastCol[0] = ps.lastCol.x; r.lastCol[1] = ps.lastCol.y; r.lastCol[2] = ps.lastCol.z;
            r.var = ps.var; r.samps = ps.samps; r.s = ps.s; r.n = 0;
            pix[i] = r;
            wants = pixel_wants_sample(ps, F);
            const size_t o = ((size_t)ly * F.w + x);
            if (out_f64) {
                double* p = (double*)out + o * 3;
                p[0] = ps.color.x; p[1] = ps.color.y; p[2] = ps.color.z;
            } else {
                float* p = (float*)out + o * 3;
                p[0] = (float)ps.color.x; p[1] = (float)ps.color.y; p[2] = (float)ps.color.z;
            }
            if (out_spp) out_spp[o] = ps.s;
        }
        (void)wave_append(n_wanting, wants);
    }
}

// Adaptive rounds on the streaming machinery (render_adaptive): the paths of a round are started here, compacted into q_new, and then
// go through the k_st_* passes; a finished path leaves its radiance in lbuf[slot] (slot_sample[slot] = slot), which the accumulate
// step folds into the pixel in sample order.
__global__ __launch_bounds__(GI_BLOCK) void k_ad_gen(Scene S, Frame F, PixRec* pix, PathPool pool, unsigned long long* slot_sample, uint32_t n_pix, int B,
                                                    uint32_t* q_new, unsigned int* n_new)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_round = (n_pix + 63u) & ~63u;
    for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n_round; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + lane;
        int x = 0, ly = 0, n = 0, s0 = 0;
        const bool inside = i < n_pix && wf_pixel_xy(F, i, x, ly);
        if (inside) {
            const int s = pix[i].s, samps = pix[i].samps;
            if (s < F.max_samples && samps < F.min_samples) {
                // min - samps samples are certain (each adds 1 to samps or subtracts 1, include/raytracer.h:143-147).  Past those the
                // pixel may stop at any sample, but which one is decided by the radiances in sample order alone: starting up to B
                // samples and folding them until the rule says stop (k_ad_accum) takes the same samples in the same order; the
                // surplus is discarded.  A handful of rounds instead of one per sample of a borderline pixel.
                // At least the certain ones, at most 8 on speculation -- but a pixel that comes back (the noisy few per cent) and could be
                // finished for good in this round gets everything it may still take: a round costs its tail (some 25 ms on a 1080p frame)
                // whatever its size, the surplus samples of those pixels cost next to nothing.
                const int rem = F.max_samples - s;
                n = min(min(B, rem), (s >= F.min_samples && rem <= B) ? rem : max(F.min_samples - samps, 8));
                s0 = s;
            }
            pix[i].n = n;
        }
        const int y = inside ? global_row(F, ly) : 0;
        for (int k = 0; k < B; k++) {                       // wave-uniform trip count: the appends below are collective
            const bool start = k < n;
            const uint32_t slot = i * (uint32_t)B + (uint32_t)k;
            if (start) {
                uint32_t idx;
                Ray ray = primary_ray(S, F, s0 + k, x, y, idx);
                pool_begin(pool, slot, ray, idx);
                slot_sample[slot] = slot;
            }
            const uint32_t at = wave_append(n_new, start);
            if (start) q_new[at] = slot;
        }
    }
}
__global__ __launch_bounds__(GI_BLOCK) void k_ad_accum(Frame F, PixRec* pix, const double* lbuf, uint32_t n_pix, int B, void* out, int out_f64,
                                                      int32_t* out_spp, unsigned int* n_wanting)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_round = (n_pix + 63u) & ~63u;
    for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n_round; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + lane;
        bool wants = false;
        int x, ly;
        if (i < n_pix && wf_pixel_xy(F, i, x, ly)) {
            PixRec r = pix[i];
            PixelState ps;
            ps.color = ld3(r.color); ps.lastCol = ld3(r.lastCol); ps.var = r.var; ps.samps = r.samps; ps.s = r.s;
            for (int k = 0; k < r.n && pixel_wants_sample(ps, F); k++) pixel_add_sample(ps, F, ld3(lbuf + ((size_t)i * B + k) * 3));
            r.color[0] = ps.color.x; r.color[1] = ps.color.y; r.color[2] = ps.color.z;
            r.lastCol[0] = ps.lastCol.x; r.lastCol[1] = ps.lastCol.y; r.lastCol[2] = ps.lastCol.z;
            r.var = ps.var; r.samps = ps.samps; r.s = ps.s; r.n = 0;
            pix[i] = r;
            wants = pixel_wants_sample(ps, F);
            const size_t o = ((size_t)ly * F.w + x);
            if (out_f64) {
                double* p = (double*)out + o * 3;
                p[0] = ps.color.x; p[1] = ps.color.y; p[2] = ps.color.z;
            } else {
                float* p = (float*)out + o * 3;
                p[0] = (float)ps.color.x; p[1] = (float)ps.color.y; p[2] = (float)ps.color.z;
            }
            if (out_spp) out_spp[o] = ps.s;
        }
        (void)wave_append(n_wanting, wants);
    }
}

// ------------------------------------------------------------------------------------------------- streaming variant (fixed spp)
// When min_samples == max_samples every pixel takes exactly spp samples, so samples can be started in any order as long as
// they are folded into the running mean in sample order.  Path slots are refilled with the next unstarted sample as soon as
// their path ends (path regeneration): every pass works on a full pool and the straggler tail is paid once per frame, not
// once per round.  Finished paths write their radiance to Lbuf[sample]; k_st_accum folds Lbuf per pixel in sample order.
// sample id = s * n_pix + i  (i = 8x8-tile-ordered pixel index): consecutive ids are neighbouring pixels of one sample index.
struct StreamCtl {
    unsigned long long next_sample;   // next sample id to start
    unsigned int n_new, n_cont, n_shade, n_gather, n_free, n_free_trace;
};
// Queue appends.  One atomic per wave per loop turn on ONE counter is what these kernels used to do -- and a returning atomic on a single
// address completes every ~11 ns on this part (exp/atomic_bench.hip: 8.2 M of them take 93 ms), so the 8 M waves of a pass could not finish
// faster than that whatever else they did.  Now every workgroup appends to a counter of its own (the same 11 ns, but only its 8-16 waves
// contend) and writes into a segment of its own in a staging queue: the segment of workgroup b starts where the items of workgroups
// 0 .. b-1 would end if every one of them produced an entry (seg_start: pure arithmetic on the grid-stride loop), so segments can never
// overlap.  k_st_compact then closes the gaps -- a streaming copy of a few bytes per entry -- and leaves the total in StreamCtl.
#define GI_CNT_STRIDE 32   // counters 128 bytes apart
#define GI_MAX_PRODUCER_BLOCKS 2048
enum { QC_SHADE = 0, QC_CONT = 1, QC_GATHER = 2, QC_FREE = 3, QC_KINDS = 4 };
__device__ __forceinline__ unsigned int* blk_counter(unsigned int* bc, int kind) { return bc + ((size_t)kind * GI_MAX_PRODUCER_BLOCKS + blockIdx.x) * GI_CNT_STRIDE; }
// items the grid-stride loop `for (i0 = b * bs; i0 < n_in; i0 += grid * bs)` gives to workgroups 0 .. b-1, every chunk counted as full
__host__ __device__ __forceinline__ uint32_t seg_start(uint32_t b, uint32_t n_in, uint32_t grid, uint32_t bs)
{
    const uint32_t n_chunks = (n_in + bs - 1) / bs, full = n_chunks / grid, extra = n_chunks % grid;
    const uint32_t lo = b < extra ? b : extra;
    return (lo * (full + 1) + (b - lo) * full) * bs;
}

// a finished path gives its slot back (its radiance is already in the per-sample buffer: the stages accumulate there)
__device__ __forceinline__ void st_release(uint32_t slot, uint32_t* q_free, unsigned int* n_free, bool finished)
{
    const uint32_t at = wave_append(n_free, finished);
    if (finished) q_free[at] = slot;
}

// items are dealt to the workgroups of the wide instances in chunks (multiples of 64); measured on the benchmark frame: chunks of the workgroup's
// size are best (64: -5 %, 4 workgroup sizes: -8 %, 16: -26 %)
#ifndef GI_TRACE_CHUNK
#define GI_TRACE_CHUNK 1024
#endif
#ifndef GI_SHADE_CHUNK
#define GI_SHADE_CHUNK 512
#endif
#ifndef GI_GATHER_CHUNK
#define GI_GATHER_CHUNK 256
#endif
#ifndef GI_TRACE_BLOCK
#define GI_TRACE_BLOCK 1024
#endif
#ifndef GI_SHADOW_BLOCK
#define GI_SHADOW_BLOCK 1024
#endif
#ifndef GI_SHADOW_LEAF_MIN
#define GI_SHADOW_LEAF_MIN 16   // lanes of a wave that must stand on a leaf before k_st_shadow tests leaves (or nobody is left walking)
#endif
#define GI_SHADE_BLOCK 512
// New samples are started inside the trace kernel (path regeneration fused into the first trace of the path): item i < g.n_gen takes the
// i-th free slot and sample id g.id_base + i, builds its primary ray in registers and traces it at once -- a primary ray that misses
// never exists in memory (its radiance goes straight to lbuf and the slot straight back to the free list), one that hits is stored
// together with its hit.  Sample ids are consecutive per lane = neighbouring pixels of one 8x8 tile, so these waves are coherent.
// What a new path of pixel p needs before its ray can be made -- the pixel's place in the frame (st_pixel_xy: four divisions by run-time numbers) and
// the Halton index of its first sample (halton_index: a 64-bit remainder) -- is the same for all the samples of the pixel: one table per frame instead
// of ~300 instructions per primary ray.
__global__ void k_pixel_table(Frame F, uint32_t n_pix, uint32_t* out);
struct GenArgs {
    Frame F;
    const uint32_t* q_free;           // free slots to fill (nullptr: slots 0 .. n_gen - 1)
    uint32_t n_gen, n_pix;
    unsigned long long id_base, sample_begin;
    int s_begin;
    const uint32_t* pixtab;           // per pixel of this rank, tile order: the Halton index of its sample 0 (k_pixel_table)
    double inv_n_pix;
};
__global__ void k_pixel_table(Frame F, uint32_t n_pix, uint32_t* out)
{
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += gridDim.x * blockDim.x) {
        int x, ly;
        st_pixel_xy(F, p, x, ly);
        const int y = global_row(F, ly);
        out[p] = halton_index(F.he, 0u, (uint32_t)x, (uint32_t)y);
    }
}
// Wide-record instances keep the lanes of a wave on DIFFERENT rays: a lane whose walk is over waits until `refill_min` lanes of its wave
// are idle, then the idle lanes write their results and take the next items of the workgroup's share of the queue (an LDS counter), while
// the others walk on -- a wave no longer runs at the pace of its longest ray with the other lanes switched off (closed scenes: a third
// of the lanes were active per leaf step).  Waves of new paths (neighbouring pixels, leaves shared through the scalar cache) stay in
// lockstep: refill_min = 64 for them.  The workgroup's items are those of the grid-stride loop, so seg_start still bounds its output.
template <int FEAT, int WIDE, bool COUNT = false>
__global__ __launch_bounds__(GI_TRACE_BLOCK) void k_st_trace(Scene S, uint64_t seed, PathPool pool, unsigned long long* slot_sample, unsigned long long sample0,
                                                       GenArgs g, const uint32_t* q_a, uint32_t n_a, const uint32_t* q_b, uint32_t n_b, unsigned int* bc, uint32_t* segs,
                                                       uint32_t* q_shade, uint32_t* q_free, double* lbuf, uint32_t refill_min, StreamCounters* sc)
{
    static_assert(!COUNT || WIDE != 0, "the executed-work counters belong to the wide walk");
    // wide instances: next unfetched item of this workgroup, in its own numbering; lives in the 128 bytes the 292 wide records leave of the 64 KB
    static_assert(sizeof(WNode) == 224 && GI_LDS_WIDE_BOXES_BYTES <= 160 * 1024, "LDS layout of the wide trace / shadow kernels");
    unsigned int* const s_next = reinterpret_cast<unsigned int*>(gi_dyn_lds + GI_LDS_BIG_CNT_OFF);
    if (WIDE != 0 && threadIdx.x == 0) *s_next = 0u;
    const typename LdsSrc<WIDE, COUNT>::type N = LdsSrc<WIDE, COUNT>::stage_for_trace(S);   // ends with a barrier
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_in = g.n_gen + n_a + n_b;
    uint32_t n_rays = 0;   // COUNT: items this lane took
    const uint32_t cs = WIDE != 0 ? (uint32_t)GI_TRACE_CHUNK : blockDim.x;
    const uint32_t seg = seg_start(blockIdx.x, n_in, gridDim.x, cs);   // this workgroup's segment of the staging queues
    if (threadIdx.x == 0) segs[blockIdx.x] = seg;
    unsigned int* const c_shade = blk_counter(bc, QC_SHADE);
    unsigned int* const c_free = blk_counter(bc, QC_FREE);
    // one item: a new path (sample id -> primary ray) or a continuing one (ray from its record)
    auto fetch = [&](uint32_t i, uint32_t& slot, Ray& ray, uint32_t& stream, int32_t& depth) {
        if (i < g.n_gen) {
            slot = g.q_free ? g.q_free[i] : i;
            const unsigned long long id = g.id_base + i;
            // ids of one chunk span less than 2^32 (the radiance buffer bounds the chunk): 32-bit division instead of 64-bit
            const uint32_t rel = (uint32_t)(id - g.sample_begin);
            uint32_t srel = (uint32_t)((double)rel * g.inv_n_pix);               // rel / n_pix: the quotient of the doubles, put right by one step either way
            if ((unsigned long long)srel * g.n_pix > rel) srel--;
            else if ((unsigned long long)(srel + 1u) * g.n_pix <= rel) srel++;
            stream = g.pixtab[rel - srel * g.n_pix] + (uint32_t)(g.s_begin + (int)srel) * g.F.he.inc;        // halton_index(he, s, x, y) = its value for s = 0 + s * inc
            ray = primary_ray_at(S, g.F, stream);
            depth = 0;
        } else {
            slot = i - g.n_gen < n_a ? q_a[i - g.n_gen] : q_b[i - g.n_gen - n_a];
            const PoolRay& p = pool.ray[slot];
            ray = make_ray_exact(ld3(p.o), ld3(p.d));
            stream = p.stream; depth = p.depth;
        }
    };
    // what a finished walk leaves behind: the hit in the record (a new path's record is created here), or the end of the path
    auto retire = [&](uint32_t i, uint32_t slot, const Ray& ray, uint32_t stream, int32_t depth, bool hit, const HitRec& h) {
        const bool gen = i < g.n_gen;
        if (hit) {
            PoolHit p;                                    // put together in registers, stored whole
            if (gen) { pool_begin(pool, slot, ray, stream); slot_sample[slot] = g.id_base + i; }
            if constexpr (WIDE != 0) {
                // the walk carries only WHICH entity it hit: hit point and barycentrics are computed again here, by the same test with the same
                // operands -- ten registers less to hold through the walk (the kernel runs at the 128 its four waves per SIMD leave it)
                double ru = 0, rv = 0;
                V3 rp = v3(0, 0, 0);
                const TriGeom& tg = S.tris[h.tri];
                const uint32_t mf = ((uint32_t)tg.mat << 3) | tg.flags;      // = LeafTri::matflags of every reference to it
                ent_hit<FEAT>(tg, mf, ray, ru, rv, rp);
                p.hpos[0] = rp.x; p.hpos[1] = rp.y; p.hpos[2] = rp.z;
                p.hu = ru; p.hv = rv; p.htri = h.tri; p.mf = mf;
            } else {
            p.hpos[0] = h.pos.x; p.hpos[1] = h.pos.y; p.hpos[2] = h.pos.z;
            p.hu = h.u; p.hv = h.v; p.htri = h.tri; p.mf = h.mf;
            }
            pool.hit[slot] = p;
            if (FEAT & GI_FEAT_TEX) { pool.gath[slot].gdir[0] = h.tu; pool.gath[slot].gdir[1] = h.tv; }   // minUV rides in the (idle between gather and shade) gather fields
        } else if (depth <= GI_MAX_DEPTH) {
            // the path ends here with a miss: L += T * ambient (stage_trace_nodes), on the per-sample radiance buffer where the path's L lives.
            // A path at depth 0 writes (L = 0, T = 1 by definition); a deeper one adds -- and when the scene has no ambient light there is nothing
            // to add (L + T * 0 = L: L is a sum of non-negative terms and photon terms, never -0), so the record's tail, the sample id and
            // the buffer are not touched at all: this is what 96 % of the benchmark's reflected rays do.
            const V3 amb = ld3(S.ambient);
            if (depth == 0) {   // generated here, or handed in by the adaptive loop's generator (k_ad_gen)
                const V3 L = v3(0, 0, 0) + v3(1, 1, 1) * amb;
                const unsigned long long id = gen ? g.id_base + i : slot_sample[slot];
                double* o = lbuf + (id - sample0) * 3;
                o[0] = L.x; o[1] = L.y; o[2] = L.z;
            } else if (amb.x != 0.0 || amb.y != 0.0 || amb.z != 0.0) {
                double* o = lbuf + (slot_sample[slot] - sample0) * 3;
                const V3 L = ld3(o) + ld3(pool.thru[slot].T) * amb;
                o[0] = L.x; o[1] = L.y; o[2] = L.z;
            }
        }
    };
    if constexpr (WIDE != 0) {
        const uint32_t bs = cs;
        const uint32_t total = seg_start(blockIdx.x + 1, n_in, gridDim.x, bs) - seg;   // this workgroup's chunks, counted as full
        bool walking = false, pend = false;
        uint32_t item = 0, slot = 0, stream = 0;
        int32_t depth = 0;
        Ray ray = make_ray_exact(v3(0, 0, 0), v3(1, 0, 0));
        TraceWalk t;
        t.intersected = false;
        HitRec h;
        bool more = true;            // wave-uniform: the workgroup may still have items
        uint32_t thr = 64u;          // wave-uniform: idle lanes that trigger a refill
        for (;;) {
            const unsigned long long busy = __ballot(walking);
            if (busy == 0ull || (more && 64u - (uint32_t)__popcll(busy) >= thr)) {
                const bool hit = pend && t.intersected, fin = pend && !t.intersected;
                if (pend) retire(item, slot, ray, stream, depth, hit, h);
                const uint32_t at = wave_append(c_shade, hit);
                if (hit) q_shade[seg + at] = slot;
                const uint32_t af = wave_append(c_free, fin);
                if (fin) q_free[seg + af] = slot;
                pend = false;
                if (more) {
                    const unsigned long long want = ~busy;
                    const uint32_t nw = (uint32_t)__popcll(want);
                    unsigned int base = 0;
                    if (lane == (uint32_t)(__ffsll((long long)want) - 1)) base = atomicAdd(s_next, nw);
                    base = (unsigned int)__shfl((int)base, __ffsll((long long)want) - 1);
                    more = base + nw < total;
                    // first item this fetch hands out: new paths run in lockstep, continuing ones refill
                    const uint32_t i_first = ((base / bs) * gridDim.x + blockIdx.x) * bs + base % bs;
                    thr = i_first < g.n_gen ? 64u : refill_min;
                    if (!walking) {
                        const uint32_t u = base + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                        const uint32_t i = ((u / bs) * gridDim.x + blockIdx.x) * bs + u % bs;
                        if (u < total && i < n_in) {
                            item = i;
                            if constexpr (COUNT) n_rays++;
                            fetch(i, slot, ray, stream, depth);
                            t.intersected = false;
                            pend = true;          // over already (past MAX_DEPTH radiance() returns 0; a ray that misses the scene's box) unless the walk starts
                            if (depth <= GI_MAX_DEPTH && trace_wide_begin<FEAT>(S, N, ray, t)) { walking = true; pend = false; }
                        }
                    }
                }
                if (__ballot(walking || pend) == 0ull) break;
            }
            if (walking) {
                Rng rng = rng_make(seed, stream);      // only an entity with an alpha test draws: built where it is used, not held through the walk
                rng.depth = (uint32_t)depth;
                if (!trace_wide_step<FEAT>(S, N, ray, rng, P_TRACE_ALPHA, t, h)) { walking = false; pend = true; }
            }
        }
    } else {
        for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n_in; i0 += gridDim.x * blockDim.x) {
            const uint32_t i = i0 + lane;
            bool hit = false, fin = false;
            uint32_t slot = 0;
            if (i < n_in) {
                Ray ray;
                uint32_t stream;
                int32_t depth = 0;
                fetch(i, slot, ray, stream, depth);
                HitRec h;
                if (depth <= GI_MAX_DEPTH) {            // radiance() returns 0 past MAX_DEPTH
                    Rng rng = rng_make(seed, stream);
                    rng.depth = (uint32_t)depth;
                    hit = trace_nodes<FEAT>(S, N, ray, rng, P_TRACE_ALPHA, h, nullptr);
                }
                fin = !hit;
                retire(i, slot, ray, stream, depth, hit, h);
            }
            const uint32_t at = wave_append(c_shade, hit);
            if (hit) q_shade[seg + at] = slot;
            const uint32_t af = wave_append(c_free, fin);
            if (fin) q_free[seg + af] = slot;
        }
    }
#ifdef GI_EXP_DIV
    if constexpr (WIDE != 0) div_flush(N, 0);
#endif
    if constexpr (COUNT) { flush_walk_cnt(sc->trace, N.wc); flush_u64(&sc->trace_rays, n_rays); }
}

__global__ void k_iota(uint32_t* v, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i; }

// the walk-free instance fits its registers at 2 waves per SIMD without a spill and runs fastest that way (round 2, records: 4 waves 260 B of scratch per
// lane, 144 ms on the benchmark; 3 waves 128 ms; 2 waves 128 ms.  Round 3, field arrays, the path held in registers: 229 VGPRs, 58 ms at 2 waves; 3 waves
// spill 212 B: 65 ms)
#ifndef GI_DEFER_WAVES
#define GI_DEFER_WAVES 2
#endif
template <int FEAT, int WIDE, int DEFER>
__global__ __launch_bounds__(GI_SHADE_BLOCK, DEFER ? GI_DEFER_WAVES : 4) void k_st_shade(Scene S, uint64_t seed, PathPool pool, const unsigned long long* slot_sample, unsigned long long sample0,
                                                       const uint32_t* q_shade, const StreamCtl* ctl, unsigned int* bc, uint32_t* segs, uint32_t* q_cont, uint32_t* k_cont, uint32_t* q_gather, double* g_pos,
                                                       uint32_t* q_free, double* lbuf, ShadowQ* shq, const uint32_t* q_orig)
{
    // the waves of a workgroup take its items 64 at a time from a counter in LDS (in the 128 bytes the wide records leave free) instead of
    // a fixed share each: a wave that drew cheap items takes more of them
    unsigned int* const s_next = reinterpret_cast<unsigned int*>(gi_dyn_lds + (size_t)GI_LDS_WNODES * sizeof(WNode));
    if (WIDE != 0 && threadIdx.x == 0) *s_next = 0u;
    typename LdsSrc<WIDE>::type N;
    if constexpr (DEFER != 0) {   // no walk in this instance: the records stay where they are
        N.g = S.wnodes; N.cboxes = S.cboxes; N.cuse = S.cuse; N.n_l = 0;
        __syncthreads();
    } else
        N = LdsSrc<WIDE>::stage(S);   // ends with a barrier
    const uint32_t n_in = ctl->n_shade;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cs = WIDE != 0 ? (uint32_t)GI_SHADE_CHUNK : blockDim.x;
    const uint32_t seg = seg_start(blockIdx.x, n_in, gridDim.x, cs);   // this workgroup's segment of the staging queues (k_st_trace)
    if (threadIdx.x == 0) segs[blockIdx.x] = seg;
    unsigned int* const c_cont = blk_counter(bc, QC_CONT);
    unsigned int* const c_gather = blk_counter(bc, QC_GATHER);
    unsigned int* const c_free = blk_counter(bc, QC_FREE);
    const uint32_t total = seg_start(blockIdx.x + 1, n_in, gridDim.x, cs) - seg;   // this workgroup's chunks of the grid-stride loop, counted as full
    for (uint32_t w = threadIdx.x >> 6;; w += blockDim.x >> 6) {
        uint32_t u = w * 64u;
        if (WIDE != 0) {
            unsigned int b = 0;
            if (lane == 0) b = atomicAdd(s_next, 64u);
            u = (uint32_t)__shfl((int)b, 0);
        }
        if (u >= total) break;
        const uint32_t i = ((u / cs) * gridDim.x + blockIdx.x) * cs + u % cs + lane;
        int fl = 0;
        bool valid = i < n_in;
        uint32_t slot = 0;
        ShadeOut so;
        so.key = 0; so.gpos = v3(0, 0, 0);
        if (valid) {
            slot = q_shade[i];
            if constexpr (DEFER != 0) {
                const int nl = DEFER == 2 ? S.n_light : 1;          // one query per light, the queries of item i side by side
                ShadowQ* const e = shq + (size_t)(q_orig ? q_orig[i] : i) * (size_t)nl;   // q_orig: the item's place in the queue the shadow kernel follows (the trace stage's order)
                // the path in registers: its arrays are read up front, whole records at a time (the loads of one vertex in flight together), and
                // written back as whole records -- only those the vertex changed, and nothing at all for a path that ends here
                PathRec p;
                {
                    const PoolRay r = pool.ray[slot];
                    const PoolHit h = pool.hit[slot];
                    for (int k = 0; k < 3; k++) { p.o[k] = r.o[k]; p.d[k] = r.d[k]; p.hpos[k] = h.hpos[k]; }
                    p.stream = r.stream; p.depth = r.depth; p.htri = h.htri; p.pad = h.mf; p.hu = h.hu; p.hv = h.hv;
                    if (r.depth != 0) { const PoolThru t = pool.thru[slot]; for (int k = 0; k < 3; k++) { p.T[k] = t.T[k]; p.contrib[k] = t.contrib[k]; } }
                    if (FEAT & GI_FEAT_TEX) { p.gdir[0] = pool.gath[slot].gdir[0]; p.gdir[1] = pool.gath[slot].gdir[1]; }
                }
                const V3 hp0 = ld3(p.hpos);
                p.L[0] = NAN;                                       // (written only for an emitting surface: its A0, a finite number)
                ShadowQ ql;                                         // one light: the query is put together in registers too and stored whole
                ql.pad = 0u;
                if constexpr (DEFER == 1) fl = stage_shade_nodes<FEAT, typename LdsSrc<WIDE>::type, DEFER>(S, N, p, seed, nullptr, &so, nullptr, &ql);
                else
                fl = stage_shade_nodes<FEAT, typename LdsSrc<WIDE>::type, DEFER>(S, N, p, seed, nullptr, &so, nullptr, e);
                if (fl != 0) {
                    PoolRay r;
                    for (int k = 0; k < 3; k++) { r.o[k] = p.o[k]; r.d[k] = p.d[k]; }
                    r.stream = p.stream; r.depth = p.depth; r.pad_[0] = 0u; r.pad_[1] = 0u;
                    pool.ray[slot] = r;
                    PoolThru t;
                    for (int k = 0; k < 3; k++) { t.T[k] = p.T[k]; t.contrib[k] = p.contrib[k]; }
                    pool.thru[slot] = t;
                    if (fl & ST_GATHER) {
                        PoolGath gq;
                        for (int k = 0; k < 3; k++) { gq.gdir[k] = p.gdir[k]; gq.gcoef[k] = p.gcoef[k]; }
                        pool.gath[slot] = gq;
                    }
                    if (p.L[0] == p.L[0]) { double* Lp = pool.L + (size_t)slot * 3; Lp[0] = p.L[0]; Lp[1] = p.L[1]; Lp[2] = p.L[2]; }   // A0 of an emitting surface waits there for k_st_shadow
                }
                if ((FEAT & GI_FEAT_FOG) && (p.hpos[0] != hp0.x || p.hpos[1] != hp0.y || p.hpos[2] != hp0.z)) {   // the segment ended in the medium: the gather of this vertex uses the scatter point
                    PoolHit& h = pool.hit[slot];
                    h.hpos[0] = p.hpos[0]; h.hpos[1] = p.hpos[1]; h.hpos[2] = p.hpos[2];
                }
                if constexpr (DEFER == 1) { ql.idx = (uint32_t)(slot_sample[slot] - sample0); ql.slot = slot; *e = ql; }
                else
                for (int li = 0; li < nl; li++) { e[li].idx = (uint32_t)(slot_sample[slot] - sample0); e[li].slot = slot; }
            } else {
                PathRef pr = pool[slot];
                fl = stage_shade_nodes<FEAT>(S, N, pr, seed, nullptr, &so, lbuf + (slot_sample[slot] - sample0) * 3);
            }
        }
        // a path with a pending gather stays alive one more pass even when it may not continue: the trace stage retires it
        const bool cont = valid && (fl & (ST_CONTINUE | ST_GATHER)) != 0;
        const uint32_t a = wave_append(c_cont, cont);
        if (cont) { q_cont[seg + a] = slot; k_cont[seg + a] = so.key; }   // the key of the next ray, from the registers that just held it
        const uint32_t g = wave_append(c_gather, (fl & ST_GATHER) != 0);
        if (fl & ST_GATHER) {
            // the gather query's position goes into the queue as well: the key kernel then reads 24 consecutive bytes per query
            // instead of one scattered sector of the path pool (that read alone kept it at HBM speed)
            q_gather[seg + g] = slot;
            g_pos[(size_t)(seg + g) * 3] = so.gpos.x; g_pos[(size_t)(seg + g) * 3 + 1] = so.gpos.y; g_pos[(size_t)(seg + g) * 3 + 2] = so.gpos.z;
        }
        st_release(slot, q_free + seg, c_free, valid && !cont);
    }
#ifdef GI_EXP_DIV
    if constexpr (WIDE != 0) div_flush(N, 1);
#endif
}

// The shadow queries the shade stage put off (ShadowQ, entry i = item i of the shade queue): RayTracer::visible towards the scene's one
// light, then L += A where the light is visible.  Lanes work like k_st_trace's: whoever has his answer waits until `refill_min` lanes of
// the wave are idle, then they write their results and take the next queries of the workgroup's share.  Inside the shade kernel these walks
// ran with 15-40 % of the lanes (a wave waited for its longest segment, under the register pressure of the whole shade stage).
// MULTI: several lights -- the reference keeps the share of the LAST visible light of a vertex, so a lane asks the item's queries from the last light
// down and stops at the first visible one (the medium's answer is then needed at the end of each walk, not only when the lane retires).
template <int FEAT, int MULTI, bool COUNT = false>
__global__ __launch_bounds__(GI_SHADOW_BLOCK) void k_st_shadow(Scene S, uint64_t seed, PathPool pool, const ShadowQ* shq, const StreamCtl* ctl, double* lbuf, uint32_t refill_min, StreamCounters* sc)
{
    const uint32_t nl = MULTI ? (uint32_t)S.n_light : 1u;
    uint32_t li = 0;   // the light this lane's walk is about
    unsigned int* const s_next = reinterpret_cast<unsigned int*>(gi_dyn_lds + GI_LDS_BIG_CNT_OFF);
    if (threadIdx.x == 0) *s_next = 0u;
    const LdsWideT<COUNT> N = stage_wide_in_lds<COUNT>(S, true, GI_LDS_WNODES_BIG, 2);   // ends with a barrier; every segment of this kernel ends at a light
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_in = ctl->n_shade, bs = (uint32_t)GI_TRACE_CHUNK;
    uint32_t n_rays = 0;   // COUNT: shadow segments this lane walked
    const uint32_t total = seg_start(blockIdx.x + 1, n_in, gridDim.x, bs) - seg_start(blockIdx.x, n_in, gridDim.x, bs);
    bool walking = false, pend = false, blocked = false;
    uint32_t item = 0;
    Ray ray = make_ray_exact(v3(0, 0, 0), v3(1, 0, 0));
    double mt = 0;
    Rng rng = rng_make(seed, 0);
    VisWalk v;
    bool at_leaf = false;
    int32_t lnode = 0, first = 0, cnt = 0;
    int lslot = 0;
    const uint32_t leaf_min = GI_SHADOW_LEAF_MIN;
    bool more = true;   // wave-uniform
    for (;;) {
        const unsigned long long busy = __ballot(walking);
        if (busy == 0ull || (more && 64u - (uint32_t)__popcll(busy) >= refill_min)) {
            if (pend) {
                const ShadowQ& e0 = shq[(size_t)item * nl];          // the item's first query carries what belongs to the vertex
                const ShadowQ& e = shq[(size_t)item * nl + li];      // the query of the light that won (or of light 0 when none did)
                const bool vis = !blocked && (MULTI ? true : visible_through_fog<FEAT>(S, ray, mt, rng, 0u));
                double* o = lbuf + (size_t)e.idx * 3;
                if (e.depth == 0 || vis || e0.live == 2u) {
                    // L += T * (color * i + emissive): i = the light's share when it is visible (A), else 0 (A0: zero unless the surface emits,
                    // and then L + 0 = L is not worth a write -- except at depth 0, where L = 0 + ... starts the sum)
                    const V3 add = vis ? ld3(e.A) : (e0.live == 2u ? ld3(pool.L + (size_t)e.slot * 3) : v3(0, 0, 0));
                    const V3 L = (e.depth == 0 ? v3(0, 0, 0) : ld3(o)) + add;
                    o[0] = L.x; o[1] = L.y; o[2] = L.z;
                }
                pend = false;
            }
            if (more) {
                const unsigned long long want = ~busy;
                const uint32_t nw = (uint32_t)__popcll(want);
                const int leader = __ffsll((long long)want) - 1;
                unsigned int base = 0;
                if ((int)lane == leader) base = atomicAdd(s_next, nw);
                base = (unsigned int)__shfl((int)base, leader);
                more = base + nw < total;
                if (!walking) {
                    const uint32_t u = base + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                    const uint32_t i = ((u / bs) * gridDim.x + blockIdx.x) * bs + u % bs;
                    if (u < total && i < n_in && shq[(size_t)i * nl].live) {
                        li = nl - 1u;
                        const ShadowQ& e = shq[(size_t)i * nl + li];
                        item = i;
                        if constexpr (COUNT) n_rays++;
                        const V3 dir = ld3(e.dir);
                        ray = make_ray(ld3(e.o), dir);
                        mt = len2(dir);
                        rng = rng_make(seed, e.stream);
                        rng.depth = (uint32_t)e.depth;
                        blocked = false;
                        pend = true;      // answered already when the segment misses the scene's box
                        if (visible_wide_begin<FEAT>(S, N, ray, mt, v)) { walking = true; pend = false; at_leaf = false; }
                        else if constexpr (MULTI != 0) blocked = !visible_through_fog<FEAT>(S, ray, mt, rng, li);
                    }
                }
            }
            if (__ballot(walking || pend) == 0ull) break;
        }
        // One turn of the walk for every lane that is between leaves; a lane that reaches a leaf waits there until `leaf_min` lanes stand on
        // one (or nobody is left walking), then those test their leaves' triangles together.  A shadow segment takes some ten turns per leaf
        // it meets: run as "walk to the next leaf, test it" per lane, the turns of a wave ran at the pace of its slowest lane.
        if (walking && !at_leaf) {
            const int r = wwalk_turn(N, v.k, ray, v.wr, 0.0, v.tmax, lnode, lslot, first, cnt);
            if (r == WALK_LEAF) at_leaf = true;
            else if (r == WALK_END) {
                walking = false; pend = true; blocked = false;
                if constexpr (MULTI != 0) blocked = !visible_through_fog<FEAT>(S, ray, mt, rng, li);
            }
        }
        const unsigned long long lf = __ballot(walking && at_leaf);
        if (lf != 0ull && ((uint32_t)__popcll(lf) >= leaf_min || __ballot(walking && !at_leaf) == 0ull)) {
            if (walking && at_leaf) {
                at_leaf = false;
                if (visible_leaf_blocks<FEAT>(S, N, ray, mt, rng, li, lnode, lslot, first, cnt, S.shadow_boxes)) { walking = false; pend = true; blocked = true; }
            }
        }
        if constexpr (MULTI != 0) {   // this light is hidden: on to the one before it, in the same lane
            while (pend && blocked && li > 0u) {
                li--;
                if constexpr (COUNT) n_rays++;
                const ShadowQ& e = shq[(size_t)item * nl + li];
                const V3 dir = ld3(e.dir);
                ray = make_ray(ld3(e.o), dir);
                mt = len2(dir);
                blocked = false;
                at_leaf = false;
                if (visible_wide_begin<FEAT>(S, N, ray, mt, v)) { walking = true; pend = false; }
                else blocked = !visible_through_fog<FEAT>(S, ray, mt, rng, li);   // the segment misses the scene's box: only the medium can hide the light
            }
        }
    }
#ifdef GI_EXP_DIV
    div_flush(N, 1);
#endif
    if constexpr (COUNT) { flush_walk_cnt(sc->shadow, N.wc); flush_u64(&sc->shadow_rays, n_rays); }
}

// Closes the gaps between the workgroups' segments of up to three staging queues (one launch per producer kernel).  Stream k copies
// width[k] 4-byte words per entry from src[k] to dst[k]; entry t of the dense queue is entry (t - prefix[r]) of segment r.  Every
// workgroup recomputes the prefix sums of the (at most 2048) per-workgroup counts in LDS; workgroup 0 leaves the totals in StreamCtl.
// The queue of gather queries is not copied but turned into the input of its sort: (key = photon-map leaf that contains the query's
// position, value = slot) -- PhotonMap::Node::getBounds' descent (gather_find_leaf) runs here, on positions read in queue order.
// Gather queries are sorted by the leaf of the photon octree they fall in; only leaves that have candidate photons matter (a query anywhere else
// adds nothing), so the sort key is the leaf's rank among those -- 15 bits for the 200 000-photon map of the benchmark (27 k such leaves of 72 k
// nodes: two 8-bit digit passes of the radix sort instead of three) -- and every other query gets the one key past them.  One workgroup, at upload.
__global__ __launch_bounds__(1024) void k_pleaf_rank(const PNode* nodes, int32_t n, int32_t* rank, int32_t* inv, int32_t* n_out)
{
    __shared__ int32_t part[1024];
    __shared__ int32_t base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int32_t i0 = 0; i0 < n; i0 += 1024) {
        const int32_t i = i0 + (int32_t)threadIdx.x;
        const int32_t f = (i < n && nodes[i].first_child < 0 && nodes[i].u.lf.nb_photons > 0) ? 1 : 0;
        part[threadIdx.x] = f;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {   // inclusive prefix sums
            const int32_t v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        if (i < n) {
            const int32_t r = base + part[threadIdx.x] - 1;
            rank[i] = f ? r : -1;
            if (f) inv[r] = i;
        }
        __syncthreads();
        if (threadIdx.x == 1023) base += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = base;
}
// The candidate lists written out (k_st_gather stages a leaf's candidates 64 at a time, one per lane: a lane that has to find "candidate number k" by
// walking the leaf's range list reads its range records one after the other; k_st_gather_wave deals a leaf's candidates to the lanes of a wave).
__global__ __launch_bounds__(256) void k_pcand_count(const PNode* nodes, const int32_t* inv, int32_t n_pleaf, uint32_t* cnt)
{
    const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r <= n_pleaf) cnt[r] = r < n_pleaf ? (uint32_t)nodes[inv[r]].u.lf.nb_photons : 0u;
}
__global__ __launch_bounds__(256) void k_pcand_fill(const PNode* nodes, const PRange* pranges, const double* ph_pos, const double* ph_dircol, const int32_t* inv, int32_t n_pleaf, const uint32_t* off,
                                                   double* out_pos, double* out_dc)
{
    // a wave per leaf: its ranges one after the other, a range's photons across the lanes
    const int32_t wave = (int32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = (int32_t)(threadIdx.x & 63u), n_waves = (int32_t)((gridDim.x * blockDim.x) >> 6);
    for (int32_t r = wave; r < n_pleaf; r += n_waves) {
        const PNode& lf = nodes[inv[r]];
        const PRange* ranges = pranges + lf.nb_off;
        uint32_t at = off[r];
        for (int32_t k = 0; k < lf.u.lf.nb_cnt; k++) {
            const PRange rg = ranges[k];
            for (int32_t j = lane; j < rg.count; j += 64) {          // the photon itself, not its index: one read less between a leaf and its candidates
                const size_t src = (size_t)(rg.first + j), dst = (size_t)at + (size_t)j;
                for (int k = 0; k < 3; k++) out_pos[dst * 3 + k] = ph_pos[src * 3 + k];
                for (int k = 0; k < 6; k++) out_dc[dst * 6 + k] = ph_dircol[src * 6 + k];
            }
            at += (uint32_t)rg.count;
        }
    }
}
__global__ __launch_bounds__(256) void k_pdescent(const PNode* nodes, int32_t n, PDescent* out)
{
    const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    PDescent d;
    d.mid[0] = nodes[i].mid[0]; d.mid[1] = nodes[i].mid[1]; d.mid[2] = nodes[i].mid[2];
    d.first_child = nodes[i].first_child; d.pad = 0;
    out[i] = d;
}
// The jump table of gather_find_leaf_fast: for every cell of a grid of GI_PJUMP_N^3 cells over the map's box the node GI_PJUMP_BITS levels down (or the leaf above that) on the way of
// the cell's centre.  *bad is raised when a split plane met on the way is not the grid's plane of that level to within 1e-12 of the extent (the reference
// computes "the middle" in three ways that differ in the last bit, no more): the table is then not used.
__global__ __launch_bounds__(256) void k_pjump(const PDescent* pd, const double* bmin3, const double* cell3, double eps, int32_t* out, int* bad)
{
    const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= GI_PJUMP_N * GI_PJUMP_N * GI_PJUMP_N) return;
    const int cell[3] = {i & (GI_PJUMP_N - 1), i >> (2 * GI_PJUMP_BITS), (i >> GI_PJUMP_BITS) & (GI_PJUMP_N - 1)};       // x fastest, then z, then y
    int32_t node = 0;
    for (int level = 0; level < GI_PJUMP_BITS; level++) {
        const PDescent nd = pd[node];
        if (nd.first_child < 0) break;
        int k = 0;
        for (int ax = 0; ax < 3; ax++) {
            const int top = cell[ax] >> (GI_PJUMP_BITS - 1 - level);                                        // the cell's index among this level's 2^(level+1) halves
            const double plane = bmin3[ax] + (double)(((top >> 1) * 2 + 1) << (GI_PJUMP_BITS - 1 - level)) * cell3[ax];   // the grid's middle of the node's interval
            if (!(fabs(nd.mid[ax] - plane) <= eps)) *bad = 1;
            if (top & 1) k |= ax == 0 ? 1 : (ax == 2 ? 2 : 4);
        }
        node = nd.first_child + k;
    }
    out[i] = node;
}
__global__ void k_find_leaves(Scene S, int32_t n, const double* pos, int32_t* fast, int32_t* full)   // gi_debug_find_leaves
{
    for (int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int32_t)(gridDim.x * blockDim.x)) {
        const V3 p = v3(pos[(size_t)i * 3], pos[(size_t)i * 3 + 1], pos[(size_t)i * 3 + 2]);
        fast[i] = S.pdescent ? gather_find_leaf_fast(S, p) : -2;
        full[i] = gather_find_leaf(S, p);
    }
}
struct CompactStream { const uint32_t* src; uint32_t* dst; int width; };
struct CompactJob {
    CompactStream st[5];    // streams of queue A (e.g. slot + key), then queue B, queue C: n_streams[q] streams each
    int n_streams[3];
    int kind[3];            // QC_* counter of each queue, -1 = unused
    int total_field[3];     // which StreamCtl field receives the total: 0 n_shade, 1 n_cont, 2 n_gather, 3 n_free (+ base), 4 n_free_trace
    int free_base_from_trace;   // queue with total_field 3 appends behind ctl->n_free_trace
    int gather_queue;       // index of the queue whose streams are (slot, position): dst of stream 0 = values, dst of stream 1 = keys; -1 = none
};
__global__ __launch_bounds__(256) void k_st_compact(Scene S, CompactJob job, const unsigned int* bc, const uint32_t* segs, uint32_t n_blocks, StreamCtl* ctl)
{
    __shared__ uint32_t prefix[GI_MAX_PRODUCER_BLOCKS + 1];
    __shared__ uint32_t part[256];
    int stream0 = 0;
    for (int q = 0; q < 3; q++) {
        if (job.kind[q] < 0) { stream0 += job.n_streams[q]; continue; }   // an unused queue still owns its streams: the next queue's begin behind them
        const unsigned int* cnt = bc + (size_t)job.kind[q] * GI_MAX_PRODUCER_BLOCKS * GI_CNT_STRIDE;
        // exclusive prefix sums of the per-workgroup counts: thread t owns counts [t * per, t * per + per)
        const uint32_t per = (n_blocks + 255u) / 256u;
        __syncthreads();
        {
            uint32_t acc = 0;
            for (uint32_t k = 0; k < per; k++) { const uint32_t b = threadIdx.x * per + k; if (b < n_blocks) acc += cnt[(size_t)b * GI_CNT_STRIDE]; }
            part[threadIdx.x] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t acc = 0; for (int t = 0; t < 256; t++) { const uint32_t v = part[t]; part[t] = acc; acc += v; } prefix[n_blocks] = acc; }
        __syncthreads();
        {
            uint32_t acc = part[threadIdx.x];
            for (uint32_t k = 0; k < per; k++) { const uint32_t b = threadIdx.x * per + k; if (b < n_blocks) { prefix[b] = acc; acc += cnt[(size_t)b * GI_CNT_STRIDE]; } }
        }
        __syncthreads();
        const uint32_t total = prefix[n_blocks];
        const uint32_t base = (job.total_field[q] == 3 && job.free_base_from_trace) ? ctl->n_free_trace : 0u;
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
            uint32_t lo = 0, hi = n_blocks;                          // largest r with prefix[r] <= t
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (prefix[mid] <= t) lo = mid; else hi = mid; }
            const size_t from = (size_t)segs[lo] + (t - prefix[lo]), to = (size_t)base + t;
            if (q == job.gather_queue) {
                const CompactStream& sl = job.st[stream0];
                const double* pos = reinterpret_cast<const double*>(job.st[stream0 + 1].src) + from * 3;
                int32_t leaf = S.pdescent ? gather_find_leaf_fast(S, v3(pos[0], pos[1], pos[2])) : -2;
                if (leaf == -2) leaf = gather_find_leaf(S, v3(pos[0], pos[1], pos[2]));
                sl.dst[to] = sl.src[from];
                const int32_t rank = leaf < 0 ? -1 : S.pleaf_rank[leaf];
                job.st[stream0 + 1].dst[to] = rank < 0 ? (uint32_t)S.n_pleaf : (uint32_t)rank;   // the key past the last leaf: nothing to gather
                continue;
            }
            for (int k = 0; k < job.n_streams[q]; k++) {
                const CompactStream& cs = job.st[stream0 + k];
                for (int w = 0; w < cs.width; w++) cs.dst[to * cs.width + w] = cs.src[from * cs.width + w];
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            switch (job.total_field[q]) {
            case 0: ctl->n_shade = total; break;
            case 1: ctl->n_cont = total; break;
            case 2: ctl->n_gather = total; break;
            case 3: ctl->n_free = base + total; break;
            default: ctl->n_free_trace = total; break;
            }
        }
        stream0 += job.n_streams[q];
    }
}

// Gather queries are sorted by the photon-map leaf that contains them (keys from k_st_compact): queries of one leaf share their candidate
// photons, so the lanes of a wave read the same photons and run loops of equal length.
// Queries arrive sorted by leaf, and a leaf typically serves thousands of queries per pass, so most waves hold 64 queries of ONE
// leaf: such a wave copies the leaf's candidate photons into LDS once (64 at a time, one photon per lane, coalesced) and every lane
// scans them from there -- a broadcast LDS read per candidate instead of an L2 round trip per lane.  Waves that straddle a leaf
// boundary take the per-lane walk.  Both run g_key / g_acc / g_end, i.e. the same arithmetic in the same order.
#define GI_GCHUNK 64
#ifndef GI_GATHER_WAVES
#define GI_GATHER_WAVES 4      // waves per SIMD the gather kernel is compiled for (launch bound)
#endif
// Pass 1 of the cooperative path keeps a lane's 32 smallest keys sorted in 32 registers and folds the candidates in 32 at a time:
// sort the 32 new keys (odd-even merge sort network), take min(best[i], new[31 - i]) -- the 32 smallest of the 64, as a bitonic sequence --
// and merge.  18 branch-free instructions per candidate; the LDS heap costs ~55, because with 64 lanes some lane always has to
// sift, so the wave pays a full sift for nearly every candidate.  The result (tau = 32nd smallest float key) is the same number.
__device__ __forceinline__ void kce(float& a, float& b) { const float lo = fminf(a, b), hi = fmaxf(a, b); a = lo; b = hi; }
__device__ __forceinline__ void ksort32(float (&v)[32])   // Batcher's odd-even merge sort: 191 compare-exchanges (the bitonic sorter takes 240)
{
#pragma unroll
    for (int p = 1; p < 32; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j <= 31 - k; j += 2 * k)
#pragma unroll
                for (int i = 0; i <= (k - 1 < 31 - j - k ? k - 1 : 31 - j - k); i++)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) kce(v[i + j], v[i + j + k]);
}
__device__ __forceinline__ void kmerge32(float (&v)[32])   // bitonic sequence -> ascending
{
#pragma unroll
    for (int j = 16; j > 0; j >>= 1)
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int l = i ^ j;
            if (l > i) kce(v[i], v[l]);
        }
}
template <bool COUNT>
__global__ __launch_bounds__(GI_BLOCK, GI_GATHER_WAVES) void k_st_gather(Scene S, PathPool pool, const uint32_t* keys, const uint32_t* vals, uint32_t n_in,
                                                                         const unsigned long long* slot_sample, unsigned long long sample0, double* lbuf, StreamCounters* sc)
{
    unsigned long long n_q = 0, n_c = 0;   // COUNT: queries of this lane, candidates they scanned (a leaf's whole candidate list per query, as PhotonMap::getInRange returns it)
    // 8 KB of LDS per wave: the staged candidates of the cooperative path (64 x 9 doubles) or the float heaps of the per-lane walk (32 x 64),
    // never both at once -- a wave is in one of the two, and the tie pass runs after the last candidate was read
    __shared__ __align__(16) float lds[GI_GATHER_K * GI_BLOCK];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float* const heap = lds + wave * (GI_GATHER_K * 64) + lane;                       // element i of this lane's heap at heap[i * 64]
    double (*const cand)[9] = reinterpret_cast<double (*)[9]>(lds + wave * (GI_GATHER_K * 64));   // cand[k] = staged candidate k of this wave
    // the waves of a workgroup draw its chunks of the grid-stride loop 64 queries at a time (k_st_shade): leaves differ in candidates
    __shared__ unsigned int s_next;
    if (threadIdx.x == 0) s_next = 0u;
    __syncthreads();
    const uint32_t cs = (uint32_t)GI_GATHER_CHUNK;
    const uint32_t total = seg_start(blockIdx.x + 1, n_in, gridDim.x, cs) - seg_start(blockIdx.x, n_in, gridDim.x, cs);
    for (;;) {
        unsigned int u = 0;
        if (lane == 0) u = atomicAdd(&s_next, 64u);
        u = (unsigned int)__shfl((int)u, 0);
        if (u >= total) break;
        const uint32_t i = ((u / cs) * gridDim.x + blockIdx.x) * cs + u % cs + lane;
        const bool valid = i < n_in;
        const uint32_t rank = valid ? keys[i] : 0xffffffffu;                       // rank of the query's leaf among the leaves with candidates
        const bool has_leaf = valid && rank < (uint32_t)S.n_pleaf;
        const uint32_t leaf = has_leaf ? (uint32_t)S.prank_leaf[rank] : 0xffffffffu;
        const uint32_t leaf0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)leaf);
        const bool uniform = __ballot(valid && leaf != leaf0) == 0ull && leaf0 < (uint32_t)S.n_pnode;
        if constexpr (COUNT) { if (valid) n_q++; }
        if (!uniform) {
            if (has_leaf) {
                if constexpr (COUNT) n_c += (unsigned long long)S.pnodes[leaf].u.lf.nb_photons;
                PathRef pr = pool[vals[i]];
                stage_gather_in_leaf(S, pr, (int32_t)leaf, heap, 64, lbuf + (slot_sample[vals[i]] - sample0) * 3);
            }
            continue;
        }
        const uint32_t rank0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rank);
        const PNode& lf = S.pnodes[leaf0];
        const int ncand = lf.u.lf.nb_photons;
        if constexpr (COUNT) { if (valid) n_c += (unsigned long long)ncand; }
        if (ncand == 0) continue;
        const PRange* ranges = S.pranges + lf.nb_off;
        GatherAcc a;
        const uint32_t gslot = valid ? vals[i] : 0u;
        g_begin(a, valid ? ld3(pool.hit[gslot].hpos) : v3(0, 0, 0), valid ? ld3(pool.gath[gslot].gdir) : v3(0, 0, 0), heap, 64, ncand);
        float best[32];
#pragma unroll
        for (int k = 0; k < 32; k++) best[k] = INFINITY;
        for (int pass = 0; pass < 2; pass++) {
            // the leaf's candidates as one sequence (its ranges back to back, the order gather_in_leaf visits them), 64 per step
            for (int32_t c0 = 0; c0 < ncand; c0 += GI_GCHUNK) {
                const int32_t m = min((int32_t)GI_GCHUNK, ncand - c0);
                __builtin_amdgcn_wave_barrier();
                if ((int32_t)lane < m) {
                    const double* pp;
                    const double* dc;
                    if (S.pcand) {
                        const size_t at = (size_t)S.pcand_off[rank0] + (size_t)(c0 + (int32_t)lane);
                        pp = S.pcand + at * 3; dc = S.pcand_dc + at * 6;
                    } else {
                        int32_t off = c0 + (int32_t)lane, r = 0;
                        while (off >= ranges[r].count) { off -= ranges[r].count; r++; }   // r < n_ranges: off < ncand = sum of the counts
                        const size_t ph = (size_t)(ranges[r].first + off);
                        pp = S.ph_pos + ph * 3; dc = S.ph_dircol + ph * 6;
                    }
                    cand[lane][0] = pp[0]; cand[lane][1] = pp[1]; cand[lane][2] = pp[2];
                    if (pass == 1)
                        for (int k = 0; k < 6; k++) cand[lane][3 + k] = dc[k];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (pass == 0) {
                    for (int32_t k0 = 0; k0 < m; k0 += 32) {
                        float nk[32];
#pragma unroll
                        for (int k = 0; k < 32; k++) {
                            const int kk = k0 + k < m ? k0 + k : m - 1;                 // wave-uniform clamp: no out-of-range LDS read
                            const double* q = cand[kk];
                            const float key = (float)len2(v3(q[0], q[1], q[2]) - a.pos);   // same expression as g_key
                            nk[k] = k0 + k < m ? key : INFINITY;
                        }
                        ksort32(nk);
                        // the first 32 keys are the best so far as they are; after the last group only the largest of the 32 smallest is wanted,
                        // which does not need them in order (wave-uniform conditions: the lanes of the wave scan the same candidates)
                        const bool first_group = c0 == 0 && k0 == 0, last_group = c0 + (int32_t)GI_GCHUNK >= ncand && k0 + 32 >= m;
                        if (first_group) {
#pragma unroll
                            for (int k = 0; k < 32; k++) best[k] = nk[k];
                        } else {
#pragma unroll
                            for (int k = 0; k < 32; k++) best[k] = fminf(best[k], nk[31 - k]);
                            if (!last_group) kmerge32(best);
                        }
                    }
                } else if (valid) {
                    // g_acc per candidate, with the rare case set aside: a candidate whose key EQUALS tau (the 32nd smallest itself; one per
                    // query, more only when float keys tie) is some lane's business for nearly every candidate of the wave, so its branch --
                    // a sum of its own, a count, a running maximum -- would run for nearly every candidate.  The loop only notes where such
                    // candidates are; they are added afterwards, in candidate order, by the same g_acc.
                    int n_eq = 0;
                    int32_t k_eq = 0;
                    for (int32_t k = 0; k < m; k++) {
                        const double* q = cand[k];
                        const double d2 = len2(v3(q[0], q[1], q[2]) - a.pos);
                        const float key = (float)d2;
                        if (key < a.tau) {
                            const V3 contrib = v3(q[6], q[7], q[8]) * dot(v3(q[3], q[4], q[5]), a.dir);
                            a.s_lt = a.s_lt + contrib; a.c_lt++;
                        } else if (key == a.tau) {
                            if (n_eq == 0) k_eq = k;
                            n_eq++;
                        }
                    }
                    if (n_eq == 1) {
                        const double* q = cand[k_eq];
                        g_acc(a, v3(q[0], q[1], q[2]), q + 3);
                    } else if (n_eq > 1) {     // float keys tie at tau: walk on from the first of them
                        for (int32_t k = k_eq; n_eq > 0 && k < m; k++) {
                            const double* q = cand[k];
                            if ((float)len2(v3(q[0], q[1], q[2]) - a.pos) == a.tau) { g_acc(a, v3(q[0], q[1], q[2]), q + 3); n_eq--; }
                        }
                    }
                }
            }
            if (pass == 0) {
                // tau = K-th smallest key, K = min(32, ncand) (what the heap's root holds after pass 1 of gather_in_leaf)
                float tau = 0.0f;
                if (ncand < 32) {
#pragma unroll
                    for (int k = 0; k < 32; k++) tau = best[k] < INFINITY ? fmaxf(tau, best[k]) : tau;
                } else {
#pragma unroll
                    for (int k = 0; k < 32; k++) tau = fmaxf(tau, best[k]);   // keys are squared distances: >= 0
                }
                a.tau = tau;
            }
        }
        if (valid) {
            V3 caustic;
            if (!g_end(a, caustic)) caustic = gather_in_leaf(S, (int32_t)leaf0, a.pos, a.dir, heap, 64, nullptr, nullptr);   // float-key tie: exact pass
            double* Lp = lbuf + (slot_sample[vals[i]] - sample0) * 3;   // the path's radiance lives in the per-sample buffer
            V3 L = ld3(Lp) + ld3(pool.gath[gslot].gcoef) * caustic;
            Lp[0] = L.x; Lp[1] = L.y; Lp[2] = L.z;
        }
    }
    if constexpr (COUNT) { flush_u64(&sc->gather_queries, n_q); flush_u64(&sc->gather_cand, n_c); }
}

// A query per WAVE, for the passes with few queries.  Late passes hold a handful of queries per photon-map leaf: a wave of k_st_gather then either serves
// sixty leaves one after the other with one or two lanes at work, or lets every lane walk its own leaf -- either way the pass lasts as long as some
// five thousand candidate steps of one wave (1.2 - 1.6 ms on the benchmark frame, whatever the number of queries).  Here the lanes take a CANDIDATE
// each: 64 keys per step, the 32 smallest kept across the lanes (a bitonic sort of the 64 new keys through lane exchanges, the smaller half against
// the best so far, a bitonic merge: ~110 instructions per 64 candidates instead of 18 per candidate and query), tau read off lane K - 1; the second
// pass computes every lane's contribution and adds those below tau in CANDIDATE ORDER (a lane at a time, read off by v_readlane), so the sums are the
// sums of g_acc to the last bit; the tie group (keys equal to tau) likewise; float-key ties across rank 32 go to lane 0's gather_in_leaf, as everywhere.
__device__ __forceinline__ double wave_read(double v, int src_lane)      // src_lane is wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ V3 gather_wave(const Scene& S, int32_t leaf, uint32_t rank, V3 pos, V3 dir, float* heap32, uint32_t lane, int* n_cand_out)
{
    const PNode& lf = S.pnodes[leaf];
    const int ncand = lf.u.lf.nb_photons;
    if (n_cand_out) *n_cand_out = ncand;
    V3 res = v3(0, 0, 0);
    if (ncand == 0) return res;
    const size_t cbase = (size_t)S.pcand_off[rank];
    const int K = ncand < GI_GATHER_K ? ncand : GI_GATHER_K;
    // pass 1: the K-th smallest float key.  lane j < 32 holds the j-th smallest so far
    float best = INFINITY;
    for (int32_t c0 = 0; c0 < ncand; c0 += 64) {
        const int32_t c = c0 + (int32_t)lane;
        float key = INFINITY;
        if (c < ncand) {
            const double* pp = S.pcand + (cbase + (size_t)c) * 3;
            key = (float)len2(v3(pp[0], pp[1], pp[2]) - pos);            // the expression of g_key
        }
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const float o = __shfl_xor(key, j);
                const bool keep_min = ((lane & (uint32_t)j) == 0u) == ((lane & (uint32_t)k) == 0u);   // ascending blocks where bit k is clear (k = 64: everywhere)
                key = keep_min ? fminf(key, o) : fmaxf(key, o);
            }
        const float r = __shfl(key, (int)((31u - lane) & 63u));         // the 32 smallest new keys, descending over lanes 0 .. 31
        float m = lane < 32u ? fminf(best, r) : INFINITY;               // the 32 smallest of both, a bitonic sequence
#pragma unroll
        for (int j = 16; j > 0; j >>= 1) {
            const float o = __shfl_xor(m, j);
            m = (lane & (uint32_t)j) == 0u ? fminf(m, o) : fmaxf(m, o);
        }
        best = lane < 32u ? m : INFINITY;
    }
    const float tau = __shfl(best, K - 1);
    // pass 2: g_acc over the candidates, in their order
    V3 s_lt = v3(0, 0, 0), s_eq = v3(0, 0, 0);
    int c_lt = 0, c_eq = 0;
    double r_eq = 0;
    for (int32_t c0 = 0; c0 < ncand; c0 += 64) {
        const int32_t c = c0 + (int32_t)lane;
        bool lt = false, eq = false;
        double d2 = 0;
        V3 contrib = v3(0, 0, 0);
        if (c < ncand) {
            const double* pp = S.pcand + (cbase + (size_t)c) * 3;
            d2 = len2(v3(pp[0], pp[1], pp[2]) - pos);
            const float key = (float)d2;
            lt = key < tau; eq = key == tau;
            if (lt || eq) {
                const double* dc = S.pcand_dc + (cbase + (size_t)c) * 6;
                contrib = v3(dc[3], dc[4], dc[5]) * dot(v3(dc[0], dc[1], dc[2]), dir);
            }
        }
        unsigned long long mlt = __ballot(lt), meq = __ballot(eq);
        c_lt += (int)__popcll(mlt); c_eq += (int)__popcll(meq);
        while (mlt != 0ull) {
            const int b = __ffsll((long long)mlt) - 1;
            mlt &= mlt - 1ull;
            s_lt = s_lt + v3(wave_read(contrib.x, b), wave_read(contrib.y, b), wave_read(contrib.z, b));
        }
        while (meq != 0ull) {
            const int b = __ffsll((long long)meq) - 1;
            meq &= meq - 1ull;
            s_eq = s_eq + v3(wave_read(contrib.x, b), wave_read(contrib.y, b), wave_read(contrib.z, b));
            const double de = wave_read(d2, b);
            r_eq = de > r_eq ? de : r_eq;
        }
    }
    const int need = K - c_lt;       // g_end
    if (c_eq > need) {               // float keys tie across rank 32: the exact pass, by one lane
        if (lane == 0u) res = gather_in_leaf(S, leaf, pos, dir, heap32, 1, nullptr, nullptr);
        res = v3(wave_read(res.x, 0), wave_read(res.y, 0), wave_read(res.z, 0));
        return res;
    }
    return (s_lt + s_eq) / (GI_PI * r_eq);
}
#define GI_GW_BLOCK 256      // four waves: a wave per SIMD and workgroup, five workgroups per CU at 87 registers
template <bool COUNT>
__global__ __launch_bounds__(GI_GW_BLOCK) void k_st_gather_wave(Scene S, PathPool pool, const uint32_t* keys, const uint32_t* vals, uint32_t n_in,
                                                            const unsigned long long* slot_sample, unsigned long long sample0, double* lbuf, StreamCounters* sc)
{
    __shared__ float heaps[(GI_GW_BLOCK / 64) * GI_GATHER_K];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long n_q = 0, n_c = 0;
    for (uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; q < n_in; q += n_waves) {
        const uint32_t rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[q]);
        if constexpr (COUNT) n_q++;
        if (rank >= (uint32_t)S.n_pleaf) continue;
        const int32_t leaf = __builtin_amdgcn_readfirstlane(S.prank_leaf[rank]);
        const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)vals[q]);
        int nc = 0;
        const V3 caustic = gather_wave(S, leaf, rank, ld3(pool.hit[slot].hpos), ld3(pool.gath[slot].gdir), heaps + wave * GI_GATHER_K, lane, &nc);
        if constexpr (COUNT) n_c += (unsigned long long)nc;
        if (lane == 0u && nc > 0) {
            double* Lp = lbuf + (slot_sample[slot] - sample0) * 3;   // the path's radiance lives in the per-sample buffer
            const V3 L = ld3(Lp) + ld3(pool.gath[slot].gcoef) * caustic;
            Lp[0] = L.x; Lp[1] = L.y; Lp[2] = L.z;
        }
    }
    if constexpr (COUNT) { if (lane != 0u) { n_q = 0; n_c = 0; } flush_u64(&sc->gather_queries, n_q); flush_u64(&sc->gather_cand, n_c); }
}

// the last stragglers of a chunk (paths bouncing between specular surfaces up to MAX_DEPTH), finished without one nearly
// empty pass per remaining depth.  A wave's bounce costs what its most divergent lanes cost (one depth-65 path alone in a
// wave: ~150 us per bounce; 15 unrelated ones sharing a wave: ~730 us), and the tail of a frame is one dependent chain of
// such bounces, so the finisher runs in stages: stage k gives each wave `lanes` paths (lanes 0..lanes-1, the rest idle),
// advances them at most `max_bounces` vertices, and hands the survivors to stage k+1, which spreads them thinner.  The
// survivor count stays on the device (n_in_dev): no host round trip between stages.
#define GI_FINISH_BLOCK 512   // 8 waves: with 64 KB of heaps and 64 KB of octree records one block fills a CU, at the 2 waves per SIMD the registers allow
// Three forms of a stage, one kernel each: MODE 0 a path per lane (heaps of 32 keys per lane: 64 KB, and the 292 records), MODE 1 a path per wave,
// MODE 2 a path per group of 16 lanes -- the latter two keep the records' content boxes in LDS as well (a lone path's bounce is a chain of dependent
// reads, and the boxes were the one left in L2 on every turn) and one heap per GROUP (its lanes carry the same path and would fill 16 or 64 identical
// heaps).  Which form a stage runs depends on how many paths reach it -- a count that stays on the device -- so all three are launched and two return at once.
#define GI_FINISH_COOP_RECORDS GI_LDS_WNODES
#define GI_FINISH_COOP_HEAP_OFF ((GI_LDS_BOXES_BYTES(GI_FINISH_COOP_RECORDS) + 15) & ~15)
#define GI_FINISH_COOP_LDS_BYTES (GI_FINISH_COOP_HEAP_OFF + (GI_FINISH_BLOCK / 16) * GI_GATHER_K * 4)
// The photon gathers the paths of a finisher wave have pending after a vertex (`pending`: this lane's -- or, with G lanes per path, this group's --
// path has one).  Up to GI_FIN_WAVE_GATHERS of them are served by the whole wave one after the other (gather_wave); more than that (a stage that still
// holds a path in most lanes), or no written-out candidate lists, and every path walks its own leaf as before (stage_gather).  Called by all 64 lanes.
#ifndef GI_FIN_WAVE_GATHERS
#define GI_FIN_WAVE_GATHERS 24
#endif
__device__ __forceinline__ void finish_gathers(const Scene& S, PathRec& p, bool pending, uint32_t G, float* heap, int heap_stride, float* heap32, uint32_t lane)
{
    unsigned long long pend = __ballot(pending && (lane & (G - 1u)) == 0u);
    if (pend == 0ull) return;
    if (!S.pcand || (uint32_t)__popcll(pend) > (uint32_t)GI_FIN_WAVE_GATHERS) {
        if (pending) stage_gather(S, p, heap, heap_stride, nullptr);
        return;
    }
    while (pend != 0ull) {
        const int src = __ffsll((long long)pend) - 1;
        pend &= pend - 1ull;
        const V3 gp = v3(wave_read(p.hpos[0], src), wave_read(p.hpos[1], src), wave_read(p.hpos[2], src));
        const V3 gd = v3(wave_read(p.gdir[0], src), wave_read(p.gdir[1], src), wave_read(p.gdir[2], src));
        int32_t leaf = S.pdescent ? gather_find_leaf_fast(S, gp) : -2;      // the descent k_st_compact keys the queries of a pass with
        if (leaf == -2) leaf = gather_find_leaf(S, gp);
        V3 caustic = v3(0, 0, 0);
        if (leaf >= 0) {
            const int32_t rank = S.pleaf_rank[leaf];
            if (rank >= 0) caustic = gather_wave(S, leaf, (uint32_t)rank, gp, gd, heap32, lane, nullptr);
        }
        if (lane / G == (uint32_t)src / G) {                                  // stage_gather, on the path's own lanes
            const V3 L = ld3(p.L) + ld3(p.gcoef) * caustic;
            p.L[0] = L.x; p.L[1] = L.y; p.L[2] = L.z;
        }
    }
}
__device__ __forceinline__ int finish_mode(int wide, int lanes, uint32_t n_in, uint32_t n_waves, uint32_t coop_factor)
{
    // coop_factor: low half = paths per group-of-16 slot up to which a stage runs one path per group, high half = paths per resident wave up to which it
    // runs one path per WAVE (the wave's lanes walk the nodes together, test a leaf's entities side by side and take a gather candidate each)
    if (!wide || lanes > 0) return 0;
    if (n_in <= (coop_factor >> 16) * n_waves) return 1;
    if (n_in <= (coop_factor & 0xffffu) * 4u * n_waves) return 2;
    return 0;
}
template <int FEAT, int WIDE, int MODE>
__global__ __launch_bounds__(GI_FINISH_BLOCK) void k_st_finish(Scene S, uint64_t seed, PathPool pool, const unsigned long long* slot_sample, unsigned long long sample0,
                                                        const uint32_t* q_in, const unsigned int* n_in_dev, uint32_t n_in_host, int lanes, int max_bounces,
                                                        uint32_t* q_out, unsigned int* n_out, double* lbuf, uint32_t coop_factor)
{
    const uint32_t n_in = n_in_dev ? *n_in_dev : n_in_host;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    if (finish_mode(WIDE, lanes, n_in, n_waves, coop_factor) != MODE) return;   // uniform over the grid
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if constexpr (WIDE != 0 && MODE != 0) {
        constexpr uint32_t G = MODE == 1 ? 64u : 16u, per_wave = 64u / G;
        const LdsWide N = stage_wide_in_lds(S, true, GI_FINISH_COOP_RECORDS);   // ends with a barrier
        LdsWideCoop<(int)G> NC;
        NC.g = N.g; NC.n_l = N.n_l; NC.cboxes = N.cboxes; NC.cuse = N.cuse; NC.n_lc = N.n_lc; NC.box_off = N.box_off; NC.use_off = N.use_off;
        float* const heap = reinterpret_cast<float*>(gi_dyn_lds + GI_FINISH_COOP_HEAP_OFF) + threadIdx.x / G;   // one heap per group, element i at heap[i * groups]
        const int heap_stride = (int)(GI_FINISH_BLOCK / G);
        const uint32_t grp = lane / G;
        // every lane of the wave stays in the loops (a group without a path idles): a pending photon gather is served by the WHOLE wave, one query at a
        // time -- its lanes take a candidate each (gather_wave: gather_in_leaf's sums to the last bit).  A group's or lane's own walk through its leaf's
        // candidates is a chain of dependent reads that the stage's other paths all wait for (0.5 - 1 ms per vertex of a stage, whatever its size).
        float* const heap32 = reinterpret_cast<float*>(gi_dyn_lds + GI_FINISH_COOP_HEAP_OFF) + (threadIdx.x >> 6) * GI_GATHER_K;
        for (uint32_t i0 = wave * per_wave; i0 < n_in; i0 += n_waves * per_wave) {
            const uint32_t i = i0 + grp;
            const bool have = i < n_in;
            const uint32_t slot = have ? q_in[i] : 0u;
            PathRec p;
            double* const Lb = lbuf + (have ? (slot_sample[slot] - sample0) * 3 : 0ull);   // the path's radiance so far; kept in registers while this stage works on it
            if (have) { p = pool.load(slot); p.L[0] = Lb[0]; p.L[1] = Lb[1]; p.L[2] = Lb[2]; }
            bool alive = have, running = have;
            for (int b = 0;;) {
                int fl = 0;
                if (running) {
                    if (!stage_trace_nodes<FEAT>(S, NC, p, seed, nullptr)) { alive = false; running = false; }
                    if (running) fl = stage_shade_nodes<FEAT>(S, NC, p, seed, nullptr);
                }
                finish_gathers(S, p, running && (fl & ST_GATHER) != 0, G, heap, heap_stride, heap32, lane);
                if (running && !(fl & ST_CONTINUE)) { alive = false; running = false; }
                if (++b >= max_bounces) running = false;
                if (__ballot(running) == 0ull) break;
            }
            if (have && (lane & (G - 1u)) == 0u) {
                Lb[0] = p.L[0]; Lb[1] = p.L[1]; Lb[2] = p.L[2];
                if (alive) {
                    pool.store(slot, p);
                    q_out[atomicAdd(n_out, 1u)] = slot;
                }
            }
        }
    } else {
        __shared__ float heap[GI_GATHER_K * GI_FINISH_BLOCK];
        const typename LdsSrc<WIDE>::type N = LdsSrc<WIDE>::stage(S);   // a lone path's bounce is a chain of dependent node reads: LDS, not L2
        if (lanes <= 0) lanes = (int)min(64u, max(1u, (n_in + n_waves - 1) / n_waves));   // spread the paths evenly over the resident waves
        for (uint32_t i0 = wave * lanes; i0 < n_in; i0 += n_waves * lanes) {
            const uint32_t i = i0 + lane;
            const bool have = lane < (uint32_t)lanes && i < n_in;
            const uint32_t slot = have ? q_in[i] : 0u;
            PathRec p;
            double* const Lb = lbuf + (have ? (slot_sample[slot] - sample0) * 3 : 0ull);
            if (have) { p = pool.load(slot); p.L[0] = Lb[0]; p.L[1] = Lb[1]; p.L[2] = Lb[2]; }
            bool alive = have, running = have;
            for (int b = 0;;) {
                int fl = 0;
                if (running) {
                    if (!stage_trace_nodes<FEAT>(S, N, p, seed, nullptr)) { alive = false; running = false; }
                    else fl = stage_shade_nodes<FEAT>(S, N, p, seed, nullptr);
                }
                finish_gathers(S, p, running && (fl & ST_GATHER) != 0, 1u, heap + threadIdx.x, GI_FINISH_BLOCK, heap + (threadIdx.x & ~63u), lane);   // (heap32: row 0 of this wave's own lanes 0 .. 31)
                if (running && !(fl & ST_CONTINUE)) { alive = false; running = false; }
                if (++b >= max_bounces) running = false;
                if (__ballot(running) == 0ull) break;
            }
            if (have) {
                Lb[0] = p.L[0]; Lb[1] = p.L[1]; Lb[2] = p.L[2];
                if (alive) {
                    pool.store(slot, p);
                    q_out[atomicAdd(n_out, 1u)] = slot;
                }
            }
        }
    }
}

// fold samples [s0, s0 + ns) of every pixel into its running mean, in sample order (include/raytracer.h:131-147)
__global__ __launch_bounds__(GI_BLOCK) void k_st_accum(Frame F, PixRec* pix, const double* lbuf, uint32_t n_pix, int ns, void* out, int out_f64, int32_t* out_spp)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += gridDim.x * blockDim.x) {
        int x, ly;
        st_pixel_xy(F, i, x, ly);
        PixRec r = pix[i];
        PixelState ps;
        ps.color = ld3(r.color); ps.lastCol = ld3(r.lastCol); ps.var = r.var; ps.samps = r.samps; ps.s = r.s;
        for (int k = 0; k < ns; k++) {
            const double* l = lbuf + ((size_t)k * n_pix + i) * 3;
            pixel_add_sample(ps, F, v3(l[0], l[1], l[2]));
        }
        r.color[0] = ps.color.x; r.color[1] = ps.color.y; r.color[2] = ps.color.z;
        r.lastCol[0] = ps.lastCol.x; r.lastCol[1] = ps.lastCol.y; r.lastCol[2] = ps.lastCol.z;
        r.var = ps.var; r.samps = ps.samps; r.s = ps.s;
        pix[i] = r;
        const size_t o = ((size_t)ly * F.w + x);
        if (out_f64) {
            double* p = (double*)out + o * 3;
            p[0] = ps.color.x; p[1] = ps.color.y; p[2] = ps.color.z;
        } else {
            float* p = (float*)out + o * 3;
            p[0] = (float)ps.color.x; p[1] = (float)ps.color.y; p[2] = (float)ps.color.z;
        }
        if (out_spp) out_spp[o] = ps.s;
    }
}

__global__ __launch_bounds__(GI_BLOCK) void k_trace(Scene S, int n, const double* rays, int32_t* hit, int32_t* ent, double* res)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = rays + (size_t)i * 6;
    Ray ray = make_ray_exact(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]));
    Rng rng = rng_make(0, (uint32_t)i);
    HitRec h;
    bool ok = trace(S, ray, rng, P_TRACE_ALPHA, h, nullptr);
    hit[i] = ok ? 1 : 0;
    ent[i] = ok ? h.tri : -1;
    double* o = res + (size_t)i * 8;
    if (ok) {
        V3 nn = shading_normal(S, h);
        o[0] = h.pos.x; o[1] = h.pos.y; o[2] = h.pos.z; o[3] = nn.x; o[4] = nn.y; o[5] = nn.z; o[6] = h.u; o[7] = h.v;
    } else
        for (int k = 0; k < 8; k++) o[k] = 0;
}

__global__ __launch_bounds__(GI_BLOCK) void k_visible(Scene S, int n, const double* q, int32_t* vis)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = q + (size_t)i * 6;
    V3 o = v3(p[0], p[1], p[2]), t = v3(p[3], p[4], p[5]);
    V3 ld = t - o;
    double maxt = len2(ld);
    Ray sr = make_ray(o, ld);
    Rng rng = rng_make(0, (uint32_t)i);
    vis[i] = visible(S, sr, maxt, rng, 0, nullptr) ? 1 : 0;
}

__global__ __launch_bounds__(GI_BLOCK) void k_visible_rays(Scene S, int n, const double* rays, const double* mt, int32_t* vis)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = rays + (size_t)i * 6;
    Ray sr = make_ray_exact(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]));
    Rng rng = rng_make(0, (uint32_t)i);
    vis[i] = visible(S, sr, mt[i], rng, 0, nullptr) ? 1 : 0;
}

__global__ __launch_bounds__(GI_BLOCK) void k_gather(Scene S, int n, const double* q, double* res3, int32_t* n_cand)
{
    __shared__ float heap[GI_GATHER_K * GI_BLOCK];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = q + (size_t)i * 6;
    int nc = 0;
    V3 r = gather(S, v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), heap + threadIdx.x, GI_BLOCK, &nc, nullptr);
    res3[(size_t)i * 3] = r.x; res3[(size_t)i * 3 + 1] = r.y; res3[(size_t)i * 3 + 2] = r.z;
    if (n_cand) n_cand[i] = nc;
}

__global__ __launch_bounds__(GI_BLOCK) void k_radiance(Scene S, int n, const double* rays, const uint32_t* stream, uint64_t seed, double* out3)
{
    __shared__ float heap[GI_GATHER_K * GI_BLOCK];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = rays + (size_t)i * 6;
    Ray ray = make_ray_exact(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]));
    V3 L = radiance_path(S, ray, stream[i], seed, heap + threadIdx.x, GI_BLOCK, nullptr);
    out3[(size_t)i * 3] = L.x; out3[(size_t)i * 3 + 1] = L.y; out3[(size_t)i * 3 + 2] = L.z;
}

__global__ __launch_bounds__(GI_BLOCK) void k_emit(Scene S, int count, int max_depth, uint64_t seed, PhotonOut* out, int32_t* stored, int32_t* tries)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count * S.n_light) return;
    int i = j / S.n_light, li = j % S.n_light;
    PhotonOut po;
    int32_t t = 0;
    bool ok = emit_photon(S, i, li, count, max_depth, seed, po, t);
    stored[j] = ok ? 1 : 0;
    tries[j] = t;
    if (ok) out[j] = po;
}

__global__ __launch_bounds__(GI_BLOCK) void k_leaf_order(Scene S, int n, const double* rays, int cap, int32_t* leaf_out, int32_t* n_out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = rays + (size_t)i * 6;
    n_out[i] = leaf_order(S, make_ray_exact(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])), cap, leaf_out + (size_t)i * cap);
}
__global__ __launch_bounds__(GI_BLOCK) void k_kat(int what, int n, const double* in, int in_stride, double* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[9], o[3];
    for (int k = 0; k < 9; k++) a[k] = k < in_stride ? in[(size_t)i * in_stride + k] : 0.0;
    kat_eval(what, a, o);
    for (int k = 0; k < 3; k++) out[(size_t)i * 3 + k] = o[k];
}

__global__ void k_halton(Scene S, int n, const uint32_t* dim, const uint32_t* index, float* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = halton_sample(S, dim[i], index[i]);
}
__global__ void k_halton_index(HaltonEnumD he, int n, const uint32_t* sxy, uint32_t* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = halton_index(he, sxy[i * 3], sxy[i * 3 + 1], sxy[i * 3 + 2]);
}

// ================================================================================================= host side
namespace {

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t upload(const std::vector<T>& v)
    {
        release();
        n = v.size();
        size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        hipError_t e = hipMalloc((void**)&p, bytes);
        if (e != hipSuccess) { p = nullptr; return e; }
        if (n) e = hipMemcpy(p, v.data(), n * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e != hipSuccess) p = nullptr;
        return e;
    }
};

}  // namespace

#define STG_COUNT_MAX 10
struct StreamGrids { int lds_refused = 0; int init = 0, trace = 0, shade = 0, shadow = 0, gather = 0, accum = 0, finish = 0, ad_gen = 0, ad_accum = 0, compact = 0; };

struct gi_ctx {
    int device = 0;
    StreamGrids grids;                // launch grids of the streaming kernels on this context's device
    int wf_grid[6] = {0, 0, 0, 0, 0, 0};
    hipStream_t stream = nullptr;
    std::string err;
    bool have_scene = false;
    Scene S{};
    DevBuf<TNode> d_tnodes;
    DevBuf<WNode> d_wnodes;
    DevBuf<TexD> d_texs;
    DevBuf<TriUV> d_tri_uv;
    DevBuf<unsigned char> d_tex_pixels;
    DevBuf<double> d_tex_lut;
    DevBuf<int32_t> d_wleaf_id;
    DevBuf<float> d_cboxes;           // content boxes of the wide records' children
    DevBuf<uint32_t> d_cuse;
    bool cull_enabled = true;         // gi_set_content_culling
    bool flat_candidates = true;      // GI_FLAT_CANDIDATES=0: k_st_gather finds a leaf's k-th candidate through its range list, no k_st_gather_wave
    DevBuf<uint32_t> d_pcand_off;
    DevBuf<double> d_pcand, d_pcand_dc;
    uint32_t gather_wave_below = 200000u;    // GI_GATHER_WAVE_BELOW: gather passes with fewer queries give every query a wave (k_st_gather_wave)
    bool descent_jump = true;         // GI_DESCENT_JUMP=0: the fast descent of the gather keys starts at the root
    DevBuf<int32_t> d_pjump;          // its jump table (gi_device.h: gather_find_leaf_fast)
    DevBuf<double> d_pjump_aux;
    bool sort_cont = true;            // GI_SORT_CONT=0: continuing rays stay in queue order
    bool sort_shade = true;           // GI_SORT_SHADE=0: the shade stage takes a pass of continuing rays in the order the trace stage left them
    int sort_shade_lo = 5;            // GI_SORT_SHADE_LO: lowest slot bit that sort looks at (32 neighbouring records = 7 KB stay in the trace stage's order)
    int sort_lo_bit = 0;              // GI_SORT_LO_BIT: lowest key bit the sort of the continuing rays looks at (27-bit key: octant, 18 Morton bits, 6 direction bits)
    uint32_t refill_min = 32;         // GI_REFILL_MIN: idle lanes of a wave that make k_st_trace hand out new rays (64: lockstep waves)
    bool wide_enabled = true;         // gi_set_wide_nodes
    bool pn_planes_ok = false;        // the uploaded photon octree qualifies for the one-record-per-level descent
    int32_t n_prange = 0;             // entries of d_pranges in use
    DevBuf<int32_t> d_refs;
    DevBuf<LeafTri> d_leaf_tris;
    DevBuf<double> d_leaf_boxes;          // every leaf reference's own box (gi_device.h: entity_survivors)
    DevBuf<double> d_trace_boxes;         // the closest-hit walk's boxes (gi_device.h: trace_wide_step)
    DevBuf<float> d_tcboxes;              // and the content boxes made of them
    DevBuf<uint32_t> d_tcuse;
    bool lights_clear = false;            // of the scene last uploaded: no entity within a light's radius + the shadow bias of it (gi_layout.h)
    double cut_margin = -1;               // of the scene last uploaded
    bool scene_clipped = false;           // its trace boxes differ from the whole ones
    bool entity_boxes = true;             // GI_ENTITY_BOXES=0: every entity of a leaf is tested, as the reference does
    bool clip_boxes = true;               // GI_CLIP_BOXES=0: the closest-hit walk uses the entities' whole boxes
    bool walk_cut = true;                 // GI_WALK_CUT=0: the closest-hit walk goes on behind its best hit, as the reference does
    DevBuf<TriGeom> d_tris;
    DevBuf<TriShade> d_shade;
    DevBuf<Mat> d_mats;
    DevBuf<LightD> d_lights;
    DevBuf<FogD> d_fogs;
    DevBuf<double> d_fog_grid;
    DevBuf<PNode> d_pnodes;
    DevBuf<PRange> d_pranges;
    DevBuf<int32_t> d_pleaf_rank, d_prank_leaf, d_n_pleaf;
    DevBuf<PDescent> d_pdescent;
    bool fast_descent = true;         // GI_FAST_DESCENT=0: every gather query walks the full photon-octree records
    DevBuf<double> d_ph_pos, d_ph_dircol;
    DevBuf<HaltonDim> d_hdims;
    DevBuf<uint16_t> d_htable;
    DevBuf<unsigned int> d_tile_counter;
    DevBuf<Counters> d_counters;
    // wavefront pipeline workspaces (grown on demand, kept between frames)
    DevBuf<PathRec> d_pool;               // paths of the round-based wavefront pipeline (records)
    DevBuf<uint32_t> d_pixtab;               // k_pixel_table of the frame being rendered
    DevBuf<unsigned char> d_spool;        // paths of the streaming pipeline: field arrays (PathPool), GI_POOL_BYTES_PER_SLOT each
    size_t spool_slots = 0;
    DevBuf<PixRec> d_pix;
    DevBuf<uint32_t> d_q[4];          // trace ping, trace pong, shade, gather
    DevBuf<unsigned int> d_wfcnt;     // [0] shade, [1] next, [2] gather, [3] pixels still wanting samples
    unsigned int* h_wfcnt = nullptr;  // pinned host mirror of d_wfcnt
    DevBuf<double> d_lbuf;            // streaming variant: per-sample radiance of the current chunk
    DevBuf<unsigned long long> d_slot_sample;
    DevBuf<uint32_t> d_qs[6];         // streaming queues: new, cont ping, cont pong, shade, gather, free ping/pong share [5] + d_q
    DevBuf<StreamCtl> d_ctl;
    DevBuf<uint32_t> d_gk[2], d_gv[2];   // gather sort: keys / values, in / out
    DevBuf<uint32_t> d_stage[4];         // staging queues the producers append to, one segment per workgroup (k_st_compact closes the gaps)
    DevBuf<double> d_stage_pos;
    DevBuf<ShadowQ> d_shq;               // shadow queries the shade stage put off (one light, wide records): entry i belongs to item i of the shade queue
    bool defer_shadows = true;           // GI_DEFER_SHADOWS=0: the shade kernel walks its shadow segments itself
    DevBuf<unsigned int> d_blkcnt;       // per-workgroup append counters [QC_KINDS][GI_MAX_PRODUCER_BLOCKS], 128 bytes apart
    DevBuf<uint32_t> d_segs;             // segment start of every producer workgroup
    DevBuf<uint32_t> d_ck[2], d_cv;      // continuing-ray sort: keys in / out, unsorted slots
    DevBuf<unsigned char> d_sort_tmp;
    DevBuf<uint32_t> d_rs_hist;          // gi_sort.inc: [digit][workgroup] counters of a radix pass
    bool rs_attr_set = false;
    StreamCtl* h_ctl = nullptr;
    size_t lbuf_bytes_max = (size_t)16 << 30;
    // per-stage device time of the last streaming frame (HIP events around every launch, same stream)
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_stage;        // stage id of event pair k (events 2k, 2k+1)
    size_t ev_used = 0;
    bool stage_timing = true;
    float stage_ms[STG_COUNT_MAX] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int render_mode = 0;              // 0 wavefront pipeline, 1 megakernel
    size_t pool_slots_max = (size_t)1 << 30;    // upper bound on paths in flight; the actual pool is also bounded by free HBM (render_streaming)
    uint32_t finish_threshold = 1u << 17;   // GI_FINISH_THRESHOLD: paths left when the finisher takes over
    uint32_t wave_factor = 0;         // GI_WAVE_FACTOR: finisher stages with at most this many paths per resident wave run one path per wave; 0 = by the size of the frame (stream_passes)
    uint32_t coop_factor = 4;         // finisher stages with at most 4 x coop_factor x (resident waves) paths run one path per group of 16 lanes (at most 2 x: one per wave)
    // finisher stages {paths per wave (0: spread evenly over the resident waves), max vertices}; the last stage runs to MAX_DEPTH.
    // Measured on the default frame (tools/stripe_probe.py): one full-wave stage 60 ms, this plan 51 ms; on 1/8 of the rows 40 -> 30 ms.
    std::vector<std::pair<int, int>> finish_plan = {{0, 1}, {0, 1}, {0, 1}, {0, 1}, {0, 2}, {0, 2}, {0, 4}, {0, 8}, {0, GI_MAX_DEPTH + 1}};
    DevBuf<unsigned int> d_fin_cnt;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0;
    int last_launches = 0;
    bool count_enabled = false;       // gi_set_counters(ctx, 1): the megakernel counts the reference's visits (per-node walk)
    bool count_stream = false;        // gi_set_counters(ctx, 2): the streaming kernels count what they execute (StreamCounters)
    DevBuf<StreamCounters> d_stream_cnt;
    unsigned long long stream_shaded = 0;   // shaded hits of the last counted frame (sum of the shade queues' lengths)
    Counters last_counters{};
    int n_cu = 256;
};

namespace {

int fail(gi_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}
#define HIP_TRY(c, expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess) return fail((c), GI_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

}  // namespace

__global__ __launch_bounds__(1024) void k_rs_scan(uint32_t* ghist, uint32_t total);   // gi_sort.inc
// the sort key of the gather queries (k_pleaf_rank), for the photon tables c->S points at
static int install_pleaf_rank(gi_ctx* c)
{
    Scene& S = c->S;
    S.pleaf_rank = nullptr; S.prank_leaf = nullptr; S.n_pleaf = 0; S.pdescent = nullptr; S.pjump = nullptr; S.pcand_off = nullptr; S.pcand = nullptr; S.pcand_dc = nullptr;
    if (S.n_pnode <= 0) return GI_OK;
    if (c->d_pleaf_rank.n < (size_t)S.n_pnode) { HIP_TRY(c, c->d_pleaf_rank.alloc((size_t)S.n_pnode)); HIP_TRY(c, c->d_prank_leaf.alloc((size_t)S.n_pnode)); }
    if (!c->d_n_pleaf.p) HIP_TRY(c, c->d_n_pleaf.alloc(1));
    hipLaunchKernelGGL(k_pleaf_rank, dim3(1), dim3(1024), 0, c->stream, S.pnodes, S.n_pnode, c->d_pleaf_rank.p, c->d_prank_leaf.p, c->d_n_pleaf.p);
    HIP_TRY(c, hipGetLastError());
    int32_t n = 0;
    HIP_TRY(c, hipMemcpyAsync(&n, c->d_n_pleaf.p, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    S.pleaf_rank = c->d_pleaf_rank.p; S.prank_leaf = c->d_prank_leaf.p; S.n_pleaf = n;
    S.pcand_off = nullptr; S.pcand = nullptr; S.pcand_dc = nullptr;
    if (c->flat_candidates && n > 0) {
        if (c->d_pcand_off.n < (size_t)n + 1) HIP_TRY(c, c->d_pcand_off.alloc((size_t)n + 1));
        hipLaunchKernelGGL(k_pcand_count, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, c->stream, S.pnodes, c->d_prank_leaf.p, n, c->d_pcand_off.p);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, c->stream, c->d_pcand_off.p, (uint32_t)n + 1u);
        HIP_TRY(c, hipGetLastError());
        uint32_t total = 0;
        HIP_TRY(c, hipMemcpyAsync(&total, c->d_pcand_off.p + n, sizeof total, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (total > 0 && total < 0x7fffffffu) {
            if (c->d_pcand.n < (size_t)total * 3) { HIP_TRY(c, c->d_pcand.alloc((size_t)total * 3)); HIP_TRY(c, c->d_pcand_dc.alloc((size_t)total * 6)); }
            hipLaunchKernelGGL(k_pcand_fill, dim3(1024), dim3(256), 0, c->stream, S.pnodes, S.pranges, S.ph_pos, S.ph_dircol, c->d_prank_leaf.p, n, c->d_pcand_off.p, c->d_pcand.p, c->d_pcand_dc.p);
            HIP_TRY(c, hipGetLastError());
            S.pcand_off = c->d_pcand_off.p; S.pcand = c->d_pcand.p; S.pcand_dc = c->d_pcand_dc.p;
        }
    }
    // the split records of the descent and the map's own box (gather_find_leaf_fast); the one-record-per-level layout only (children from the parent's planes)
    if (c->fast_descent && S.pn_planes) {
        if (c->d_pdescent.n < (size_t)S.n_pnode) HIP_TRY(c, c->d_pdescent.alloc((size_t)S.n_pnode));
        hipLaunchKernelGGL(k_pdescent, dim3((unsigned)((S.n_pnode + 255) / 256)), dim3(256), 0, c->stream, S.pnodes, S.n_pnode, c->d_pdescent.p);
        HIP_TRY(c, hipGetLastError());
        PNode root;
        HIP_TRY(c, hipMemcpyAsync(&root, S.pnodes, sizeof(PNode), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < 3; k++) { S.pmap_bmin[k] = root.bmin[k]; S.pmap_bmax[k] = root.bmax[k]; }
        S.pdescent = c->d_pdescent.p;
        S.pjump = nullptr;
        if (c->descent_jump) {
            double ext = 0.0, host[7];
            for (int k = 0; k < 3; k++) { ext = std::max(ext, root.bmax[k] - root.bmin[k]); S.pjump_cell[k] = (root.bmax[k] - root.bmin[k]) / (double)GI_PJUMP_N; S.pjump_inv[k] = (double)GI_PJUMP_N / (root.bmax[k] - root.bmin[k]); }
            bool ok = ext > 0.0;
            for (int k = 0; k < 3; k++) { host[k] = root.bmin[k]; host[3 + k] = S.pjump_cell[k]; if (!(S.pjump_cell[k] > 0.0) || !std::isfinite(S.pjump_inv[k])) ok = false; }
            if (ok) {
                if (c->d_pjump.n < (size_t)GI_PJUMP_N * GI_PJUMP_N * GI_PJUMP_N) HIP_TRY(c, c->d_pjump.alloc((size_t)GI_PJUMP_N * GI_PJUMP_N * GI_PJUMP_N));
                if (c->d_pjump_aux.n < 8) HIP_TRY(c, c->d_pjump_aux.alloc(8));
                host[6] = 0.0;                                                    // (as an int: the "bad" flag)
                HIP_TRY(c, hipMemcpyAsync(c->d_pjump_aux.p, host, sizeof host, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(k_pjump, dim3((unsigned)((GI_PJUMP_N * GI_PJUMP_N * GI_PJUMP_N + 255) / 256)), dim3(256), 0, c->stream, c->d_pdescent.p, c->d_pjump_aux.p, c->d_pjump_aux.p + 3, 1e-12 * ext, c->d_pjump.p, reinterpret_cast<int*>(c->d_pjump_aux.p + 6));
                HIP_TRY(c, hipGetLastError());
                int bad = 1;
                HIP_TRY(c, hipMemcpyAsync(&bad, c->d_pjump_aux.p + 6, sizeof bad, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (!bad) S.pjump = c->d_pjump.p;
            }
        }
    }
    return GI_OK;
}

#include "gi_sort.inc"
#include "gi_photon_build.inc"

extern "C" {

int gi_create(gi_ctx** out, int device_ordinal)
{
    if (!out) return GI_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GI_E_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return GI_E_INVALID;
    if (hipSetDevice(device_ordinal) != hipSuccess) return GI_E_NO_DEVICE;
    gi_ctx* c = new gi_ctx();
    c->device = device_ordinal;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    std::vector<HaltonDim> dims;
    std::vector<uint16_t> table;
    build_halton_tables(dims, table);
    if (c->d_hdims.upload(dims) != hipSuccess || c->d_htable.upload(table) != hipSuccess || c->d_tile_counter.alloc(1) != hipSuccess ||
        c->d_counters.alloc(1) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return GI_E_HIP;
    }
    c->S.hdims = c->d_hdims.p;
    c->S.htable = c->d_htable.p;
    if (const char* e = getenv("GI_LBUF_MAX_BYTES")) c->lbuf_bytes_max = (size_t)strtoull(e, nullptr, 0);   // per-sample radiance buffer: frames beyond it run in sample chunks
    if (const char* e = getenv("GI_COOP_FACTOR")) c->coop_factor = (uint32_t)strtoul(e, nullptr, 0);
    if (const char* e = getenv("GI_WAVE_FACTOR")) c->wave_factor = std::min<uint32_t>((uint32_t)strtoul(e, nullptr, 0), 0xffffu);
    if (const char* e = getenv("GI_SORT_CONT")) c->sort_cont = atoi(e) != 0;
    if (const char* e = getenv("GI_DESCENT_JUMP")) c->descent_jump = atoi(e) != 0;
    if (const char* e = getenv("GI_FLAT_CANDIDATES")) c->flat_candidates = atoi(e) != 0;
    if (const char* e = getenv("GI_GATHER_WAVE_BELOW")) c->gather_wave_below = (uint32_t)strtoul(e, nullptr, 10);
    if (const char* e = getenv("GI_SORT_SHADE")) c->sort_shade = atoi(e) != 0;
    if (const char* e = getenv("GI_SORT_SHADE_LO")) c->sort_shade_lo = std::max(0, atoi(e));
    if (const char* e = getenv("GI_SORT_LO_BIT")) c->sort_lo_bit = std::min(26, std::max(0, atoi(e)));
    if (const char* e = getenv("GI_DEFER_SHADOWS")) c->defer_shadows = atoi(e) != 0;
    if (const char* e = getenv("GI_FAST_DESCENT")) c->fast_descent = atoi(e) != 0;
    if (const char* e = getenv("GI_ENTITY_BOXES")) c->entity_boxes = atoi(e) != 0;
    if (const char* e = getenv("GI_CLIP_BOXES")) c->clip_boxes = atoi(e) != 0;
    if (const char* e = getenv("GI_WALK_CUT")) c->walk_cut = atoi(e) != 0;
    if (const char* e = getenv("GI_REFILL_MIN")) c->refill_min = (uint32_t)std::min(64, std::max(1, atoi(e)));
    if (const char* e = getenv("GI_FINISH_THRESHOLD")) c->finish_threshold = (uint32_t)strtoul(e, nullptr, 0);   // tuning knobs
    if (const char* e = getenv("GI_FINISH_PLAN")) {   // "lanes:vertices,lanes:vertices,..."
        std::vector<std::pair<int, int>> plan;
        for (const char* q = e; *q;) {
            int l = 0, b = 0, used = 0;
            if (sscanf(q, "%d:%d%n", &l, &b, &used) != 2 || l < 0 || l > 64 || b < 1) { plan.clear(); break; }
            plan.push_back({l, b});
            q += used;
            if (*q == ',') q++;
        }
        if (!plan.empty()) c->finish_plan = plan;
    }
    *out = c;
    return GI_OK;
}

void gi_destroy(gi_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->h_wfcnt) (void)hipHostFree(c->h_wfcnt);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

const char* gi_last_error(const gi_ctx* c) { return c ? c->err.c_str() : "null context"; }

int gi_set_stream(gi_ctx* c, void* s)
{
    if (!c) return GI_E_INVALID;
    c->stream = (hipStream_t)s;
    return GI_OK;
}

// Which of the walks' short cuts are on (all of them leave every result as it is: DESIGN.md section 4): entity boxes; for the closest-hit walk
// the boxes cut to the leaves, and no look behind the best hit.  gi_set_entity_boxes(ctx, 0) turns all three off: the walks then ask what the reference asks.
static void set_walk_shortcuts(gi_ctx* c)
{
    Scene& S = c->S;
    S.leaf_boxes = c->entity_boxes ? c->d_leaf_boxes.p : nullptr;
    S.trace_boxes = !c->entity_boxes ? nullptr : ((c->clip_boxes && c->scene_clipped) ? c->d_trace_boxes.p : c->d_leaf_boxes.p);
    S.cut_margin = (c->entity_boxes && c->walk_cut) ? c->cut_margin : -1.0;
    const bool cut_to_leaves = S.trace_boxes == c->d_trace_boxes.p && S.cboxes && c->d_tcboxes.n == c->d_cboxes.n;
    S.tcboxes = cut_to_leaves ? c->d_tcboxes.p : S.cboxes;
    S.tcuse = cut_to_leaves ? c->d_tcuse.p : S.cuse;
    // segments that end at a light (k_st_shadow): the same boxes, as long as nothing can block a segment inside its last GI_SHADOW_BIAS (gi_device.h: visible_leaf_blocks)
    const bool sh = c->lights_clear && S.trace_boxes == c->d_trace_boxes.p;
    S.shadow_boxes = sh ? S.trace_boxes : S.leaf_boxes;
    S.scboxes = sh ? S.tcboxes : S.cboxes;
    S.scuse = sh ? S.tcuse : S.cuse;
}

int gi_upload_scene(gi_ctx* c, const gi_scene_desc* d)
{
    if (!c) return GI_E_INVALID;
    HostScene H;
    std::string err;
    if (!layout_scene(d, H, err)) return fail(c, GI_E_INVALID, err);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->d_tnodes.upload(H.tnodes));
    HIP_TRY(c, c->d_wnodes.upload(H.wnodes));
    HIP_TRY(c, c->d_wleaf_id.upload(H.wleaf_id));
    HIP_TRY(c, c->d_cboxes.upload(H.cboxes));
    HIP_TRY(c, c->d_cuse.upload(H.cuse));
    HIP_TRY(c, c->d_texs.upload(H.texs));
    HIP_TRY(c, c->d_tri_uv.upload(H.tri_uv));
    HIP_TRY(c, c->d_tex_pixels.upload(H.tex_pixels));
    HIP_TRY(c, c->d_tex_lut.upload(H.tex_lut));
    HIP_TRY(c, c->d_refs.upload(H.refs));
    HIP_TRY(c, c->d_leaf_tris.upload(H.leaf_tris));
    HIP_TRY(c, c->d_leaf_boxes.upload(H.leaf_boxes));
    HIP_TRY(c, c->d_trace_boxes.upload(H.trace_boxes));
    HIP_TRY(c, c->d_tcboxes.upload(H.tcboxes));
    HIP_TRY(c, c->d_tcuse.upload(H.tcuse));
    c->cut_margin = H.cut_margin;
    c->scene_clipped = H.clipped;
    c->lights_clear = H.lights_clear;
    HIP_TRY(c, c->d_tris.upload(H.tris));
    HIP_TRY(c, c->d_shade.upload(H.shade));
    HIP_TRY(c, c->d_mats.upload(H.mats));
    HIP_TRY(c, c->d_lights.upload(H.lights));
    HIP_TRY(c, c->d_fogs.upload(H.fogs));
    HIP_TRY(c, c->d_fog_grid.upload(H.fog_grid));
    Scene& S = c->S;
    S.tnodes = c->d_tnodes.p; S.leaf_refs = c->d_refs.p; S.leaf_tris = c->d_leaf_tris.p; S.tris = c->d_tris.p; S.shade = c->d_shade.p;
    S.mats = c->d_mats.p; S.lights = c->d_lights.p;
    set_walk_shortcuts(c);
    S.n_node = H.n_node; S.n_tri = H.n_tri; S.n_light = H.n_light;
    for (int k = 0; k < 3; k++) { S.root_bmin[k] = H.tnodes[0].bmin[k]; S.root_bmax[k] = H.tnodes[0].bmax[k]; }
    S.n_wnode = (int32_t)H.wnodes.size();
    S.wnodes = (c->wide_enabled && S.n_wnode > 0) ? c->d_wnodes.p : nullptr;
    S.wleaf_id = c->d_wleaf_id.p;
    S.cboxes = (c->cull_enabled && S.wnodes && !H.cboxes.empty()) ? c->d_cboxes.p : nullptr;
    S.cuse = c->d_cuse.p;
    set_walk_shortcuts(c);
    S.tri_uv = c->d_tri_uv.p; S.texs = c->d_texs.p; S.tex_pixels = c->d_tex_pixels.p; S.tex_lut = c->d_tex_lut.p; S.n_tex = H.n_tex();
    S.has_spheres = 0;
    S.fogs = c->d_fogs.p; S.fog_grid = c->d_fog_grid.p; S.n_fog = H.n_fog();
    for (const TriGeom& g : H.tris) if (g.flags & 4u) S.has_spheres = 1;
    for (int k = 0; k < 3; k++) S.ambient[k] = H.ambient[k];
    // the photon map stays as it is: the reference keeps a valid map when the scene is edited and rebuilt (include/raytracer.h:56-72);
    // RayTracer::setScene, which allocates a fresh PhotonMap (include/raytracer.h:38), is gi_upload_scene + gi_clear_photons
    c->have_scene = true;
    return GI_OK;
}

int gi_clear_photons(gi_ctx* c)
{
    if (!c) return GI_E_INVALID;
    Scene& S = c->S;
    S.pnodes = nullptr; S.ph_pos = nullptr; S.ph_dircol = nullptr; S.n_pnode = 0; S.n_photon = 0; S.n_pleaf = 0;
    return GI_OK;
}

int gi_upload_photons(gi_ctx* c, const gi_photon_map_desc* d)
{
    if (!c || !d) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "upload_photons: no scene");
    HostPhotons H;
    std::string err;
    if (!layout_photons(d, H, err)) return fail(c, GI_E_INVALID, err);
    HIP_TRY(c, hipSetDevice(c->device));
    Scene& S = c->S;
    if (H.n_node == 0) {
        S.pnodes = nullptr; S.ph_pos = nullptr; S.ph_dircol = nullptr; S.n_pnode = 0; S.n_photon = 0; S.n_pleaf = 0;
        return GI_OK;
    }
    HIP_TRY(c, c->d_pnodes.upload(H.nodes));
    HIP_TRY(c, c->d_pranges.upload(H.ranges));
    c->n_prange = (int32_t)H.ranges.size();
    HIP_TRY(c, c->d_ph_pos.upload(H.pos));
    HIP_TRY(c, c->d_ph_dircol.upload(H.dircol));
    S.pnodes = c->d_pnodes.p; S.pranges = c->d_pranges.p; S.ph_pos = c->d_ph_pos.p; S.ph_dircol = c->d_ph_dircol.p;
    S.n_pnode = H.n_node; S.n_photon = H.n_photon;
    c->pn_planes_ok = H.planes_ok;
    S.pn_planes = (c->wide_enabled && c->pn_planes_ok) ? 1 : 0;
    return install_pleaf_rank(c);
}

int gi_local_rows(const gi_render_params* p) { return local_rows(p); }

static int grid_for(gi_ctx* c, const void* kernel, size_t dyn_lds = 0, int block = GI_BLOCK)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, dyn_lds) != hipSuccess || per_cu <= 0) per_cu = 1;
    return c->n_cu * per_cu;
}

static int render_megakernel(gi_ctx* c, const Frame& F, void* d_out, int out_is_f64, int32_t* d_spp)
{
    HIP_TRY(c, hipMemsetAsync(c->d_tile_counter.p, 0, sizeof(unsigned int), c->stream));
    if (c->count_enabled) HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(Counters), c->stream));
    const int grid = c->n_cu * 2;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    if (c->count_enabled)
        hipLaunchKernelGGL(k_render<true>, dim3(grid), dim3(GI_BLOCK), 0, c->stream, c->S, F, d_out, out_is_f64, d_spp, c->d_tile_counter.p, c->d_counters.p);
    else
        hipLaunchKernelGGL(k_render<false>, dim3(grid), dim3(GI_BLOCK), 0, c->stream, c->S, F, d_out, out_is_f64, d_spp, c->d_tile_counter.p, c->d_counters.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->last_launches = 1;
    return GI_OK;
}

static int render_wavefront(gi_ctx* c, const Frame& F, void* d_out, int out_is_f64, int32_t* d_spp, volatile const int* cancel)
{
    const uint32_t tiles = (uint32_t)(((F.w + 7) >> 3) * ((F.local_rows + 7) >> 3));
    const uint32_t n_pix = tiles * 64u;   // padded to whole 8x8 tiles (wf_pixel_xy)
    int B = (int)std::min<size_t>(32, std::max<size_t>(1, c->pool_slots_max / n_pix));
    B = std::max(1, std::min(B, std::max(F.max_samples, 1)));
    const size_t slots = (size_t)n_pix * (size_t)B;
    if (slots > 0xfffffff0ull) return fail(c, GI_E_INVALID, "render: frame too large for 32-bit path slots");
    if (c->d_pool.n < slots) HIP_TRY(c, c->d_pool.alloc(slots));
    if (c->d_pix.n < n_pix) HIP_TRY(c, c->d_pix.alloc(n_pix));
    for (int k = 0; k < 4; k++) if (c->d_q[k].n < slots) HIP_TRY(c, c->d_q[k].alloc(slots));
    if (!c->d_wfcnt.p) HIP_TRY(c, c->d_wfcnt.alloc(4));
    if (!c->h_wfcnt) HIP_TRY(c, hipHostMalloc((void**)&c->h_wfcnt, 4 * sizeof(unsigned int), hipHostMallocDefault));
    int &g_init = c->wf_grid[0], &g_gen = c->wf_grid[1], &g_trace = c->wf_grid[2], &g_shade = c->wf_grid[3], &g_gather = c->wf_grid[4], &g_accum = c->wf_grid[5];
    if (!g_trace) {
        g_init = grid_for(c, (const void*)k_wf_init); g_gen = grid_for(c, (const void*)k_wf_gen); g_trace = grid_for(c, (const void*)k_wf_trace);
        g_shade = grid_for(c, (const void*)k_wf_shade); g_gather = grid_for(c, (const void*)k_wf_gather); g_accum = grid_for(c, (const void*)k_wf_accum);
    }
    hipStream_t st = c->stream;
    PathRec* pool = c->d_pool.p;
    unsigned int* cnt = c->d_wfcnt.p;
    int launches = 0;
    HIP_TRY(c, hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(k_wf_init, dim3(g_init), dim3(GI_BLOCK), 0, st, c->d_pix.p, n_pix);
    launches++;
    // 0 samples per pixel still has to write the initial colour: run the accumulate step once in that case
    bool any = F.max_samples > 0 && F.min_samples > 0;
    if (!any) {
        HIP_TRY(c, hipMemsetAsync(cnt, 0, 4 * sizeof(unsigned int), st));
        hipLaunchKernelGGL(k_wf_accum, dim3(g_accum), dim3(GI_BLOCK), 0, st, F, c->d_pix.p, pool, n_pix, B, d_out, out_is_f64, d_spp, cnt + 3);
        launches++;
    }
    while (any) {
        if (cancel && *cancel) { c->last_launches = launches; return fail(c, GI_E_CANCELLED, "render: cancelled"); }
        hipLaunchKernelGGL(k_wf_gen, dim3(g_gen), dim3(GI_BLOCK), 0, st, c->S, F, c->d_pix.p, pool, n_pix, B);
        launches++;
        const uint32_t* q_in = nullptr;
        uint32_t n_in = (uint32_t)slots;
        int ping = 0;
        for (int depth = 0; depth <= GI_MAX_DEPTH && n_in > 0; depth++) {
            HIP_TRY(c, hipMemsetAsync(cnt, 0, 3 * sizeof(unsigned int), st));
            uint32_t* q_next = c->d_q[ping].p;
            hipLaunchKernelGGL(k_wf_trace, dim3(g_trace), dim3(GI_BLOCK), 0, st, c->S, F.seed, pool, q_in, n_in, c->d_q[2].p, cnt + 0);
            hipLaunchKernelGGL(k_wf_shade, dim3(g_shade), dim3(GI_BLOCK), 0, st, c->S, F.seed, pool, c->d_q[2].p, cnt + 0, q_next, cnt + 1, c->d_q[3].p, cnt + 2);
            if (c->S.n_pnode > 0) { hipLaunchKernelGGL(k_wf_gather, dim3(g_gather), dim3(GI_BLOCK), 0, st, c->S, pool, c->d_q[3].p, cnt + 2); launches++; }
            launches += 2;
            HIP_TRY(c, hipMemcpyAsync(c->h_wfcnt, cnt, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            if (getenv("GI_DEBUG_WF")) fprintf(stderr, "[wf] depth %d in %u shade %u next %u gather %u\n", depth, n_in, c->h_wfcnt[0], c->h_wfcnt[1], c->h_wfcnt[2]);
            n_in = c->h_wfcnt[1];
            q_in = q_next;
            ping ^= 1;
        }
        HIP_TRY(c, hipMemsetAsync(cnt + 3, 0, sizeof(unsigned int), st));
        hipLaunchKernelGGL(k_wf_accum, dim3(g_accum), dim3(GI_BLOCK), 0, st, F, c->d_pix.p, pool, n_pix, B, d_out, out_is_f64, d_spp, cnt + 3);
        launches++;
        HIP_TRY(c, hipMemcpyAsync(c->h_wfcnt + 3, cnt + 3, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        any = c->h_wfcnt[3] > 0;
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, st));
    c->last_launches = launches;
    return GI_OK;
}

enum { STG_REGEN = 0, STG_TRACE, STG_SHADE, STG_SORT, STG_GATHER, STG_FINISH, STG_ACCUM, STG_OTHER, STG_SHADOW, STG_COUNT };
static void stage_begin(gi_ctx* c, int stage)
{
    if (!c->stage_timing) return;
    if (c->ev_used + 2 > c->ev_pool.size()) {
        for (int k = 0; k < 2; k++) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->ev_pool.push_back(e); }
    }
    c->ev_stage.push_back(stage);
    (void)hipEventRecord(c->ev_pool[c->ev_used], c->stream);
}
static void stage_end(gi_ctx* c)
{
    if (!c->stage_timing || c->ev_used + 2 > c->ev_pool.size()) return;
    (void)hipEventRecord(c->ev_pool[c->ev_used + 1], c->stream);
    c->ev_used += 2;
}

static const size_t kLdsNodes = (size_t)GI_LDS_NODES * sizeof(TNode);
static const size_t kLdsFinishCoop = (size_t)GI_FINISH_COOP_LDS_BYTES;   // the one-path-per-group forms of the finisher: 292 records + content boxes + a heap per group
static const size_t kLdsWideBoxes = (size_t)GI_LDS_WIDE_BOXES_BYTES;   // wide records + their content boxes: k_st_trace / k_st_shadow, one 1024-thread workgroup per CU
static const StreamGrids& stream_grids(gi_ctx* c)   // per context: one process may drive several devices (gi_group_*)
{
    StreamGrids& g = c->grids;
    if (!g.trace) {
        // more than 64 KB of dynamic LDS has to be asked for, per kernel (and per device: the attribute belongs to the loaded code object)
        const void* big[] = {(const void*)k_st_trace<0, 1, true>, (const void*)k_st_shadow<0, 0, true>, (const void*)k_st_shadow<0, 1, true>,
                             (const void*)k_st_trace<0, 1>, (const void*)k_st_trace<GI_FEAT_SPHERES, 1>, (const void*)k_st_trace<7, 1>,
                             (const void*)k_st_shadow<0, 0>, (const void*)k_st_shadow<GI_FEAT_SPHERES, 0>, (const void*)k_st_shadow<3, 0>, (const void*)k_st_shadow<7, 0>,
                             (const void*)k_st_shadow<0, 1>, (const void*)k_st_shadow<GI_FEAT_SPHERES, 1>, (const void*)k_st_shadow<3, 1>, (const void*)k_st_shadow<7, 1>};
        // a refusal here (a part with less LDS than gfx950's 160 KB per CU) would make every later launch of these kernels fail: say so by name
        for (const void* k : big) if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsWideBoxes) != hipSuccess) g.lds_refused = (int)kLdsWideBoxes;
        const void* coop[] = {(const void*)k_st_finish<0, 1, 1>, (const void*)k_st_finish<0, 1, 2>, (const void*)k_st_finish<GI_FEAT_SPHERES, 1, 1>, (const void*)k_st_finish<GI_FEAT_SPHERES, 1, 2>,
                              (const void*)k_st_finish<3, 1, 1>, (const void*)k_st_finish<3, 1, 2>, (const void*)k_st_finish<7, 1, 1>, (const void*)k_st_finish<7, 1, 2>};
        for (const void* k : coop) if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsFinishCoop) != hipSuccess) g.lds_refused = (int)kLdsFinishCoop;
        g.init = grid_for(c, (const void*)k_wf_init); g.trace = grid_for(c, (const void*)k_st_trace<7, 1>, kLdsWideBoxes, GI_TRACE_BLOCK);
        g.shade = grid_for(c, (const void*)k_st_shade<7, 1, 0>, kLdsNodes, GI_SHADE_BLOCK); g.shadow = grid_for(c, (const void*)k_st_shadow<7, 1>, kLdsWideBoxes, GI_SHADOW_BLOCK); g.gather = grid_for(c, (const void*)k_st_gather<false>); g.accum = grid_for(c, (const void*)k_st_accum);
        g.compact = grid_for(c, (const void*)k_st_compact, 0, 256); g.finish = grid_for(c, (const void*)k_st_finish<7, 1, 0>, kLdsNodes, GI_FINISH_BLOCK); g.ad_gen = grid_for(c, (const void*)k_ad_gen); g.ad_accum = grid_for(c, (const void*)k_ad_accum);
    }
    return g;
}
// everything the pass loop needs for P paths in flight (the radiance buffer is the caller's)
// a few lights, wide records: the shadow walks of the shade stage run in a kernel of their own (ShadowQ)
static bool defers_shadows(const gi_ctx* c) { return c->defer_shadows && c->S.wnodes != nullptr && c->S.n_light >= 1 && c->S.n_light <= 4; }   // one query per light and shaded hit
static int stream_alloc(gi_ctx* c, uint32_t P)
{
    if (c->spool_slots < P) { HIP_TRY(c, c->d_spool.alloc((size_t)P * GI_POOL_BYTES_PER_SLOT)); c->spool_slots = P; }
    if (c->d_slot_sample.n < P) HIP_TRY(c, c->d_slot_sample.alloc(P));
    for (int k = 0; k < 6; k++) if (c->d_qs[k].n < P) HIP_TRY(c, c->d_qs[k].alloc(P));
    if (c->d_q[0].n < P) HIP_TRY(c, c->d_q[0].alloc(P));
    for (int k = 0; k < 2; k++) { if (c->d_gk[k].n < P) HIP_TRY(c, c->d_gk[k].alloc(P)); if (c->d_gv[k].n < P) HIP_TRY(c, c->d_gv[k].alloc(P)); }
    for (int k = 0; k < 2; k++) if (c->d_ck[k].n < P) HIP_TRY(c, c->d_ck[k].alloc(P));
    const size_t PS = (size_t)P + 4096;   // segments are laid out as if every chunk of a producer's loop were full: up to one chunk of slack
    for (int k = 0; k < 4; k++) if (c->d_stage[k].n < PS) HIP_TRY(c, c->d_stage[k].alloc(PS));
    if (c->d_stage_pos.n < PS * 3) HIP_TRY(c, c->d_stage_pos.alloc(PS * 3));
    if (defers_shadows(c) && c->d_shq.n < (size_t)P * (size_t)c->S.n_light) HIP_TRY(c, c->d_shq.alloc((size_t)P * (size_t)c->S.n_light));
    if (!c->d_blkcnt.p) HIP_TRY(c, c->d_blkcnt.alloc((size_t)QC_KINDS * GI_MAX_PRODUCER_BLOCKS * GI_CNT_STRIDE));
    if (!c->d_segs.p) HIP_TRY(c, c->d_segs.alloc(GI_MAX_PRODUCER_BLOCKS));
    if (c->d_cv.n < P) HIP_TRY(c, c->d_cv.alloc(P));
    {
        const size_t need = (size_t)P * 8;   // gi_sort.inc ping-pongs through one more copy of keys and values
        if (c->d_sort_tmp.n < need) HIP_TRY(c, c->d_sort_tmp.alloc(need));
        if (!c->d_rs_hist.p) HIP_TRY(c, c->d_rs_hist.alloc((size_t)GI_RS_MAXBINS * GI_MAX_PRODUCER_BLOCKS));
    }
    if (!c->d_ctl.p) HIP_TRY(c, c->d_ctl.alloc(1));
    if (!c->h_ctl) HIP_TRY(c, hipHostMalloc((void**)&c->h_ctl, sizeof(StreamCtl), hipHostMallocDefault));
    return GI_OK;
}
// The pass loop: trace -> shade -> (keys, sort, gather) -> sort of the continuing rays, until nothing is in flight.  refill(n_free, qf)
// starts up to n_free new paths in the slots of the free list qf (nullptr: slots 0 .. n_free - 1), writes them to c->d_qs[0] and
// returns how many; `exhausted` tells that it will start no more.  A finished path leaves its radiance at lbuf[slot_sample[slot] - sample0].
static int stream_passes(gi_ctx* c, const Frame& F, unsigned long long sample0, double* lbuf, uint32_t n_free,
                         const std::function<uint32_t(uint32_t, const uint32_t*, GenArgs&)>& refill, const bool& exhausted,
                         volatile const int* cancel, int& launches)
{
    // the finisher's one-path-per-wave form: up to 2 paths per resident wave (more of them side by side are faster in groups of 16: benchmark frame's
    // finisher 19.6 ms against 27.3 with 32 per wave; closed box 4.4 against 7.0) -- except for a small frame that gathers photons, such as a rank's share
    // of the benchmark frame on 8 GPUs (66 M samples): its tail is a larger part of it and holds fewer paths, and a lone path's gather is the wave
    // routine's: up to 32 (a 1/8 share 60.6 ms against 64.0; tools/fin_share.sh)
    const bool small_frame = (unsigned long long)F.w * (unsigned long long)F.local_rows * (unsigned long long)std::max(F.max_samples, 1) < 200000000ull;
    const uint32_t wave_factor = c->wave_factor ? c->wave_factor : ((small_frame && c->S.pcand) ? 32u : 2u);
    const StreamGrids& G = stream_grids(c);
    if (G.lds_refused) return fail(c, GI_E_HIP, "render: the device refused " + std::to_string(G.lds_refused) + " bytes of dynamic LDS per workgroup (the traversal kernels are laid out for gfx950's 160 KB per CU)");
    hipStream_t st = c->stream;
    const bool wide = c->S.wnodes != nullptr;
    // executed-work counters: compiled for the instances the BASELINE scenes run (triangles only, no medium, no texture, shadow walks put off)
    const bool counting = c->count_stream;
    if (counting && !(wide && defers_shadows(c) && !c->S.has_spheres && c->S.n_fog == 0 && c->S.n_tex == 0))
        return fail(c, GI_E_STATE, "render: the streaming work counters (gi_set_counters 2) cover triangle scenes without spheres, fog or textures, walked over wide records with one to four lights; use mode 1 (reference visits, megakernel) for this scene");
    const PathPool pool = make_path_pool(c->d_spool.p, c->spool_slots);
    StreamCtl* ctl = c->d_ctl.p;
    uint32_t* q_new = c->d_qs[0].p;
    uint32_t* q_cont[2] = {c->d_qs[1].p, c->d_qs[2].p};
    uint32_t* q_shade = c->d_qs[3].p;
    uint32_t* q_gather = c->d_qs[4].p;
    uint32_t* q_free[2] = {c->d_qs[5].p, c->d_q[0].p};
    uint32_t n_cont = 0;
    const uint32_t* qf = nullptr;
    int ping = 0;
    for (;;) {
        if (cancel && *cancel) { c->last_launches = launches; return fail(c, GI_E_CANCELLED, "render: cancelled"); }
        // new paths of this pass: either samples the trace kernel starts itself (gen, streaming frames) or paths a caller prepared in
        // the new-path queue (adaptive rounds) -- refill() says which by filling gen.n_gen or returning a count
        GenArgs gen;
        memset(&gen, 0, sizeof gen);
        gen.F = F;
        const uint32_t n_prepared = refill(n_free, qf, gen);
        const uint32_t n_new = n_prepared + gen.n_gen;
        if (n_new + n_cont == 0) break;
        uint32_t* qcont_out = q_cont[ping];
        const uint32_t* qcont_in = q_cont[ping ^ 1];
        if (exhausted && n_new == 0 && n_cont <= c->finish_threshold && !counting) {   // (a counted frame runs its stragglers through the counting passes)
            const bool sphf = c->S.has_spheres != 0, fogf = c->S.n_fog > 0, texf = c->S.n_tex > 0;
            if (c->d_fin_cnt.n < 16) HIP_TRY(c, c->d_fin_cnt.alloc(16));
            HIP_TRY(c, hipMemsetAsync(c->d_fin_cnt.p, 0, 16 * sizeof(unsigned int), st));
            const uint32_t* fq_in = qcont_in;
            uint32_t* fq_out = qcont_out;
            const size_t n_stage = std::min<size_t>(c->finish_plan.size(), 15);
            for (size_t k = 0; k < n_stage; k++) {
                const int lanes = c->finish_plan[k].first, vertices = k + 1 == n_stage ? GI_MAX_DEPTH + 1 : c->finish_plan[k].second;
                const unsigned int* n_in_dev = k == 0 ? nullptr : c->d_fin_cnt.p + (k - 1);
                stage_begin(c, STG_FINISH);
                auto fin = [&](auto mode) {
                    constexpr int M = decltype(mode)::value;
                    hipLaunchKernelGGL(texf ? (wide ? k_st_finish<7, 1, M> : k_st_finish<7, 0, 0>) : wide ? (fogf ? k_st_finish<3, 1, M> : (sphf ? k_st_finish<GI_FEAT_SPHERES, 1, M> : k_st_finish<0, 1, M>)) : (fogf ? k_st_finish<3, 0, 0> : (sphf ? k_st_finish<GI_FEAT_SPHERES, 0, 0> : k_st_finish<0, 0, 0>)), dim3(G.finish), dim3(GI_FINISH_BLOCK),
                                       M == 0 ? kLdsNodes : kLdsFinishCoop, st, c->S, F.seed, pool, c->d_slot_sample.p, sample0,
                                       fq_in, n_in_dev, n_cont, lanes, vertices, fq_out, c->d_fin_cnt.p + k, lbuf, (wave_factor << 16) | (c->coop_factor & 0xffffu));
                };
                fin(std::integral_constant<int, 0>());
                if (wide && lanes <= 0) { fin(std::integral_constant<int, 1>()); fin(std::integral_constant<int, 2>()); launches += 2; }
                stage_end(c);
                launches++;
                uint32_t* t = const_cast<uint32_t*>(fq_in); fq_in = fq_out; fq_out = t;   // both are this chunk's continuation queues
            }
            break;
        }
        HIP_TRY(c, hipMemsetAsync(ctl, 0, sizeof(StreamCtl), st));
        uint32_t* qfree_out = q_free[ping];
        const bool sph = c->S.has_spheres != 0, fog = c->S.n_fog > 0, tex = c->S.n_tex > 0;
        if (G.trace > GI_MAX_PRODUCER_BLOCKS || G.shade > GI_MAX_PRODUCER_BLOCKS) return fail(c, GI_E_STATE, "render: more producer workgroups than per-workgroup counters");
        unsigned int* bc = c->d_blkcnt.p;
        const size_t bc_bytes = (size_t)QC_KINDS * GI_MAX_PRODUCER_BLOCKS * GI_CNT_STRIDE * sizeof(unsigned int);
        // trace: hits -> staging 0, finished paths -> staging 1; compacted into the shade queue and the head of the free list
        HIP_TRY(c, hipMemsetAsync(bc, 0, bc_bytes, st));
        StreamCounters* const sc = counting ? c->d_stream_cnt.p : nullptr;
        stage_begin(c, STG_TRACE); hipLaunchKernelGGL(counting ? (k_st_trace<0, 1, true>) : tex ? (wide ? k_st_trace<7, 1> : k_st_trace<7, 0>) : wide ? (sph ? k_st_trace<GI_FEAT_SPHERES, 1> : k_st_trace<0, 1>) : (sph ? k_st_trace<GI_FEAT_SPHERES, 0> : k_st_trace<0, 0>), dim3(G.trace), dim3(GI_TRACE_BLOCK), wide ? kLdsWideBoxes : kLdsNodes, st, c->S, F.seed, pool, c->d_slot_sample.p, sample0, gen, q_new, n_prepared, qcont_in, n_cont, bc, c->d_segs.p,
                           c->d_stage[0].p, c->d_stage[1].p, lbuf, c->refill_min, sc); stage_end(c);
        {
            CompactJob job;
            memset(&job, 0, sizeof job);
            job.st[0] = {c->d_stage[0].p, q_shade, 1}; job.n_streams[0] = 1; job.kind[0] = QC_SHADE; job.total_field[0] = 0;
            job.st[1] = {c->d_stage[1].p, qfree_out, 1}; job.n_streams[1] = 1; job.kind[1] = QC_FREE; job.total_field[1] = 4;
            job.kind[2] = -1; job.gather_queue = -1;
            stage_begin(c, STG_OTHER); hipLaunchKernelGGL(k_st_compact, dim3(G.compact), dim3(256), 0, st, c->S, job, bc, c->d_segs.p, (uint32_t)G.trace, ctl); stage_end(c);
        }
        // A pass of continuing rays leaves the trace stage in the rays' coherence order, which scatters the shade stage's reads and writes of
        // the path records over the whole pool.  The shade queue is put into slot order for it (a radix sort of slot / place pairs); the shadow
        // queries still land at the place the trace stage gave the item, so the shadow walks keep that (coherent) order.
        const uint32_t* q_shade_use = q_shade;
        const uint32_t* q_orig = nullptr;
        if (c->sort_shade && n_cont > 0) {
            int sbits = 1;
            while ((1ull << sbits) < (unsigned long long)c->spool_slots) sbits++;
            uint32_t* const tk = reinterpret_cast<uint32_t*>(c->d_sort_tmp.p);
            const uint32_t bound = n_new + n_cont;      // every item of the trace stage may have hit something
            stage_begin(c, STG_SORT);
            hipLaunchKernelGGL(k_iota, dim3(std::min<uint32_t>((bound + 1023u) / 1024u, 4096u)), dim3(1024), 0, st, c->d_gv[0].p, bound);
            const int rc = rs_sort_pairs(c, q_shade, c->d_gk[1].p, c->d_gv[0].p, c->d_gv[1].p, tk, tk + bound, bound, reinterpret_cast<const uint32_t*>(&ctl->n_shade), std::min(c->sort_shade_lo, sbits - 1), sbits, c->d_rs_hist.p);
            if (rc) return rc;
            stage_end(c);
            q_shade_use = c->d_gk[1].p;
            q_orig = c->d_gv[1].p;
        }
        ShadowQ* const shq = defers_shadows(c) ? c->d_shq.p : nullptr;
        // shade: continuing rays (slot + key) -> staging 0 / 1, gather queries (slot + position) -> staging 2 / pos, finished paths -> staging 3
        HIP_TRY(c, hipMemsetAsync(bc, 0, bc_bytes, st));
        const bool many = c->S.n_light > 1;
        auto shade_kernel = shq ? (many ? (tex ? k_st_shade<7, 1, 2> : fog ? k_st_shade<3, 1, 2> : sph ? k_st_shade<GI_FEAT_SPHERES, 1, 2> : k_st_shade<0, 1, 2>)
                                        : (tex ? k_st_shade<7, 1, 1> : fog ? k_st_shade<3, 1, 1> : sph ? k_st_shade<GI_FEAT_SPHERES, 1, 1> : k_st_shade<0, 1, 1>))
                                : tex ? (wide ? k_st_shade<7, 1, 0> : k_st_shade<7, 0, 0>) : wide ? (fog ? k_st_shade<3, 1, 0> : (sph ? k_st_shade<GI_FEAT_SPHERES, 1, 0> : k_st_shade<0, 1, 0>)) : (fog ? k_st_shade<3, 0, 0> : (sph ? k_st_shade<GI_FEAT_SPHERES, 0, 0> : k_st_shade<0, 0, 0>));
        stage_begin(c, STG_SHADE); hipLaunchKernelGGL(shade_kernel, dim3(G.shade), dim3(GI_SHADE_BLOCK), kLdsNodes, st, c->S, F.seed, pool, c->d_slot_sample.p, sample0, q_shade_use, ctl, bc, c->d_segs.p,
                           c->d_stage[0].p, c->d_stage[1].p, c->d_stage[2].p, c->d_stage_pos.p, c->d_stage[3].p, lbuf, shq, q_orig);
        stage_end(c);
        if (shq) {   // the walks it put off; before the gather of the same vertices (the order in which a path's radiance is summed)
            stage_begin(c, STG_SHADOW);
            hipLaunchKernelGGL(counting ? (many ? (k_st_shadow<0, 1, true>) : (k_st_shadow<0, 0, true>))
                               : many ? (tex ? k_st_shadow<7, 1> : fog ? k_st_shadow<3, 1> : sph ? k_st_shadow<GI_FEAT_SPHERES, 1> : k_st_shadow<0, 1>)
                                      : (tex ? k_st_shadow<7, 0> : fog ? k_st_shadow<3, 0> : sph ? k_st_shadow<GI_FEAT_SPHERES, 0> : k_st_shadow<0, 0>), dim3(G.shadow), dim3(GI_SHADOW_BLOCK), kLdsWideBoxes, st,
                               c->S, F.seed, pool, shq, ctl, lbuf, c->refill_min, sc);
            stage_end(c);
            launches++;
        }
        {
            CompactJob job;
            memset(&job, 0, sizeof job);
            // streams in queue order: [0] continuing slot, [1] its coherence key, [2] gather slot, [3] gather position (3 doubles = 6 words), [4] freed slot
            job.st[0] = {c->d_stage[0].p, c->d_cv.p, 1}; job.st[1] = {c->d_stage[1].p, c->d_ck[0].p, 1};
            job.n_streams[0] = 2; job.kind[0] = QC_CONT; job.total_field[0] = 1;
            job.st[2] = {c->d_stage[2].p, c->d_gv[0].p, 1};                                                     // gather: slot -> values of the sort by leaf
            job.st[3] = {reinterpret_cast<const uint32_t*>(c->d_stage_pos.p), c->d_gk[0].p, 6};                 //         position -> key (leaf)
            job.n_streams[1] = 2; job.kind[1] = QC_GATHER; job.total_field[1] = 2; job.gather_queue = c->S.n_pnode > 0 ? 1 : -1;
            if (c->S.n_pnode <= 0) job.kind[1] = -1;                                                            // no photon map: no gather queries
            job.st[4] = {c->d_stage[3].p, qfree_out, 1};
            job.n_streams[2] = 1; job.kind[2] = QC_FREE; job.total_field[2] = 3; job.free_base_from_trace = 1;
            stage_begin(c, STG_OTHER); hipLaunchKernelGGL(k_st_compact, dim3(G.compact), dim3(256), 0, st, c->S, job, bc, c->d_segs.p, (uint32_t)G.shade, ctl); stage_end(c);
        }
        launches += 4;
        HIP_TRY(c, hipMemcpyAsync(c->h_ctl, ctl, sizeof(StreamCtl), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        const uint32_t n_gather = c->h_ctl->n_gather;
        if (counting) c->stream_shaded += c->h_ctl->n_shade;
        if (c->S.n_pnode > 0 && n_gather > 0) {
            int bits = 1;
            while ((1u << bits) <= (uint32_t)c->S.n_pleaf) bits++;   // keys 0 .. n_pleaf
            stage_begin(c, STG_SORT);
            {
                uint32_t* const tk = reinterpret_cast<uint32_t*>(c->d_sort_tmp.p);
                const int rc = rs_sort_pairs(c, c->d_gk[0].p, c->d_gk[1].p, c->d_gv[0].p, c->d_gv[1].p, tk, tk + n_gather, n_gather, nullptr, 0, bits, c->d_rs_hist.p);
                if (rc) return rc;
            }
            stage_end(c);
            stage_begin(c, STG_GATHER);
            if (c->S.pcand && n_gather < c->gather_wave_below)      // few queries: a wave each
                hipLaunchKernelGGL(counting ? (k_st_gather_wave<true>) : (k_st_gather_wave<false>), dim3(std::min<uint32_t>((uint32_t)(5 * c->n_cu), (n_gather + 3u) / 4u)), dim3(GI_GW_BLOCK), 0, st, c->S, pool, c->d_gk[1].p, c->d_gv[1].p, n_gather, c->d_slot_sample.p, sample0, lbuf, sc);
            else
                hipLaunchKernelGGL(counting ? (k_st_gather<true>) : (k_st_gather<false>), dim3(G.gather), dim3(GI_BLOCK), 0, st, c->S, pool, c->d_gk[1].p, c->d_gv[1].p, n_gather, c->d_slot_sample.p, sample0, lbuf, sc);
            stage_end(c);
            launches += 2;
        }
        n_cont = c->h_ctl->n_cont;
        n_free = c->h_ctl->n_free;
        if (n_cont > 0) {   // continuing rays in coherence order for the next trace pass
            stage_begin(c, STG_SORT);
            if (c->sort_cont) {
                uint32_t* const tk = reinterpret_cast<uint32_t*>(c->d_sort_tmp.p);
                const int rc = rs_sort_pairs(c, c->d_ck[0].p, c->d_ck[1].p, c->d_cv.p, qcont_out, tk, tk + n_cont, n_cont, nullptr, c->sort_lo_bit, 27, c->d_rs_hist.p);
                if (rc) return rc;
            }
            else HIP_TRY(c, hipMemcpyAsync(qcont_out, c->d_cv.p, (size_t)n_cont * 4, hipMemcpyDeviceToDevice, st));   // GI_SORT_CONT=0: queue order (tuning aid)
            stage_end(c);
            launches++;
        }
        qf = qfree_out;
        ping ^= 1;
        if (getenv("GI_DEBUG_WF")) {
            fprintf(stderr, "[st] new %u cont %u free %u gather %u\n", n_new, n_cont, n_free, n_gather);
            if (counting) {   // what this pass executed (tuning aid): cumulative counters, printed per pass
                StreamCounters h;
                if (hipMemcpy(&h, c->d_stream_cnt.p, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
                    fprintf(stderr, "[cnt] trace rays %llu walks %llu records %llu boxes %llu cboxes %llu leaves %llu eboxes %llu tris %llu | shadow rays %llu records %llu boxes %llu cboxes %llu leaves %llu eboxes %llu tris %llu | gather q %llu cand %llu\n",
                            h.trace_rays, h.trace[0], h.trace[1], h.trace[2], h.trace[3], h.trace[4], h.trace[6], h.trace[5], h.shadow_rays, h.shadow[1], h.shadow[2], h.shadow[3], h.shadow[4], h.shadow[6], h.shadow[5], h.gather_queries, h.gather_cand);
            }
        }
    }
    return GI_OK;
}

static int render_streaming(gi_ctx* c, const Frame& F, void* d_out, int out_is_f64, int32_t* d_spp, volatile const int* cancel)
{
    const uint32_t n_pix = (uint32_t)F.w * (uint32_t)F.local_rows;   // valid pixels only, enumerated in 8x8-tile order (st_pixel_xy)
    const int spp = F.max_samples;
    // Paths in flight: as many as fit -- the whole frame when HBM allows (1080p x 256 spp = 531 M paths = 119 GB of PathRec on a
    // 288 GB part).  More paths per pass = fewer passes and, above all, better-sorted (more coherent) queues.
    size_t slots_budget = c->pool_slots_max;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t held = c->d_spool.n + c->d_lbuf.n * 8 + (c->d_qs[0].n + c->d_q[0].n) * 4 * 7 + c->d_shq.n * sizeof(ShadowQ);   // ours, re-usable
            const size_t per_slot = GI_POOL_BYTES_PER_SLOT + 8 + 13 * 4 + 24 + 40 + (defers_shadows(c) ? sizeof(ShadowQ) * (size_t)c->S.n_light : 0);   // record, sample id, 13 queue / key words, sort scratch, staging queues, shadow queries
            const size_t lbuf = (size_t)n_pix * (size_t)std::min<size_t>((size_t)spp, c->lbuf_bytes_max / ((size_t)n_pix * 24)) * 24;
            const size_t avail = (size_t)((double)(free_b + held) * 0.90);
            if (avail > lbuf) slots_budget = std::min(slots_budget, (avail - lbuf) / per_slot);
            else slots_budget = std::min<size_t>(slots_budget, 1u << 20);
        }
    }
    const uint32_t P = (uint32_t)std::max<size_t>(64, std::min<size_t>(std::min<size_t>(slots_budget, 0xfffffff0u), (size_t)n_pix * (size_t)spp));
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)spp, c->lbuf_bytes_max / ((size_t)n_pix * 24)));
    int rc = stream_alloc(c, P);
    if (rc) return rc;
    if (c->d_pix.n < n_pix) HIP_TRY(c, c->d_pix.alloc(n_pix));
    if (c->d_lbuf.n < (size_t)n_pix * chunk * 3) HIP_TRY(c, c->d_lbuf.alloc((size_t)n_pix * chunk * 3));
    const StreamGrids& G = stream_grids(c);
    hipStream_t st = c->stream;
    int launches = 0;
    c->ev_used = 0; c->ev_stage.clear();
    if (c->count_stream) {
        if (!c->d_stream_cnt.p) HIP_TRY(c, c->d_stream_cnt.alloc(1));
        HIP_TRY(c, hipMemsetAsync(c->d_stream_cnt.p, 0, sizeof(StreamCounters), st));
        c->stream_shaded = 0;
    }
    HIP_TRY(c, hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(k_wf_init, dim3(G.init), dim3(GI_BLOCK), 0, st, c->d_pix.p, n_pix);
    if (c->d_pixtab.n < n_pix) HIP_TRY(c, c->d_pixtab.alloc(n_pix));
    hipLaunchKernelGGL(k_pixel_table, dim3(G.init), dim3(GI_BLOCK), 0, st, F, n_pix, c->d_pixtab.p);
    launches += 2;
    for (int s0 = 0; s0 < spp; s0 += chunk) {
        const int ns = std::min(chunk, spp - s0);
        const unsigned long long sample0 = (unsigned long long)s0 * n_pix, sample_end = (unsigned long long)(s0 + ns) * n_pix;
        unsigned long long next = sample0;
        bool exhausted = false;
        auto refill = [&](uint32_t n_free, const uint32_t* qf, GenArgs& gen) -> uint32_t {   // path regeneration: free slots take the next samples
            const uint32_t n_new = (uint32_t)std::min<unsigned long long>(n_free, sample_end - next);
            gen.q_free = qf; gen.n_gen = n_new; gen.n_pix = n_pix; gen.id_base = next; gen.sample_begin = sample0; gen.s_begin = s0;
            gen.pixtab = c->d_pixtab.p; gen.inv_n_pix = 1.0 / (double)n_pix;
            next += n_new;
            exhausted = next >= sample_end;
            return 0;                                                                          // nothing prepared: the trace kernel starts them
        };
        rc = stream_passes(c, F, sample0, c->d_lbuf.p, P, refill, exhausted, cancel, launches);   // pass 0: every slot is free
        if (rc) return rc;
        stage_begin(c, STG_ACCUM); hipLaunchKernelGGL(k_st_accum, dim3(G.accum), dim3(GI_BLOCK), 0, st, F, c->d_pix.p, c->d_lbuf.p, n_pix, ns, d_out, out_is_f64, d_spp); stage_end(c);
        launches++;
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, st));
    c->last_launches = launches;
    return GI_OK;
}

// Adaptive sampling (min_samples != max_samples, include/raytracer.h:108-148): rounds as in render_wavefront -- a pixel starts the
// samples it is certain to take, the variance rule is applied in sample order -- but the paths of a round run through the streaming
// passes (sorted queues, octree records in LDS, wave-cooperative gather, staged finisher) instead of one generic kernel per depth.
static int render_adaptive(gi_ctx* c, const Frame& F, void* d_out, int out_is_f64, int32_t* d_spp, volatile const int* cancel)
{
    const uint32_t tiles = (uint32_t)(((F.w + 7) >> 3) * ((F.local_rows + 7) >> 3));
    const uint32_t n_pix = tiles * 64u;   // padded to whole 8x8 tiles (wf_pixel_xy)
    int B = (int)std::min<size_t>(32, std::max<size_t>(1, c->pool_slots_max / n_pix));
    B = std::max(1, std::min(B, std::max(F.max_samples, 1)));
    const size_t slots = (size_t)n_pix * (size_t)B;
    if (slots > 0xfffffff0ull) return fail(c, GI_E_INVALID, "render: frame too large for 32-bit path slots");
    int rc = stream_alloc(c, (uint32_t)slots);
    if (rc) return rc;
    if (c->d_pix.n < n_pix) HIP_TRY(c, c->d_pix.alloc(n_pix));
    if (c->d_lbuf.n < slots * 3) HIP_TRY(c, c->d_lbuf.alloc(slots * 3));
    if (!c->d_wfcnt.p) HIP_TRY(c, c->d_wfcnt.alloc(4));
    if (!c->h_wfcnt) HIP_TRY(c, hipHostMalloc((void**)&c->h_wfcnt, 4 * sizeof(unsigned int), hipHostMallocDefault));
    const StreamGrids& G = stream_grids(c);
    hipStream_t st = c->stream;
    unsigned int* cnt = c->d_wfcnt.p;
    int launches = 0;
    c->ev_used = 0; c->ev_stage.clear();
    HIP_TRY(c, hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(k_wf_init, dim3(G.init), dim3(GI_BLOCK), 0, st, c->d_pix.p, n_pix);
    launches++;
    bool any = F.max_samples > 0 && F.min_samples > 0;
    if (!any) {   // 0 samples per pixel still has to write the initial colour
        HIP_TRY(c, hipMemsetAsync(cnt, 0, 4 * sizeof(unsigned int), st));
        hipLaunchKernelGGL(k_ad_accum, dim3(G.ad_accum), dim3(GI_BLOCK), 0, st, F, c->d_pix.p, c->d_lbuf.p, n_pix, B, d_out, out_is_f64, d_spp, cnt + 3);
        launches++;
    }
    while (any) {
        if (cancel && *cancel) { c->last_launches = launches; return fail(c, GI_E_CANCELLED, "render: cancelled"); }
        HIP_TRY(c, hipMemsetAsync(cnt, 0, 4 * sizeof(unsigned int), st));
        stage_begin(c, STG_REGEN);
        hipLaunchKernelGGL(k_ad_gen, dim3(G.ad_gen), dim3(GI_BLOCK), 0, st, c->S, F, c->d_pix.p, make_path_pool(c->d_spool.p, c->spool_slots), c->d_slot_sample.p, n_pix, B, c->d_qs[0].p, cnt + 0);
        stage_end(c);
        launches++;
        HIP_TRY(c, hipMemcpyAsync(c->h_wfcnt, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        uint32_t pending = c->h_wfcnt[0];          // paths started by this round, already in the new-path queue
        const bool exhausted = true;
        auto refill = [&](uint32_t, const uint32_t*, GenArgs&) -> uint32_t { const uint32_t n = pending; pending = 0; return n; };
        rc = stream_passes(c, F, 0ull, c->d_lbuf.p, 0u, refill, exhausted, cancel, launches);
        if (rc) return rc;
        stage_begin(c, STG_ACCUM);
        hipLaunchKernelGGL(k_ad_accum, dim3(G.ad_accum), dim3(GI_BLOCK), 0, st, F, c->d_pix.p, c->d_lbuf.p, n_pix, B, d_out, out_is_f64, d_spp, cnt + 3);
        stage_end(c);
        launches++;
        HIP_TRY(c, hipMemcpyAsync(c->h_wfcnt + 3, cnt + 3, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        any = c->h_wfcnt[3] > 0;
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, st));
    c->last_launches = launches;
    return GI_OK;
}

int gi_render_device(gi_ctx* c, const gi_render_params* p, void* d_out, int out_is_f64, int32_t* d_spp, volatile const int* cancel)
{
    if (!c || !d_out) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "render: no scene uploaded");
    Frame F;
    std::string ferr;
    if (!make_frame(p, F, ferr)) return fail(c, GI_E_INVALID, ferr);
    if (cancel && *cancel) return fail(c, GI_E_CANCELLED, "render: cancelled");
    HIP_TRY(c, hipSetDevice(c->device));
    c->last_ms = 0; c->last_launches = 0;
    if (F.local_rows == 0) return GI_OK;
    if (c->render_mode == 1 || c->count_enabled) return render_megakernel(c, F, d_out, out_is_f64, d_spp);
    // fixed sample count: streaming pool with path regeneration; adaptive sampling: rounds (sample-order decisions) on the same passes;
    // mode 2: the plain per-depth rounds, kept as a second schedule of the same arithmetic
    if (c->count_stream && !(c->render_mode == 0 && F.min_samples == F.max_samples && F.max_samples > 0))
        return fail(c, GI_E_STATE, "render: the streaming work counters (gi_set_counters 2) belong to fixed-sample-count frames of the wavefront pipeline");
    if (c->render_mode == 0 && F.min_samples == F.max_samples && F.max_samples > 0) return render_streaming(c, F, d_out, out_is_f64, d_spp, cancel);
    if (c->render_mode == 0) return render_adaptive(c, F, d_out, out_is_f64, d_spp, cancel);
    return render_wavefront(c, F, d_out, out_is_f64, d_spp, cancel);
}

int gi_set_wide_nodes(gi_ctx* c, int enable)
{
    if (!c) return GI_E_INVALID;
    c->wide_enabled = enable != 0;
    c->S.wnodes = (c->wide_enabled && c->S.n_wnode > 0) ? c->d_wnodes.p : nullptr;
    c->S.cboxes = (c->cull_enabled && c->S.wnodes && c->d_cboxes.n > 1) ? c->d_cboxes.p : nullptr;
    set_walk_shortcuts(c);
    c->S.pn_planes = (c->wide_enabled && c->pn_planes_ok) ? 1 : 0;   // the photon octree's counterpart (gather_find_leaf)
    c->S.pdescent = (c->S.pn_planes && c->fast_descent && c->S.n_pnode > 0 && c->d_pdescent.n >= (size_t)c->S.n_pnode) ? c->d_pdescent.p : nullptr;
    return (c->S.wnodes ? 1 : 0) | (c->S.pn_planes ? 2 : 0);
}

int gi_set_content_culling(gi_ctx* c, int enable)
{
    if (!c) return GI_E_INVALID;
    c->cull_enabled = enable != 0;
    c->S.cboxes = (c->cull_enabled && c->S.wnodes && c->d_cboxes.n > 1) ? c->d_cboxes.p : nullptr;
    set_walk_shortcuts(c);
    return c->S.cboxes ? 1 : 0;
}

int gi_set_entity_boxes(gi_ctx* c, int enable)
{
    if (!c) return GI_E_INVALID;
    c->entity_boxes = enable != 0;
    if (c->have_scene) set_walk_shortcuts(c);
    return c->S.leaf_boxes ? 1 : 0;
}

int gi_set_render_mode(gi_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 2) return GI_E_INVALID;
    c->render_mode = mode;
    return GI_OK;
}

int gi_set_pool_slots(gi_ctx* c, int64_t slots)
{
    if (!c || slots < 64) return GI_E_INVALID;
    c->pool_slots_max = (size_t)slots;
    return GI_OK;
}

int gi_last_render_ms(gi_ctx* c, float* ms, int32_t* n_launches)
{
    if (!c) return GI_E_INVALID;
    if (c->last_launches > 0) {
        HIP_TRY(c, hipEventSynchronize(c->ev1));
        HIP_TRY(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    }
    for (int k = 0; k < STG_COUNT_MAX; k++) c->stage_ms[k] = 0;
    for (size_t k = 0; k + 1 < c->ev_used + 1 && k / 2 < c->ev_stage.size() && k + 1 < c->ev_pool.size() && k < c->ev_used; k += 2) {
        float t = 0;
        if (hipEventElapsedTime(&t, c->ev_pool[k], c->ev_pool[k + 1]) == hipSuccess) c->stage_ms[c->ev_stage[k / 2]] += t;
        if (getenv("GI_DEBUG_STAGES")) fprintf(stderr, "[stage] %d %.3f ms\n", c->ev_stage[k / 2], t);
    }
    if (ms) *ms = c->last_ms;
    if (n_launches) *n_launches = c->last_launches;
    return GI_OK;
}

int gi_last_stage_ms(gi_ctx* c, float* out8)
{
    if (!c || !out8) return GI_E_INVALID;
    int rc = gi_last_render_ms(c, nullptr, nullptr);
    if (rc) return rc;
    for (int k = 0; k < 8; k++) out8[k] = c->stage_ms[k];
    out8[STG_SHADE] += c->stage_ms[STG_SHADOW];   // the shade stage of the 8-entry form includes the shadow walks it put off
    return GI_OK;
}

int gi_last_kernel_ms(gi_ctx* c, float* out10)
{
    if (!c || !out10) return GI_E_INVALID;
    int rc = gi_last_render_ms(c, nullptr, nullptr);
    if (rc) return rc;
    for (int k = 0; k < STG_COUNT_MAX; k++) out10[k] = c->stage_ms[k];
    return GI_OK;
}

int gi_set_counters(gi_ctx* c, int enable)
{
    if (!c || enable < 0 || enable > 2) return GI_E_INVALID;
    c->count_enabled = enable == 1;
    c->count_stream = enable == 2;
    return GI_OK;
}
int gi_get_stream_counters(gi_ctx* c, int64_t* out17 /* [19] */)
{
    if (!c || !out17) return GI_E_INVALID;
    if (!c->d_stream_cnt.p) return fail(c, GI_E_STATE, "stream counters: no counted frame was rendered (gi_set_counters(ctx, 2), fixed sample count)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    StreamCounters h;
    HIP_TRY(c, hipMemcpy(&h, c->d_stream_cnt.p, sizeof h, hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; k++) { out17[k] = (int64_t)h.trace[k]; out17[7 + k] = (int64_t)h.shadow[k]; }
    out17[6] = (int64_t)h.trace_rays; out17[13] = (int64_t)h.shadow_rays;
    out17[14] = (int64_t)h.gather_queries; out17[15] = (int64_t)h.gather_cand; out17[16] = (int64_t)c->stream_shaded;
    out17[17] = (int64_t)h.trace[6]; out17[18] = (int64_t)h.shadow[6];
    return GI_OK;
}
int gi_get_counters(gi_ctx* c, int64_t* out8)
{
    if (!c || !out8) return GI_E_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    Counters h;
    HIP_TRY(c, hipMemcpy(&h, c->d_counters.p, sizeof h, hipMemcpyDeviceToHost));
    out8[0] = (int64_t)h.v_trace; out8[1] = (int64_t)h.v_shadow; out8[2] = (int64_t)h.tri; out8[3] = (int64_t)h.shaded;
    out8[4] = (int64_t)h.pcand; out8[5] = (int64_t)h.traces; out8[6] = (int64_t)h.shadows; out8[7] = (int64_t)h.gathers;
    return GI_OK;
}

int gi_render_host(gi_ctx* c, const gi_render_params* p, void* h_out, int out_is_f64, int32_t* h_spp, volatile const int* cancel)
{
    if (!c || !h_out || !p) return GI_E_INVALID;
    const size_t npix = (size_t)gi_local_rows(p) * (size_t)std::max(p->width, 0);
    if (npix == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = npix * 3 * (out_is_f64 ? 8 : 4);
    void* d_out = nullptr;
    int32_t* d_spp = nullptr;
    HIP_TRY(c, hipMalloc(&d_out, bytes));
    if (h_spp && hipMalloc((void**)&d_spp, npix * 4) != hipSuccess) { (void)hipFree(d_out); return fail(c, GI_E_HIP, "hipMalloc spp"); }
    int rc = gi_render_device(c, p, d_out, out_is_f64, d_spp, cancel);
    if (rc == GI_OK) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(h_out, d_out, bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess && h_spp) e = hipMemcpy(h_spp, d_spp, npix * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, GI_E_HIP, std::string("render_host: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_out);
    if (d_spp) (void)hipFree(d_spp);
    return rc;
}

// ---- function-level entries: host in, host out
#define GI_GRID(n) dim3((unsigned)(((n) + GI_BLOCK - 1) / GI_BLOCK)), dim3(GI_BLOCK)

int gi_trace(gi_ctx* c, int32_t n, const double* rays, int32_t* hit, int32_t* ent, double* res)
{
    if (!c || n < 0 || (n && (!rays || !hit || !ent || !res))) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "trace: no scene uploaded");
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<double> d_r, d_o;
    DevBuf<int32_t> d_h, d_e;
    HIP_TRY(c, d_r.upload(std::vector<double>(rays, rays + (size_t)n * 6)));
    HIP_TRY(c, d_o.alloc((size_t)n * 8)); HIP_TRY(c, d_h.alloc(n)); HIP_TRY(c, d_e.alloc(n));
    hipLaunchKernelGGL(k_trace, GI_GRID(n), 0, c->stream, c->S, n, d_r.p, d_h.p, d_e.p, d_o.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(hit, d_h.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(ent, d_e.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(res, d_o.p, (size_t)n * 64, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_visible(gi_ctx* c, int32_t n, const double* q, int32_t* vis)
{
    if (!c || n < 0 || (n && (!q || !vis))) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "visible: no scene uploaded");
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<double> d_q;
    DevBuf<int32_t> d_v;
    HIP_TRY(c, d_q.upload(std::vector<double>(q, q + (size_t)n * 6)));
    HIP_TRY(c, d_v.alloc(n));
    hipLaunchKernelGGL(k_visible, GI_GRID(n), 0, c->stream, c->S, n, d_q.p, d_v.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(vis, d_v.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_visible_rays(gi_ctx* c, int32_t n, const double* rays, const double* mt, int32_t* vis)
{
    if (!c || n < 0 || (n && (!rays || !mt || !vis))) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "visible: no scene uploaded");
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<double> d_r, d_m;
    DevBuf<int32_t> d_v;
    HIP_TRY(c, d_r.upload(std::vector<double>(rays, rays + (size_t)n * 6)));
    HIP_TRY(c, d_m.upload(std::vector<double>(mt, mt + n)));
    HIP_TRY(c, d_v.alloc(n));
    hipLaunchKernelGGL(k_visible_rays, GI_GRID(n), 0, c->stream, c->S, n, d_r.p, d_m.p, d_v.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(vis, d_v.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_gather(gi_ctx* c, int32_t n, const double* q, double* res3, int32_t* n_cand)
{
    if (!c || n < 0 || (n && (!q || !res3))) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "gather: no scene uploaded");
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<double> d_q, d_r;
    DevBuf<int32_t> d_n;
    HIP_TRY(c, d_q.upload(std::vector<double>(q, q + (size_t)n * 6)));
    HIP_TRY(c, d_r.alloc((size_t)n * 3)); HIP_TRY(c, d_n.alloc(n));
    hipLaunchKernelGGL(k_gather, GI_GRID(n), 0, c->stream, c->S, n, d_q.p, d_r.p, d_n.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(res3, d_r.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (n_cand) HIP_TRY(c, hipMemcpy(n_cand, d_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_radiance(gi_ctx* c, int32_t n, const double* rays, const uint32_t* stream, uint64_t seed, double* out3)
{
    if (!c || n < 0 || (n && (!rays || !stream || !out3))) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "radiance: no scene uploaded");
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<double> d_r, d_o;
    DevBuf<uint32_t> d_s;
    HIP_TRY(c, d_r.upload(std::vector<double>(rays, rays + (size_t)n * 6)));
    HIP_TRY(c, d_s.upload(std::vector<uint32_t>(stream, stream + n)));
    HIP_TRY(c, d_o.alloc((size_t)n * 3));
    hipLaunchKernelGGL(k_radiance, GI_GRID(n), 0, c->stream, c->S, n, d_r.p, d_s.p, seed, d_o.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out3, d_o.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_emit_photons(gi_ctx* c, int32_t count, int32_t max_depth, uint64_t seed, double* photons_out, int32_t cap, int64_t* tries_out)
{
    if (!c || count < 0 || cap < 0 || (cap && !photons_out)) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "emit_photons: no scene uploaded");
    const long long total = (long long)count * c->S.n_light;
    if (tries_out) *tries_out = 0;
    if (total == 0) return 0;
    if (total > 0x7fffffffLL) return fail(c, GI_E_INVALID, "emit_photons: count too large");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<PhotonOut> d_p;
    DevBuf<int32_t> d_s, d_t;
    HIP_TRY(c, d_p.alloc((size_t)total)); HIP_TRY(c, d_s.alloc((size_t)total)); HIP_TRY(c, d_t.alloc((size_t)total));
    hipLaunchKernelGGL(k_emit, GI_GRID(total), 0, c->stream, c->S, count, max_depth, seed, d_p.p, d_s.p, d_t.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<PhotonOut> hp((size_t)total);
    std::vector<int32_t> hs((size_t)total), ht((size_t)total);
    HIP_TRY(c, hipMemcpy(hp.data(), d_p.p, (size_t)total * sizeof(PhotonOut), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(hs.data(), d_s.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(ht.data(), d_t.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    // compaction in (photon index, light) order = the order one reference thread appends them (include/raytracer.h:593-706)
    int stored = 0;
    int64_t tries = 0;
    for (long long j = 0; j < total; j++) {
        tries += ht[(size_t)j];
        if (!hs[(size_t)j]) continue;
        if (stored < cap) memcpy(photons_out + (size_t)stored * 9, hp[(size_t)j].v, 72);
        stored++;
    }
    if (tries_out) *tries_out = tries;
    if (stored > cap) return fail(c, GI_E_INVALID, "emit_photons: output capacity too small");
    return stored;
}

int gi_build_photon_map(gi_ctx* c, int32_t n, const double* photons, const double* box6)
{
    if (!c || n < 0 || (n && !photons)) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "build_photon_map: no scene uploaded");
    HIP_TRY(c, hipSetDevice(c->device));
    double box[6];
    for (int k = 0; k < 3; k++) { box[k] = box6 ? box6[k] : c->S.root_bmin[k]; box[3 + k] = box6 ? box6[3 + k] : c->S.root_bmax[k]; }
    DevBuf<double> d_ph;
    if (n) HIP_TRY(c, d_ph.upload(std::vector<double>(photons, photons + (size_t)n * 9)));
    return build_photon_map_on_device(c, d_ph.p, n, box);
}

int gi_trace_photons(gi_ctx* c, int32_t count, int32_t max_depth, uint64_t seed, const double* box6, int64_t* tries_out)
{
    if (!c || count < 0) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "trace_photons: no scene uploaded");
    const long long total = (long long)count * c->S.n_light;
    if (tries_out) *tries_out = 0;
    if (total > 0x7fffffffLL) return fail(c, GI_E_INVALID, "trace_photons: count too large");
    HIP_TRY(c, hipSetDevice(c->device));
    double box[6];
    for (int k = 0; k < 3; k++) { box[k] = box6 ? box6[k] : c->S.root_bmin[k]; box[3 + k] = box6 ? box6[3 + k] : c->S.root_bmax[k]; }
    if (total == 0) { const int rc = build_photon_map_on_device(c, nullptr, 0, box); return rc < 0 ? rc : 0; }
    DevBuf<PhotonOut> d_p;
    DevBuf<int32_t> d_s, d_t, d_x;
    HIP_TRY(c, d_p.alloc((size_t)total)); HIP_TRY(c, d_s.alloc((size_t)total)); HIP_TRY(c, d_t.alloc((size_t)total)); HIP_TRY(c, d_x.alloc((size_t)total));
    hipLaunchKernelGGL(k_emit, GI_GRID(total), 0, c->stream, c->S, count, max_depth, seed, d_p.p, d_s.p, d_t.p);
    // stored photons in (photon index, light) order = the order one reference thread appends them (include/raytracer.h:593-706)
    HIP_TRY(c, hipMemcpyAsync(d_x.p, d_s.p, (size_t)total * 4, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, c->stream, reinterpret_cast<uint32_t*>(d_x.p), (uint32_t)total);   // exclusive prefix sums, in place (gi_sort.inc)
    int32_t last_x = 0, last_s = 0;
    HIP_TRY(c, hipMemcpyAsync(&last_x, d_x.p + (total - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&last_s, d_s.p + (total - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int32_t n = last_x + last_s;
    if (tries_out) {      // total emission tries (diagnostic): summed on the host from the per-index counts
        std::vector<int32_t> ht((size_t)total);
        HIP_TRY(c, hipMemcpy(ht.data(), d_t.p, (size_t)total * 4, hipMemcpyDeviceToHost));
        int64_t tries = 0;
        for (int32_t v : ht) tries += v;
        *tries_out = tries;
    }
    DevBuf<double> d_ph;
    HIP_TRY(c, d_ph.alloc((size_t)std::max(n, 1) * 9));
    hipLaunchKernelGGL(k_pb_compact_emitted, GI_GRID(total), 0, c->stream, d_p.p, d_s.p, d_x.p, (uint32_t)total, d_ph.p);
    HIP_TRY(c, hipGetLastError());
    const int rc = build_photon_map_on_device(c, d_ph.p, n, box);
    return rc < 0 ? rc : n;
}

int gi_debug_photon_tables(gi_ctx* c, int32_t* n_node, int32_t* n_range, int32_t* n_photon, void* nodes128, int32_t* ranges2, double* pos3, double* dircol6)
{
    if (!c) return GI_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const Scene& S = c->S;
    if (n_node) *n_node = S.n_pnode;
    if (n_range) *n_range = S.n_pnode > 0 ? c->n_prange : 0;
    if (n_photon) *n_photon = S.n_photon;
    if (S.n_pnode <= 0) return GI_OK;
    if (nodes128) HIP_TRY(c, hipMemcpy(nodes128, S.pnodes, (size_t)S.n_pnode * sizeof(PNode), hipMemcpyDeviceToHost));
    if (ranges2) HIP_TRY(c, hipMemcpy(ranges2, S.pranges, (size_t)c->n_prange * sizeof(PRange), hipMemcpyDeviceToHost));
    if (pos3 && S.n_photon) HIP_TRY(c, hipMemcpy(pos3, S.ph_pos, (size_t)S.n_photon * 24, hipMemcpyDeviceToHost));
    if (dircol6 && S.n_photon) HIP_TRY(c, hipMemcpy(dircol6, S.ph_dircol, (size_t)S.n_photon * 48, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_debug_sort_pairs(gi_ctx* c, int32_t n, const uint32_t* keys, const uint32_t* vals, int32_t begin_bit, int32_t end_bit, uint32_t* keys_out, uint32_t* vals_out)
{
    if (!c || n < 0 || begin_bit < 0 || end_bit > 32 || end_bit <= begin_bit || (n && (!keys || !vals || !keys_out || !vals_out))) return GI_E_INVALID;
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<uint32_t> d[6], d_hist;
    for (int k = 0; k < 6; k++) HIP_TRY(c, d[k].alloc((size_t)n));
    HIP_TRY(c, d_hist.alloc((size_t)GI_RS_MAXBINS * GI_MAX_PRODUCER_BLOCKS));
    HIP_TRY(c, hipMemcpy(d[0].p, keys, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d[1].p, vals, (size_t)n * 4, hipMemcpyHostToDevice));
    int rc = rs_sort_pairs(c, d[0].p, d[2].p, d[1].p, d[3].p, d[4].p, d[5].p, (uint32_t)n, nullptr, begin_bit, end_bit, d_hist.p);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(keys_out, d[2].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(vals_out, d[3].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_debug_find_leaves(gi_ctx* c, int32_t n, const double* pos, int32_t* fast_out, int32_t* full_out)
{
    if (!c || n < 0 || (n && (!pos || !fast_out || !full_out))) return GI_E_INVALID;
    if (c->S.n_pnode <= 0) return fail(c, GI_E_STATE, "find_leaves: no photon map");
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<double> d_p;
    DevBuf<int32_t> d_a, d_b;
    HIP_TRY(c, d_p.upload(std::vector<double>(pos, pos + (size_t)n * 3)));
    HIP_TRY(c, d_a.alloc(n)); HIP_TRY(c, d_b.alloc(n));
    hipLaunchKernelGGL(k_find_leaves, GI_GRID(n), 0, c->stream, c->S, n, d_p.p, d_a.p, d_b.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(fast_out, d_a.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(full_out, d_b.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_debug_leaf_order(gi_ctx* c, int32_t n, const double* rays, int32_t cap, int32_t* leaf_out, int32_t* n_out)
{
    if (!c || n < 0 || cap < 1 || (n && (!rays || !leaf_out || !n_out))) return GI_E_INVALID;
    if (!c->have_scene) return fail(c, GI_E_STATE, "leaf_order: no scene uploaded");
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<double> d_r;
    DevBuf<int32_t> d_l, d_n;
    HIP_TRY(c, d_r.upload(std::vector<double>(rays, rays + (size_t)n * 6)));
    HIP_TRY(c, d_l.alloc((size_t)n * cap)); HIP_TRY(c, d_n.alloc(n));
    HIP_TRY(c, hipMemsetAsync(d_l.p, 0xff, (size_t)n * cap * 4, c->stream));
    hipLaunchKernelGGL(k_leaf_order, GI_GRID(n), 0, c->stream, c->S, n, d_r.p, cap, d_l.p, d_n.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(leaf_out, d_l.p, (size_t)n * cap * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(n_out, d_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_kat(gi_ctx* c, int32_t what, int32_t n, const double* in, int32_t in_stride, double* out3)
{
    if (!c || n < 0 || in_stride < 1 || in_stride > 9 || (n && (!in || !out3))) return GI_E_INVALID;
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<double> d_i, d_o;
    HIP_TRY(c, d_i.upload(std::vector<double>(in, in + (size_t)n * in_stride)));
    HIP_TRY(c, d_o.alloc((size_t)n * 3));
    hipLaunchKernelGGL(k_kat, GI_GRID(n), 0, c->stream, what, n, d_i.p, in_stride, d_o.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out3, d_o.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_halton_sample(gi_ctx* c, int32_t n, const uint32_t* dim, const uint32_t* index, float* out)
{
    if (!c || n < 0 || (n && (!dim || !index || !out))) return GI_E_INVALID;
    if (n == 0) return GI_OK;
    for (int i = 0; i < n; i++) if (dim[i] > 255) return fail(c, GI_E_INVALID, "halton_sample: dimension > 255");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<uint32_t> d_d, d_i;
    DevBuf<float> d_o;
    HIP_TRY(c, d_d.upload(std::vector<uint32_t>(dim, dim + n)));
    HIP_TRY(c, d_i.upload(std::vector<uint32_t>(index, index + n)));
    HIP_TRY(c, d_o.alloc(n));
    hipLaunchKernelGGL(k_halton, GI_GRID(n), 0, c->stream, c->S, n, d_d.p, d_i.p, d_o.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, d_o.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GI_OK;
}

int gi_halton_index(gi_ctx* c, int32_t width, int32_t height, int32_t n, const uint32_t* sxy, uint32_t* out)
{
    if (!c || width <= 0 || height <= 0 || n < 0 || (n && (!sxy || !out))) return GI_E_INVALID;
    if (n == 0) return GI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<uint32_t> d_i, d_o;
    HIP_TRY(c, d_i.upload(std::vector<uint32_t>(sxy, sxy + (size_t)n * 3)));
    HIP_TRY(c, d_o.alloc(n));
    hipLaunchKernelGGL(k_halton_index, GI_GRID(n), 0, c->stream, make_halton_enum((unsigned)width, (unsigned)height), n, d_i.p, d_o.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, d_o.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GI_OK;
}

// ================================================================================================= several GPUs from one process
// The reference parallelises RayTracer::run over image rows with OpenMP (include/raytracer.h:93).  A group does the same over the GPUs of a
// node from ONE process (the Qt application): one context and one host thread per device, the frame's stripes dealt round-robin, the scene
// and photon tables replicated, the finished stripes gathered into one frame -- on device 0 through peer copies over xGMI
// (gi_group_render_device) or straight into the caller's host frame (gi_group_render_host).
struct gi_group {
    std::vector<gi_ctx*> ctx;
    std::vector<void*> d_part;          // per device: its stripes, compact [local_rows][w][3]
    std::vector<size_t> part_bytes;
    std::vector<int32_t*> d_spp;
    std::vector<size_t> spp_bytes;
    std::string err;
};
}  // extern "C" (C++ helpers below)
#include <thread>
namespace {
int group_fail(gi_group* g, int code, const std::string& m) { if (g) g->err = m; return code; }
}
extern "C" {

int gi_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int gi_group_create(gi_group** out, int32_t n_devices, const int32_t* device_ordinals)
{
    if (!out || n_devices < 0) return GI_E_INVALID;
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return GI_E_NO_DEVICE;
    if (n_devices == 0) n_devices = visible;
    gi_group* g = new gi_group();
    for (int i = 0; i < n_devices; i++) {
        gi_ctx* c = nullptr;
        const int rc = gi_create(&c, device_ordinals ? device_ordinals[i] : i);
        if (rc != GI_OK) { for (gi_ctx* k : g->ctx) gi_destroy(k); delete g; return rc; }
        g->ctx.push_back(c);
    }
    g->d_part.assign((size_t)n_devices, nullptr); g->part_bytes.assign((size_t)n_devices, 0);
    g->d_spp.assign((size_t)n_devices, nullptr); g->spp_bytes.assign((size_t)n_devices, 0);
    // peer access towards device 0 for the xGMI gather (ignored where it is the same device or already enabled)
    for (int i = 1; i < n_devices; i++) {
        if (g->ctx[(size_t)i]->device == g->ctx[0]->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g->ctx[(size_t)i]->device, g->ctx[0]->device) == hipSuccess && can) {
            (void)hipSetDevice(g->ctx[(size_t)i]->device);
            (void)hipDeviceEnablePeerAccess(g->ctx[0]->device, 0);
            (void)hipGetLastError();
        }
    }
    *out = g;
    return GI_OK;
}

void gi_group_destroy(gi_group* g)
{
    if (!g) return;
    for (size_t i = 0; i < g->ctx.size(); i++) {
        (void)hipSetDevice(g->ctx[i]->device);
        if (g->d_part[i]) (void)hipFree(g->d_part[i]);
        if (g->d_spp[i]) (void)hipFree(g->d_spp[i]);
        gi_destroy(g->ctx[i]);
    }
    delete g;
}

int gi_group_size(const gi_group* g) { return g ? (int)g->ctx.size() : 0; }
gi_ctx* gi_group_ctx(gi_group* g, int32_t i) { return (g && i >= 0 && i < (int)g->ctx.size()) ? g->ctx[(size_t)i] : nullptr; }
const char* gi_group_last_error(const gi_group* g) { return g ? g->err.c_str() : "null group"; }

int gi_group_upload_scene(gi_group* g, const gi_scene_desc* d)
{
    if (!g) return GI_E_INVALID;
    for (gi_ctx* c : g->ctx) { const int rc = gi_upload_scene(c, d); if (rc) return group_fail(g, rc, gi_last_error(c)); }
    return GI_OK;
}
int gi_group_clear_photons(gi_group* g)
{
    if (!g) return GI_E_INVALID;
    for (gi_ctx* c : g->ctx) gi_clear_photons(c);
    return GI_OK;
}
int gi_group_upload_photons(gi_group* g, const gi_photon_map_desc* d)
{
    if (!g) return GI_E_INVALID;
    for (gi_ctx* c : g->ctx) { const int rc = gi_upload_photons(c, d); if (rc) return group_fail(g, rc, gi_last_error(c)); }
    return GI_OK;
}

// Stripes [first_stripe, first_stripe + n_stripes) of the frame (cut into stripes of stripe_h rows), stripe s on device (s - first_stripe) % n.
// Every device renders on its own host thread; `sink(i, ctx, rp, d_part, d_spp)` then moves device i's stripes where they belong.
static int group_render(gi_group* g, const gi_render_params* p, int32_t stripe_h, int32_t first_stripe, int32_t n_stripes, int out_is_f64, bool want_spp, volatile const int* cancel,
                        const std::function<int(int, gi_ctx*, const gi_render_params&, int32_t /*stripe of the call or -1 = all of this device's*/, const void*, const int32_t*)>& sink)
{
    if (!g || !p || stripe_h <= 0 || p->width <= 0 || p->height <= 0) return GI_E_INVALID;
    const int n = (int)g->ctx.size();
    const int total = (p->height + stripe_h - 1) / stripe_h;
    if (first_stripe < 0 || n_stripes < 0 || first_stripe + n_stripes > total) return group_fail(g, GI_E_INVALID, "group render: stripe window outside the frame");
    const bool whole = first_stripe == 0 && n_stripes == total;
    const size_t px = (size_t)(out_is_f64 ? 8 : 4) * 3;
    std::vector<int> rcs((size_t)n, GI_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < n; i++)
        th.emplace_back([&, i]() {
            gi_ctx* c = g->ctx[(size_t)i];
            if (hipSetDevice(c->device) != hipSuccess) { rcs[(size_t)i] = GI_E_HIP; return; }
            gi_render_params rp = *p;
            rp.stripe_h = stripe_h;
            // whole frame: one call renders all stripes of this device (rank i of n); a window: one call per stripe (rank = the stripe, world = all)
            std::vector<int32_t> calls;
            if (whole) { if (i < total) calls.push_back(-1); }
            else for (int32_t s = first_stripe + i; s < first_stripe + n_stripes; s += n) calls.push_back(s);
            for (int32_t s : calls) {
                rp.stripe_rank = s < 0 ? i : s;
                rp.stripe_world = s < 0 ? n : total;
                const size_t rows = (size_t)gi_local_rows(&rp);
                if (rows == 0) continue;
                const size_t need = rows * (size_t)p->width * px, need_spp = want_spp ? rows * (size_t)p->width * 4 : 0;
                if (g->part_bytes[(size_t)i] < need) {
                    if (g->d_part[(size_t)i]) (void)hipFree(g->d_part[(size_t)i]);
                    g->d_part[(size_t)i] = nullptr; g->part_bytes[(size_t)i] = 0;
                    if (hipMalloc(&g->d_part[(size_t)i], need) != hipSuccess) { rcs[(size_t)i] = GI_E_HIP; c->err = "group render: hipMalloc of the stripe buffer"; return; }
                    g->part_bytes[(size_t)i] = need;
                }
                if (g->spp_bytes[(size_t)i] < need_spp) {
                    if (g->d_spp[(size_t)i]) (void)hipFree(g->d_spp[(size_t)i]);
                    g->d_spp[(size_t)i] = nullptr; g->spp_bytes[(size_t)i] = 0;
                    if (hipMalloc((void**)&g->d_spp[(size_t)i], need_spp) != hipSuccess) { rcs[(size_t)i] = GI_E_HIP; c->err = "group render: hipMalloc of the sample-count buffer"; return; }
                    g->spp_bytes[(size_t)i] = need_spp;
                }
                int rc = gi_render_device(c, &rp, g->d_part[(size_t)i], out_is_f64, want_spp ? g->d_spp[(size_t)i] : nullptr, cancel);
                if (rc == GI_OK) rc = sink(i, c, rp, s, g->d_part[(size_t)i], want_spp ? g->d_spp[(size_t)i] : nullptr);
                if (rc == GI_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = GI_E_HIP;
                if (rc != GI_OK) { rcs[(size_t)i] = rc; return; }
            }
        });
    for (std::thread& t : th) t.join();
    for (int i = 0; i < n; i++)
        if (rcs[(size_t)i] != GI_OK) return group_fail(g, rcs[(size_t)i], std::string("device ") + std::to_string(g->ctx[(size_t)i]->device) + ": " + g->ctx[(size_t)i]->err);
    return GI_OK;
}

// rows of the frame held by a call's compact buffer, as (frame row of the block, rows, local row of the block)
static void stripe_blocks(const gi_render_params& rp, std::vector<std::array<int, 3>>& out)
{
    const int total = (rp.height + rp.stripe_h - 1) / rp.stripe_h;
    int local = 0;
    for (int k = rp.stripe_rank; k < total; k += rp.stripe_world) {
        const int rows = std::min(rp.stripe_h, rp.height - k * rp.stripe_h);
        out.push_back({k * rp.stripe_h, rows, local});
        local += rows;
    }
}

int gi_group_render_host(gi_group* g, const gi_render_params* p, int32_t stripe_h, int32_t first_stripe, int32_t n_stripes, void* h_frame, int out_is_f64, int32_t* h_spp, volatile const int* cancel)
{
    if (!h_frame) return GI_E_INVALID;
    const size_t px = (size_t)(out_is_f64 ? 8 : 4) * 3;
    return group_render(g, p, stripe_h, first_stripe, n_stripes, out_is_f64, h_spp != nullptr, cancel,
                        [&](int, gi_ctx* c, const gi_render_params& rp, int32_t, const void* d_part, const int32_t* d_spp) -> int {
                            std::vector<std::array<int, 3>> blocks;
                            stripe_blocks(rp, blocks);
                            if (hipStreamSynchronize(c->stream) != hipSuccess) return GI_E_HIP;
                            for (const auto& b : blocks) {     // every device writes its own rows of the caller's frame
                                const size_t w = (size_t)rp.width;
                                if (hipMemcpy((char*)h_frame + (size_t)b[0] * w * px, (const char*)d_part + (size_t)b[2] * w * px, (size_t)b[1] * w * px, hipMemcpyDeviceToHost) != hipSuccess) return GI_E_HIP;
                                if (h_spp && hipMemcpy(h_spp + (size_t)b[0] * w, d_spp + (size_t)b[2] * w, (size_t)b[1] * w * 4, hipMemcpyDeviceToHost) != hipSuccess) return GI_E_HIP;
                            }
                            return GI_OK;
                        });
}

int gi_group_render_device(gi_group* g, const gi_render_params* p, int32_t stripe_h, void* d_frame_on_device0, int out_is_f64, volatile const int* cancel)
{
    if (!g || !p || !d_frame_on_device0) return GI_E_INVALID;
    const size_t px = (size_t)(out_is_f64 ? 8 : 4) * 3;
    const int dev0 = g->ctx[0]->device;
    const int total = stripe_h > 0 ? (p->height + stripe_h - 1) / stripe_h : 0;
    return group_render(g, p, stripe_h, 0, total, out_is_f64, false, cancel,
                        [&](int, gi_ctx* c, const gi_render_params& rp, int32_t, const void* d_part, const int32_t*) -> int {
                            std::vector<std::array<int, 3>> blocks;
                            stripe_blocks(rp, blocks);
                            for (const auto& b : blocks) {     // the gather: this device's stripes into the frame on device 0, over xGMI when the devices differ
                                const size_t w = (size_t)rp.width;
                                void* dst = (char*)d_frame_on_device0 + (size_t)b[0] * w * px;
                                const void* src = (const char*)d_part + (size_t)b[2] * w * px;
                                const hipError_t e = c->device == dev0 ? hipMemcpyAsync(dst, src, (size_t)b[1] * w * px, hipMemcpyDeviceToDevice, c->stream)
                                                                       : hipMemcpyPeerAsync(dst, dev0, src, c->device, (size_t)b[1] * w * px, c->stream);
                                if (e != hipSuccess) return GI_E_HIP;
                            }
                            return GI_OK;
                        });
}

}  // extern "C"
