/* include/gi_hip.h -- C ABI of the MI355X (gfx950) render hot path of GI_Raytracer.
 *
 * This is the drop-in boundary (DESIGN.md "Boundary").  The reference has no FFI of its own: its
 * de-facto interface is the public C++ surface of include/raytracer.h, include/octree.h, include/photonMap.h and
 * include/entities.h.  The C++ headers under include/gi/ keep those spellings and are thin callers of the entry points
 * below; a maintainer of the reference swaps the bodies of the cited functions for these calls.
 *
 * Conventions: every entry returns 0 on success and a negative GI_E_* code on failure; gi_last_error() gives the text;
 * nothing throws, nothing prints.  All array arguments are caller-owned, read-only, plain host pointers and are copied
 * to the device by the call that receives them.  One context is used from one thread at a time (the reference calls
 * RayTracer::run from one worker thread, include/viewer.h:52-56).  There is NO CPU fallback: without a usable HIP
 * device gi_create fails with GI_E_NO_DEVICE.
 */
#ifndef GI_HIP_H
#define GI_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GI_OK 0
#define GI_E_NO_DEVICE (-1)
#define GI_E_INVALID (-2)
#define GI_E_HIP (-3)
#define GI_E_STATE (-4)
#define GI_E_CANCELLED (-5)

typedef struct gi_ctx gi_ctx;

/* Flattened scene = what Octree holds after push_back()/rebuild() (include/octree.h:39-64, include/octree.cpp:25-119). */
typedef struct gi_scene_desc {
    int32_t n_tri;           /* number of entities (triangles, and spheres when ent_kind is given)                          */
    const double* tri_pos;   /* [n_tri][3][3] vertex positions          (triangle::vertices[k].pos, include/entities.h:331) */
    const double* tri_nrm;   /* [n_tri][3][3] vertex normals (all-zero row = flat shading, include/entities.h:478)          */
    const double* tri_uv;    /* [n_tri][3][2] texture coordinates                                                           */
    const int32_t* tri_mat;  /* [n_tri] index into mats                                                                     */
    int32_t n_mat;
    const double* mats;      /* [n_mat][9] roughness, opacity, IOR, diffuse rgb, emissive rgb (include/material.h:84-100)   */
    int32_t n_light;
    const double* lights;    /* [n_light][11] pos, col, rad, dir, angle (include/light.h:10-58; dir/angle from Octree::rebuild) */
    double ambient[3];       /* RayTracer::ambient (include/raytracer.h:726)                                                */
    /* linearised Octree: nodes in pre-order, children 0..7 as in Octree::Node::partition (include/octree.cpp:321-328)      */
    int32_t n_node;
    const double* node_bbox;     /* [n_node][6] min xyz, max xyz                                                            */
    const int32_t* node_child;   /* [n_node][8] node index or -1 (null child)                                               */
    const int32_t* node_ent_off; /* [n_node+1] range of node_ent_idx owned by the node (leaves only)                        */
    const int32_t* node_ent_idx; /* [node_ent_off[n_node]] triangle indices in Node::_entities order                        */
    /* entity kinds (include/entities.h): NULL = all triangles; else [n_tri] with 0 = triangle, 1 = analytic sphere
     * (include/entities.h:51-142) whose centre is tri_pos[i][0] and radius tri_pos[i][1].x (normals / uvs unused)          */
    const int32_t* ent_kind;
    /* atmosphere entities (Octree::at, include/octree.h:43): HeightFog (include/atmosphere.h:30-83).  fog [n_fog][12] = pos, size,
     * colour, density, scatter, noise scale; the noise grid of entity i is fog_grid[fog_grid_off[i] .. fog_grid_off[i+1])         */
    int32_t n_fog;
    const double* fog;
    const int32_t* fog_grid_off;
    const double* fog_grid;
    /* textures (include/material.h:10-81).  n_tex = 0: every material is the constant colour pair held in mats[] (colorTex).
     * Else mat_tex [n_mat][2] = diffuse and emissive texture of each material, and texture t is
     *   tex_kind[t] = 0  texture(col):            tex_param[t] = col rgb
     *   tex_kind[t] = 1  checkerboard(t, a, b):   tex_param[t] = a rgb, b rgb, tiles
     *   tex_kind[t] = 2  imageTexture(file, tile): tex_param[t] = tile u, tile v, width, height, has alpha channel (0/1), offset of
     *                    its first pixel in tex_pixels (pixels: RGBA8, rows top to bottom as QImage addresses them)           */
    int32_t n_tex;
    const int32_t* tex_kind;
    const double* tex_param;     /* [n_tex][8] */
    const int32_t* mat_tex;      /* [n_mat][2] */
    const uint8_t* tex_pixels;
    int64_t n_tex_pixel_bytes;
} gi_scene_desc;

/* Photon set + linearised PhotonMap (include/photonMap.h:13-49, include/photon.h:5-15). */
typedef struct gi_photon_map_desc {
    int32_t n_photon;
    const double* photons;       /* [n_photon][9] origin, dir, col                                                          */
    int32_t n_node;              /* 0 = no map (gather returns 0, as an empty PhotonMap does)                                */
    const double* node_bbox;     /* [n_node][6]                                                                             */
    const int32_t* node_child;   /* [n_node][8] node index, or -1 in every slot for a leaf (PhotonMap::Node::is_leaf)       */
    const int32_t* node_off;     /* [n_node+1] range of node_idx                                                            */
    const int32_t* node_idx;     /* photon indices in Node::_entities order                                                 */
} gi_photon_map_desc;

/* Camera (include/camera.h:7-31) + frame + sampling budget (include/raytracer.h:721-726). */
typedef struct gi_render_params {
    double cam_pos[3], cam_up[3], cam_forward[3];
    double sensor_diag, focal_dist;
    int32_t width, height;
    /* Row sharding for multi-GPU runs: the frame is cut into stripes of stripe_h rows; this context renders stripes
     * k with k % stripe_world == stripe_rank.  Single GPU: stripe_h = height, rank 0, world 1.                            */
    int32_t stripe_h, stripe_rank, stripe_world;
    int32_t min_samples, max_samples;
    double noise_thresh;
    uint64_t seed;           /* counter-RNG seed (replaces the reference's time-seeded drand(), DESIGN.md "RNG contract")  */
} gi_render_params;

int gi_create(gi_ctx** out, int device_ordinal);
void gi_destroy(gi_ctx*);
const char* gi_last_error(const gi_ctx*);
/* Launch kernels of this context on a caller-owned HIP stream (hipStream_t passed as void*; NULL = default stream). */
int gi_set_stream(gi_ctx*, void* hip_stream);

/* replaces: the scene half of RayTracer::setScene + the tree walk of Octree::intersect/intersectSorted (include/raytracer.h:35-39,
 * include/octree.cpp:150-211,256-313) -- uploads the tables every kernel reads.  It does NOT drop the photon map (ABI change of round 2,
 * INTEGRATION.md "ABI notes"): a caller that uploads a DIFFERENT scene calls gi_clear_photons as well -- RayTracer::setScene is the pair --
 * or the gather keeps answering from the previous scene's photons.                                                       */
int gi_upload_scene(gi_ctx*, const gi_scene_desc*);
/* replaces: PhotonMap::push_back/rebuild as the source of the gather (include/photonMap.cpp:24-47).  The map outlives gi_upload_scene
 * (the reference keeps a valid map when an edited scene is rebuilt, include/raytracer.h:56-72); gi_clear_photons drops it -- RayTracer::setScene,
 * which allocates a fresh PhotonMap (include/raytracer.h:38), is gi_upload_scene + gi_clear_photons.                                    */
int gi_upload_photons(gi_ctx*, const gi_photon_map_desc*);
int gi_clear_photons(gi_ctx*);

/* replaces: the pixel loop of RayTracer::run (include/raytracer.h:93-160) with radiance/trace/visible/secondaryRay/
 * samplePhotons inside (include/raytracer.h:167-579).
 * Number of rows this context renders for the given sharding: gi_local_rows().
 * d_out_lin: DEVICE pointer to [local_rows][width][3] linear (pre-gamma, unclamped) radiance, float (out_is_f64 = 0) or
 * double (1).  d_out_spp: optional DEVICE pointer to [local_rows][width] int32 samples taken (adaptive loop).
 * cancel: optional host flag polled between launches (RayTracer::stop, include/raytracer.h:98,718).                        */
int gi_local_rows(const gi_render_params*);
int gi_render_device(gi_ctx*, const gi_render_params*, void* d_out_lin, int out_is_f64, int32_t* d_out_spp, volatile const int* cancel);
/* Same, result copied to HOST memory (what a Qt-side caller wants). */
int gi_render_host(gi_ctx*, const gi_render_params*, void* h_out_lin, int out_is_f64, int32_t* h_out_spp, volatile const int* cancel);
/* Schedules of the same per-path arithmetic: 0 = wavefront pipeline (default: trace / shade / gather kernels over compacted
 * path queues in HBM; fixed-spp frames refill finished slots with new samples, adaptive frames run in synchronous rounds),
 * 1 = megakernel (one lane keeps one pixel, whole path in registers), 2 = wavefront in synchronous rounds always.            */
int gi_set_render_mode(gi_ctx*, int mode);
/* Octree walk: 1 (default) = wide records, the boxes of a node's children tested together from the five planes per axis that
 * Octree::Node::partition builds them from (include/octree.cpp:318-328); 0 = one box test per node record.  Same box
 * arithmetic (include/bbox.h:47-73), same visiting order, same results; a tree whose children are not those exact octants
 * always takes the per-node walk.  The photon octree has the same option (PhotonMap::Node::partition builds its children from
 * the same planes, include/photonMap.cpp:139-149): the descent of getBounds reads one record per level instead of two.
 * Exists so that the two can be compared.  Returns a bit set, negative on error: 1 = the wide walk is in use after the call
 * (a scene is uploaded and its tree qualifies), 2 = the uploaded photon octree is walked one record per level.               */
int gi_set_wide_nodes(gi_ctx*, int enable);
/* Content-box culling: 1 (default) = a child of the wide walk whose octant the ray enters is skipped when the ray misses the box of everything
 * referenced in that child's sub-tree (no entity there can report a hit, so the walk's results are the same by construction); 0 = every such
 * child is visited, as Octree::Node::intersectSorted does (include/octree.cpp:285-313).  Exists so that the two can be compared (frames are
 * identical bit for bit).  Returns 1 when culling is in use after the call (a scene is uploaded, its tree takes the wide walk), else 0.      */
int gi_set_content_culling(gi_ctx*, int enable);
/* Entity boxes: 1 (default) = the wide walk runs Entity::intersect only on the references of a leaf whose own (widened) box the ray touches; 0 = on
 * every reference, as RayTracer::trace / visible do (include/raytracer.h:290-305,446-472).  A ray that misses an entity's box cannot hit the entity:
 * same hits, same order, same frame bit for bit; exists so that the two can be compared.  Returns 1 when in use after the call.
 * The closest-hit walk's further cuts hang on the same switch (0 = off, the walks ask exactly what the reference asks): boxes cut to the part of an
 * opaque entity inside its leaf, content boxes made of those, no look behind the best hit -- and with them the re-walk of a ray that met two
 * entities at exactly the same distance, and the plain walk of rays along an axis plane (gi_device.h: trace_wide_step).       */
int gi_set_entity_boxes(gi_ctx*, int enable);
/* Upper bound on paths in flight in the wavefront pipeline (232 B each, as field arrays).  Default: as many as 90 % of the free HBM holds
 * next to their queues, up to the whole frame (1080p x 256 spp = 531 M paths = 123 GB).                                                           */
int gi_set_pool_slots(gi_ctx*, int64_t slots);
/* Device time in ms of the render kernel(s) of the last gi_render_* call, measured with hipEvents on the launch stream. */
int gi_last_render_ms(gi_ctx*, float* ms, int32_t* n_launches);
/* Device time per pipeline stage of the last render, summed over its launches (HIP events around every launch):
 * [0] regenerate, [1] trace, [2] shade, [3] sorts (+ key kernels), [4] gather, [5] finish, [6] accumulate, [7] other.      */
int gi_last_stage_ms(gi_ctx*, float* out8);
/* The same per kernel family, out10: [0..7] as above except [2] = k_st_shade alone, [8] = k_st_shadow (the shadow walks the shade stage put off),
 * [9] reserved (0).                                                                                                         */
int gi_last_kernel_ms(gi_ctx*, float* out10);
/* Work counters of the last render.  gi_set_counters(ctx, 1): the REFERENCE's visits -- the frame is rendered by the megakernel with the
 * per-node walk and nothing culled, and gi_get_counters gives node visits in trace (BoundingBox::intersect calls of Octree::Node::intersectSorted,
 * include/octree.cpp:285-313), node visits in visible, triangle tests, shaded hits, photon candidates, trace calls, shadow rays, gathers.
 * gi_set_counters(ctx, 2): what the streaming kernels EXECUTE -- the frame is rendered by the pipeline that is benchmarked, its walks count per
 * lane, and gi_get_stream_counters gives out17 = k_st_trace: [0] walks begun (root box tests), [1] wide records visited, [2] child boxes tested
 * from them, [3] content boxes tested, [4] non-empty leaves met, [5] entity tests, [6] rays handed to the kernel; k_st_shadow: [7..12] the same
 * six, [13] shadow segments; [14] gather queries, [15] photon candidates they scanned, [16] shaded hits; [17] / [18] references sorted by their
 * entity box in k_st_trace / k_st_shadow (out must hold 19 values).  With content-box culling and entity boxes off, [5] is the reference's count of
 * entity tests in trace() exactly, and [0] the number of its trace() calls.  Covers fixed-sample-count frames of triangle scenes
 * without spheres, fog or textures (the BASELINE scenes); other scenes get GI_E_STATE.  0 = off (default).                                  */
int gi_set_counters(gi_ctx*, int mode);
int gi_get_counters(gi_ctx*, int64_t* out8);
int gi_get_stream_counters(gi_ctx*, int64_t* out19);

/* Function-level entry points (parity tests and the C++ API's public methods).  Host pointers.
 * replaces RayTracer::trace (include/raytracer.h:382-478): rays [n][6] origin + unit dir -> hit, entity, res [n][8]        */
int gi_trace(gi_ctx*, int32_t n, const double* rays, int32_t* hit, int32_t* ent, double* res);
/* replaces RayTracer::visible (include/raytracer.h:280-319): q [n][6] = shadow-ray origin, target point                    */
int gi_visible(gi_ctx*, int32_t n, const double* q, int32_t* vis);
/* the same with the reference's own arguments: rays [n][6] = shadow-ray origin + unit direction, mt [n] = squared distance to the light point */
int gi_visible_rays(gi_ctx*, int32_t n, const double* rays, const double* mt, int32_t* vis);
/* replaces RayTracer::samplePhotons(pos, dir, 32) (include/raytracer.h:532-579): q [n][6] = pos, dir                       */
int gi_gather(gi_ctx*, int32_t n, const double* q, double* res3, int32_t* n_cand);
/* replaces RayTracer::radiance(ray, 0, ...) (include/raytracer.h:167-276): rays [n][6], stream [n] = Halton sample index  */
int gi_radiance(gi_ctx*, int32_t n, const double* rays, const uint32_t* stream, uint64_t seed, double* out3);
/* replaces RayTracer::tracePhotons (include/raytracer.h:582-715): emits `count` photon indices per light on the device;
 * photons_out [cap][9]; returns the number stored (<= count * n_light) or a negative error; tries_out = emission tries.    */
int gi_emit_photons(gi_ctx*, int32_t count, int32_t max_depth, uint64_t seed, double* photons_out, int32_t cap, int64_t* tries_out);

/* Halton_sampler::sample / Halton_enum::get_index on the device tables (include/halton_sampler.h:626-888,
 * include/halton_enum.h:106-114) -- known-answer access for tests                                                          */
int gi_halton_sample(gi_ctx*, int32_t n, const uint32_t* dim, const uint32_t* index, float* out);
int gi_halton_index(gi_ctx*, int32_t width, int32_t height, int32_t n, const uint32_t* sxy /*[n][3]*/, uint32_t* out);

/* The photon map built on the device.  replaces: PhotonMap::push_back x n + PhotonMap::rebuild / PhotonMap::Node::partition
 * (include/photonMap.cpp:24-47,137-192) and the upload, for photons handed in (gi_build_photon_map: photons [n][9] on the host) or emitted
 * by the device itself (gi_trace_photons = RayTracer::tracePhotons(max_depth, count) + rebuild, include/raytracer.h:61-72,582-715: the
 * photons never leave the device; returns the number stored, tries_out = emission tries).  box6 = the box of the map (PhotonMap(min, max));
 * NULL = the root box of the uploaded scene (RayTracer::setScene, include/raytracer.h:38).  The tables are, byte for byte, the ones
 * gi_upload_photons derives from the host builder's tree.  gi_debug_photon_tables copies the installed tables back (NULL pointers: sizes only):
 * nodes128 [n_node][128 bytes], ranges2 [n_range][2], pos3 [n_photon][3], dircol6 [n_photon][6].                                      */
int gi_build_photon_map(gi_ctx*, int32_t n, const double* photons, const double* box6);
int gi_trace_photons(gi_ctx*, int32_t count, int32_t max_depth, uint64_t seed, const double* box6, int64_t* tries_out);
int gi_debug_photon_tables(gi_ctx*, int32_t* n_node, int32_t* n_range, int32_t* n_photon, void* nodes128, int32_t* ranges2, double* pos3, double* dircol6);

/* Several GPUs from one process.  Replaces the OpenMP row loop of RayTracer::run (include/raytracer.h:93: `#pragma omp parallel for
 * schedule(dynamic, 10)` over image rows) for a caller that is one process -- the Qt application: a group holds one context per device, the
 * scene and photon tables are replicated, the frame's stripes of stripe_h rows are dealt round-robin to the devices, each rendered on its own
 * host thread, and gathered into one frame.
 * gi_group_create: n_devices = 0 takes every visible device; device_ordinals may repeat an ordinal (several contexts on one device: how the
 * one-GPU tests exercise this path).  gi_group_ctx gives a member context, e.g. for gi_emit_photons on device 0.
 * gi_group_render_host: stripes [first_stripe, first_stripe + n_stripes) of the frame, stripe s on device (s - first_stripe) % n; every device
 * copies its rows into the caller's whole-frame host buffer h_frame [height][width][3] (float or double) and h_spp [height][width] (optional);
 * other rows are not touched -- RayTracer::run's progressive display calls this with a window of n stripes per step.
 * gi_group_render_device: the whole frame, gathered into d_frame_on_device0 (DEVICE pointer on the first context's device) with peer copies
 * over xGMI.  Both return after every device has finished; errors name the device (gi_group_last_error).                              */
typedef struct gi_group gi_group;
int gi_device_count(void);   /* usable HIP devices (0 without a GPU or a driver) */
int gi_group_create(gi_group** out, int32_t n_devices, const int32_t* device_ordinals);
void gi_group_destroy(gi_group*);
int gi_group_size(const gi_group*);
gi_ctx* gi_group_ctx(gi_group*, int32_t i);
const char* gi_group_last_error(const gi_group*);
int gi_group_upload_scene(gi_group*, const gi_scene_desc*);
int gi_group_upload_photons(gi_group*, const gi_photon_map_desc*);
int gi_group_clear_photons(gi_group*);
int gi_group_render_host(gi_group*, const gi_render_params*, int32_t stripe_h, int32_t first_stripe, int32_t n_stripes, void* h_frame, int out_is_f64, int32_t* h_spp, volatile const int* cancel);
int gi_group_render_device(gi_group*, const gi_render_params*, int32_t stripe_h, void* d_frame_on_device0, int out_is_f64, volatile const int* cancel);

/* Diagnostics for parity tests (not needed to render).
 * gi_debug_leaf_order: what Octree::intersectSorted(ray, 0, inf) returns (include/octree.cpp:188-211,285-313) as the device walk produces
 * it: for ray i, the non-empty leaves in visiting order as pre-order node indices in leaf_out[i*cap ..], their number in n_out[i]
 * (may exceed cap; only cap are stored).  Uses the walk the kernels use (wide records or per-node links, gi_set_wide_nodes).
 * gi_kat: known answers of the scalar building blocks as the device computes them.  what: 0 fastPow(a,b), 1 fastPrecisePow(a,b)
 * (include/util.h:100-136), 2 hemisphereSample_cos(n,u,v,power), 3 sample_phong(outdir,power,sx,sy), 4 sphereCapSample_cos(n,u,v,power,frac)
 * (include/util.cpp:35-107), 5 randomUnitVec(x,y), 6 refr(inc,n,eta) (include/util.h:173-188), 7 glm::reflect(inc,n); 16..22 the libm
 * calls of the path: sin, cos, acos, asin, atan2(a,b), pow(a,b), sqrt.  8 = the counter RNG that stands in for drand() (include/util.h:52-80;
 * DESIGN.md "RNG contract"): in = seed high 32 bits, seed low 32 bits, stream, depth, purpose, a, b (integers carried in doubles),
 * out = the draw, the stream key's high and low 32 bits.  in [n][in_stride] (arguments in the order given), out3 [n][3].  */
int gi_debug_leaf_order(gi_ctx*, int32_t n, const double* rays, int32_t cap, int32_t* leaf_out, int32_t* n_out);
/* gi_debug_sort_pairs: the pipeline's own radix sort (gi_sort.inc: the gather queries by photon-map leaf, the continuing rays by coherence key) on
 * caller data -- n (key, value) pairs sorted by bits [begin_bit, end_bit) of the key, stable.  Host pointers.                                */
int gi_debug_sort_pairs(gi_ctx*, int32_t n, const uint32_t* keys, const uint32_t* vals, int32_t begin_bit, int32_t end_bit, uint32_t* keys_out, uint32_t* vals_out);
/* gi_debug_find_leaves: the photon-map leaf around each position (PhotonMap::Node::getBounds, include/photonMap.cpp:115-134) as the two descents of the
 * pipeline find it: full_out the one that asks every box on the way (-1: no leaf contains the position), fast_out the one the gather keys of a pass
 * take (split records only, the first five levels through a jump table), -2 where that one declines (a position within 1e-12 of a split plane). */
int gi_debug_find_leaves(gi_ctx*, int32_t n, const double* pos, int32_t* fast_out, int32_t* full_out);
int gi_kat(gi_ctx*, int32_t what, int32_t n, const double* in, int32_t in_stride, double* out3);


#ifdef __cplusplus
}
#endif
#endif
