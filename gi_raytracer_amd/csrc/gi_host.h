/* gi_host.h -- host-side feeders of the hot path (C API): scene ingestion, Octree::rebuild, PhotonMap::rebuild and the
 * flattening into the gi_scene_desc / gi_photon_map_desc tables that include/gi_hip.h consumes.
 *
 * These are the "callers and data formats either side of the path" (SURVEY.md 8 f2/f3).  They run on the host exactly
 * as in the reference (its loaders and tree builders are host code too); all per-ray / per-pixel work is in the HIP
 * kernels.  The C++ classes of include/gi/ (Octree, PhotonMap, RayTracer ...) are built on this API; Python reaches it
 * through ctypes (gi_raytracer_amd/__init__.py).
 */
#ifndef GI_HOST_H
#define GI_HOST_H
#include <stdint.h>
#include "../../include/gi_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gih_scene gih_scene;

gih_scene* gih_scene_create(void);
void gih_scene_destroy(gih_scene*);
const char* gih_last_error(const gih_scene*);

/* replaces loadScene + loadOBJ (include/sceneLoader.cpp:12-185, include/meshLoader.cpp:18-99): same keyword set and the
 * same word-wise tokenising (keywords colorTex, checkerboardTex, imTex, mat, multiMat, mesh, sphere, box, light, heightFog, photons, samples, ambient, camera);
 * vertices, normals and uvs are rounded to float as the reference's loader does.  A `mat` line
 * without its 5th number gets IOR 1.0 (the reference leaves it uninitialised).  Missing mesh files are skipped.
 * Returns 0, or -1 when the .scn itself cannot be opened.                                                            */
int gih_load_scn(gih_scene*, const char* path);

/* replaces loadOBJ alone (include/meshLoader.cpp:18-99): the faces of one .obj, rotated (glm::eulerAngleXYZ) and moved, with material mat_idx;
 * returns 0, or non-zero when the file cannot be opened / a face is not a v/vt/vn triple                                                */
int gih_load_obj(gih_scene*, const char* path, const double* pos3, const double* rot3, int32_t mat_idx);

/* programmatic construction = Octree::push_back (include/octree.cpp:25-50) */
int gih_add_material(gih_scene*, const double* mat9);                 /* returns the material index */
/* textures (include/material.h:10-81): kind and params8 as gi_scene_desc::tex_kind / tex_param; an image (kind 2) passes its
 * width x height RGBA8 pixels (rows top to bottom) and params8 = tile u, tile v, width, height, has alpha; returns the texture
 * index.  gih_add_material_tex = new Material(tex[dif], tex[em], roughness, opacity, IOR), returns the material index.
 * gih_load_png decodes an 8-bit, non-interlaced PNG the way QImage presents it (the loader of `imTex` lines).                */
int gih_add_texture(gih_scene*, int32_t kind, const double* params8, const uint8_t* rgba, int64_t n_bytes);
int gih_add_material_tex(gih_scene*, int32_t dif_tex, int32_t em_tex, double roughness, double opacity, double ior);
int gih_load_png(const char* path, int32_t* width, int32_t* height, int32_t* has_alpha, uint8_t** rgba_out /* free with gih_free */, char* err, int32_t err_len);
void gih_free(void*);
int gih_add_triangles(gih_scene*, int32_t n, const double* pos, const double* nrm, const double* uv, const int32_t* mat_idx);
int gih_add_sphere(gih_scene*, const double* centre3, double radius, int32_t mat_idx);   /* new sphere(pos, rad, mat) */
/* new HeightFog(pos, size, col, density, scatter, noiseScale) (include/atmosphere.h:37-47): params12 as in gi_scene_desc::fog;
 * grid = its (sx+1)(sy+1)(sz+1) scale^3 noise values, or NULL to fill them from the counter RNG with `seed`                  */
int gih_add_height_fog(gih_scene*, const double* params12, const double* grid, int32_t n_grid, uint64_t seed);
int gih_add_height_fog_grid(gih_scene*, const double* params12, const double* grid, int32_t n_grid);   /* a HeightFog that carries its own grid, any size */
int gih_add_light(gih_scene*, const double* pos3, const double* col3, double rad);
int gih_set_ambient(gih_scene*, const double* rgb3);

/* render settings a .scn may carry (include/raytracer.h:721-726, include/sceneLoader.cpp:160-179) */
typedef struct gih_settings {
    int32_t photons, photon_depth, min_samples, max_samples;
    double noise_thresh;
    double ambient[3];
    double cam_pos[3], cam_up[3], cam_forward[3], sensor_diag, focal_dist;
} gih_settings;
int gih_get_settings(const gih_scene*, gih_settings* out);
/* Camera(pos, lookAt) / Camera::setDir (include/camera.h:9-23) */
int gih_set_camera(gih_scene*, const double* pos3, const double* look_at3);

/* replaces Octree::rebuild (include/octree.cpp:53-119): light cone precompute + Octree::Node::partition (:316-384) */
int gih_build_octree(gih_scene*);
/* pointers stay valid until the scene is modified or destroyed */
int gih_get_scene_desc(const gih_scene*, gi_scene_desc* out);
int gih_counts(const gih_scene*, int32_t* n_tri, int32_t* n_mat, int32_t* n_light, int32_t* n_node, int32_t* n_ref);

/* replaces PhotonMap::push_back + rebuild (include/photonMap.cpp:24-47,137-192) over the scene's root box
 * (RayTracer::setScene, include/raytracer.h:38); photons [n][9] are copied                                           */
int gih_build_photon_map(gih_scene*, int32_t n, const double* photons);
/* the same in a box given by the caller -- PhotonMap(min, max) of the C++ surface (include/photonMap.h:33); the handle then serves as the map's container only */
int gih_build_photon_map_in_box(gih_scene*, const double* box6, int32_t n, const double* photons);
int gih_get_photon_desc(const gih_scene*, gi_photon_map_desc* out);

/* Single-object forms of the builder's entity tests and of two generators, for the C++ entity classes (include/gi/entities.h, atmosphere.h):
 * Entity::boundingBox / Entity::intersect(BoundingBox) of a triangle (kind 0: pos9 = three vertices) or sphere (kind 1: centre, radius in
 * pos9[3]) (include/entities.h:103-141,522-557); boxMesh's 12 triangles [12][3][3] (include/entities.h:740-785); the noise grid of a
 * HeightFog from the counter RNG (grid == NULL: returns the number of values).                                                       */
int gih_entity_bbox(int32_t kind, const double* pos9, double* out6);
int gih_entity_overlaps_box(int32_t kind, const double* pos9, const double* box6);   /* 1 / 0, negative on bad arguments */
int gih_box_mesh(const double* pos3, const double* size3, const double* rot3, double* out108);
int gih_fog_grid(const double* params12, uint64_t seed, double* grid, int32_t n_grid);

/* replaces the 8-bit sink of RayTracer::run (include/raytracer.h:150-157: gamma(color, 2.2), glm::clamp) + Image::setPixel
 * (include/image.h:14-16: truncating (int)(255 c)): n_values linear channel values (float or double) -> bytes.  A negative value
 * (the caustic term can be negative) gives 0, as in the reference.                                                      */
int gih_to_rgb8(const void* lin, int32_t is_f64, int64_t n_values, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
