// include/gi/builtin_loaders.h -- loadScene / loadOBJ of this package, for programs that do not bring the reference's sceneLoader.cpp /
// meshLoader.cpp: the files are parsed by the C loader of the host library (gih_load_scn / gih_load_obj: the reference's keyword set,
// word-wise tokenising and float rounding, include/sceneLoader.cpp:12-185, include/meshLoader.cpp:18-99) and handed to the Octree as the
// entity objects the reference's loaders would have pushed -- triangle, sphere, Light, HeightFog with their Material / texture objects.
// Include this header in exactly one translation unit.
#pragma once
#include <cstdio>
#include <vector>
#include "meshLoader.h"
#include "sceneLoader.h"
namespace gi {
inline void push_loaded_entities(Octree* o, gih_scene* s, const std::vector<const Material*>* given)
{
    if (gih_build_octree(s) != 0) { fprintf(stderr, "loader: %s\n", gih_last_error(s)); return; }   // the tables are handed out with the tree
    gi_scene_desc d;
    gih_get_scene_desc(s, &d);
    std::vector<texture*> tex;
    for (int t = 0; t < d.n_tex; t++) {
        const double* q = d.tex_param + (size_t)t * 8;
        if (d.tex_kind[t] == 0) tex.push_back(new texture(dvec3(q[0], q[1], q[2])));
        else if (d.tex_kind[t] == 1) tex.push_back(new checkerboard((int)q[6], dvec3(q[0], q[1], q[2]), dvec3(q[3], q[4], q[5])));
        else tex.push_back(new imageTexture((int)q[2], (int)q[3], q[4] != 0, d.tex_pixels + (size_t)q[5], dvec2(q[0], q[1])));
    }
    std::vector<const Material*> mats;
    for (int m = 0; m < d.n_mat; m++) {
        if (given) { mats.push_back((*given)[std::min((size_t)m, given->size() - 1)]); continue; }
        const double* q = d.mats + (size_t)m * 9;
        texture* dt = d.n_tex > 0 && d.mat_tex[m * 2] >= 0 ? tex[(size_t)d.mat_tex[m * 2]] : new texture(dvec3(q[3], q[4], q[5]));
        texture* et = d.n_tex > 0 && d.mat_tex[m * 2 + 1] >= 0 ? tex[(size_t)d.mat_tex[m * 2 + 1]] : new texture(dvec3(q[6], q[7], q[8]));
        mats.push_back(new Material(dt, et, q[0], q[1], q[2]));
    }
    for (int i = 0; i < d.n_tri; i++) {
        const double* P = d.tri_pos + (size_t)i * 9;
        const Material& m = *mats[(size_t)d.tri_mat[i]];
        if (d.ent_kind && d.ent_kind[i] == 1) { o->push_back(new sphere(get3(P), P[3], m)); continue; }
        vertex v[3];   // fields set directly: the tables hold the normals as the vertex constructor left them
        for (int k = 0; k < 3; k++) {
            v[k].pos = get3(P + 3 * k);
            v[k].norm = get3(d.tri_nrm + (size_t)i * 9 + 3 * k);
            v[k].texCoord = dvec2(d.tri_uv[(size_t)i * 6 + 2 * k], d.tri_uv[(size_t)i * 6 + 2 * k + 1]);
        }
        o->push_back(new triangle(v[0], v[1], v[2], m));
    }
    for (int i = 0; i < d.n_light; i++) {
        const double* L = d.lights + (size_t)i * 11;
        o->push_back(new Light(get3(L), dvec3(0, 0, 0), get3(L + 3), L[6]));
    }
    for (int i = 0; i < d.n_fog; i++) {
        const double* q = d.fog + (size_t)i * 12;
        HeightFog* hf = new HeightFog(get3(q), get3(q + 3), get3(q + 6), q[9], q[10], (int)q[11]);
        hf->noiseGrid.assign(d.fog_grid + d.fog_grid_off[i], d.fog_grid + d.fog_grid_off[i + 1]);
        o->push_back(hf);
    }
}
}  // namespace gi

inline void loadScene(Octree* o, RayTracer& r, const char* fname)
{
    gih_scene* s = gih_scene_create();
    if (gih_load_scn(s, fname) != 0) { fprintf(stderr, "error while loading scene %s: %s\n", fname, gih_last_error(s)); gih_scene_destroy(s); return; }
    gih_settings st;
    gih_get_settings(s, &st);
    r.photons = st.photons; r.photon_depth = st.photon_depth;
    r.min_samples = st.min_samples; r.max_samples = st.max_samples; r.noise_thresh = st.noise_thresh;
    r.ambient = gi::dvec3(st.ambient[0], st.ambient[1], st.ambient[2]);
    r._camera.pos = gi::dvec3(st.cam_pos[0], st.cam_pos[1], st.cam_pos[2]);
    r._camera.setDir(gi::dvec3(st.cam_forward[0], st.cam_forward[1], st.cam_forward[2]));
    gi::push_loaded_entities(o, s, nullptr);
    gih_scene_destroy(s);
}
inline void loadOBJ(Octree* o, const char* fname, gi::dvec3 pos, gi::dvec3 rotation, std::vector<const Material*> materials)
{
    gih_scene* s = gih_scene_create();
    const double m[9] = {1, 1, 1, 0, 0, 0, 0, 0, 0}, p[3] = {pos.x, pos.y, pos.z}, rot[3] = {rotation.x, rotation.y, rotation.z};
    gih_add_material(s, m);
    if (gih_load_obj(s, fname, p, rot, 0) != 0) { fprintf(stderr, "error while opening file: %s\n", fname); gih_scene_destroy(s); return; }
    gi::push_loaded_entities(o, s, &materials);
    gih_scene_destroy(s);
}
inline void loadOBJ(Octree* o, const char* fname, gi::dvec3 pos, gi::dvec3 rotation, const Material& material)
{
    std::vector<const Material*> ms = {&material};
    loadOBJ(o, fname, pos, rotation, ms);
}
