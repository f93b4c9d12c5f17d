// include/gi/sceneLoader.h -- drop-in for loadScene (include/sceneLoader.h, include/sceneLoader.cpp:12-185): fills the Octree
// and writes the RayTracer's public settings, as the reference's loader does through `RayTracer& r`.
#pragma once
#include "raytracer.h"
inline void loadScene(Octree* o, RayTracer& r, const char* fname)
{
    if (gih_load_scn(o->handle(), fname) != 0) { fprintf(stderr, "error while loading scene %s: %s\n", fname, gih_last_error(o->handle())); return; }
    o->adopt_loaded_scene();
    gih_settings st;
    gih_get_settings(o->handle(), &st);
    r.photons = st.photons; r.photon_depth = st.photon_depth;
    r.min_samples = st.min_samples; r.max_samples = st.max_samples; r.noise_thresh = st.noise_thresh;
    r.ambient = gi::dvec3(st.ambient[0], st.ambient[1], st.ambient[2]);
    r._camera.pos = gi::dvec3(st.cam_pos[0], st.cam_pos[1], st.cam_pos[2]);
    r._camera.setDir(gi::dvec3(st.cam_forward[0], st.cam_forward[1], st.cam_forward[2]));
    // lights of the file, so that Octree::lights has the reference's content (include/sceneLoader.cpp:150-158)
    gi_scene_desc d;
    if (gih_build_octree(o->handle()) == 0 && gih_get_scene_desc(o->handle(), &d) == 0) {
        for (int i = 0; i < d.n_light; i++) {
            const double* L = d.lights + (size_t)i * 11;
            o->lights.push_back(new Light(gi::dvec3(L[0], L[1], L[2]), gi::dvec3(0, 0, 0), gi::dvec3(L[3], L[4], L[5]), L[6]));
        }
    }
}
