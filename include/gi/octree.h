// include/gi/octree.h -- drop-in for the public surface of the reference's Octree (include/octree.h:17-65) on this path.
// push_back keeps the caller's raw pointers exactly as the reference does; rebuild() runs Octree::rebuild's light precompute and
// partition on the host (csrc/gi_host.cpp, node-for-node the reference's tree) and leaves the flattened tables that
// RayTracer uploads to the GPU.  The per-ray queries (intersect / intersectSorted) are gone from the host: they are the
// traversal inside the HIP kernels.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../gi_raytracer_amd/csrc/gi_host.h"
#include "atmosphere.h"
#include "entities.h"
#include "light.h"

class Octree {
  public:
    Octree(gi::dvec3 = gi::dvec3(0, 0, 0), gi::dvec3 = gi::dvec3(0, 0, 0)) : _h(gih_scene_create()) {}
    ~Octree() { gih_scene_destroy(_h); }
    Octree(const Octree&) = delete;
    Octree& operator=(const Octree&) = delete;

    std::vector<Light*> lights;
    std::vector<AtmosphereEntity*> at;
    bool valid = false;

    void push_back(Entity* object) { _entities.push_back(object); valid = false; }
    void push_back(Light* light) { lights.push_back(light); valid = false; }
    void push_back(AtmosphereEntity* entity) { at.push_back(entity); valid = false; }

    // Octree::rebuild, include/octree.cpp:53-119
    void rebuild()
    {
        gih_scene* fresh = gih_scene_create();
        if (_loaded) { gih_scene_destroy(fresh); fresh = nullptr; }   // filled by loadScene(): keep that scene
        gih_scene* s = fresh ? fresh : _h;
        if (fresh) {
            std::map<const texture*, int> tex_id;   // each texture object registered once (include/material.h:10-81)
            std::map<std::vector<double>, int> mat_id;
            auto tex_of = [&](const texture* t) {
                auto it = tex_id.find(t);
                if (it != tex_id.end()) return it->second;
                const int id = t->gi_register(s);
                if (id < 0) { std::string e = gih_last_error(s); gih_scene_destroy(fresh); throw std::runtime_error("Octree::rebuild: " + e); }
                tex_id[t] = id;
                return id;
            };
            for (Entity* e : _entities) {
                const int dt = tex_of(e->material.diffuse), et = tex_of(e->material.emissive);
                const std::vector<double> key = {(double)dt, (double)et, e->material.roughness, e->material.opacity, e->material.IOR};
                auto mit = mat_id.find(key);
                const int mi = mit != mat_id.end() ? mit->second : (mat_id[key] = gih_add_material_tex(s, dt, et, e->material.roughness, e->material.opacity, e->material.IOR));
                if (sphere* sp = dynamic_cast<sphere*>(e)) {
                    const double c[3] = {sp->pos.x, sp->pos.y, sp->pos.z};
                    gih_add_sphere(s, c, sp->rad, mi);
                    continue;
                }
                triangle* t = dynamic_cast<triangle*>(e);
                if (!t) { gih_scene_destroy(fresh); throw std::runtime_error("Octree::rebuild: only triangle and sphere entities are on the GPU path"); }
                double pos[9], nrm[9], uv[6];
                for (int k = 0; k < 3; k++) {
                    const vertex& v = t->vertices[k];
                    pos[k * 3] = v.pos.x; pos[k * 3 + 1] = v.pos.y; pos[k * 3 + 2] = v.pos.z;
                    nrm[k * 3] = v.norm.x; nrm[k * 3 + 1] = v.norm.y; nrm[k * 3 + 2] = v.norm.z;
                    uv[k * 2] = v.texCoord.x; uv[k * 2 + 1] = v.texCoord.y;
                }
                const int32_t mm = mi;
                gih_add_triangles(s, 1, pos, nrm, uv, &mm);
            }
            for (Light* l : lights) {
                const double p[3] = {l->pos.x, l->pos.y, l->pos.z}, c[3] = {l->col.x, l->col.y, l->col.z};
                gih_add_light(s, p, c, l->rad);
            }
            for (AtmosphereEntity* a : at) {
                HeightFog* hf = dynamic_cast<HeightFog*>(a);
                if (!hf) { gih_scene_destroy(fresh); throw std::runtime_error("Octree::rebuild: only HeightFog atmosphere entities are on the GPU path"); }
                const double q[12] = {hf->pos.x, hf->pos.y, hf->pos.z, hf->s.x, hf->s.y, hf->s.z, hf->col.x, hf->col.y, hf->col.z, hf->d, hf->sc, (double)hf->nscale};
                if (gih_add_height_fog(s, q, nullptr, 0, 0x9E3779B97F4A7C15ull) != 0) { std::string e = gih_last_error(s); gih_scene_destroy(fresh); throw std::runtime_error(e); }
            }
            gih_scene_destroy(_h);
            _h = fresh;
        }
        if (gih_build_octree(_h) != 0) throw std::runtime_error(std::string("Octree::rebuild: ") + gih_last_error(_h));
        // light dir / angle are outputs of rebuild in the reference (include/octree.cpp:81-101): mirror them back
        gi_scene_desc d;
        gih_get_scene_desc(_h, &d);
        for (size_t i = 0; i < lights.size() && (int)i < d.n_light; i++) {
            const double* L = d.lights + i * 11;
            lights[i]->dir = gi::dvec3(L[7], L[8], L[9]);
            lights[i]->angle = L[10];
        }
        valid = true;
    }

    gih_scene* handle() const { return _h; }
    // used by loadScene(): the C loader has filled the handle directly
    void adopt_loaded_scene() { _loaded = true; valid = false; }

  private:
    gih_scene* _h;
    std::vector<Entity*> _entities;
    bool _loaded = false;
};
