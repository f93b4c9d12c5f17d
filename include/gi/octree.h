// include/gi/octree.h -- the scene container of the reference (include/octree.h:17-65) for its callers: loaders and generators push entities,
// lights and atmosphere entities; RayTracer::setScene takes the pointer.  rebuild() (include/octree.cpp:53-119) hands the entities to the
// host tree builder (gi_raytracer_amd/csrc/gi_host.cpp: the reference's light-cone precompute and Octree::Node::partition, node for node),
// keeps the flattened tables for the upload to the GPU and mirrors the tree into `_root` (Node: box, entity pointers, eight children), so
// that code walking `_root` sees the reference's structure.  intersect / intersectSorted / atmosphere* are host-side queries over `_root`
// for such callers; the renderer never uses them -- its traversal is the HIP kernels' walk over the uploaded tables.
#pragma once
#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "atmosphere.h"
#include "bbox.h"
#include "light.h"
struct Entity;
struct triangle;
struct sphere;

class Octree {
  public:
    struct Node {
        explicit Node(const BoundingBox& bbox) : _bbox(bbox) {}
        void partition() {}   // the tree arrives partitioned from Octree::rebuild; kept for source compatibility
        bool is_leaf() const { for (const auto& c : _children) if (c) return false; return true; }
        // entities of every leaf under this node whose box the segment touches (duplicates kept), include/octree.cpp:256-282
        void intersect(const Ray& ray, std::vector<Entity*>& res, double tmin, double tmax, float* = nullptr, float* = nullptr, int = 0) const
        {
            std::vector<const Node*> todo = {this};
            while (!todo.empty()) {
                const Node* n = todo.back();
                todo.pop_back();
                if (!n->_bbox.intersectSimple(ray, tmin, tmax)) continue;
                if (n->is_leaf()) { res.insert(res.end(), n->_entities.begin(), n->_entities.end()); continue; }
                for (int i = 7; i >= 0; i--) if (n->_children[i]) todo.push_back(n->_children[i].get());   // popped in child order 0..7
            }
        }
        // non-empty leaves the ray enters, ascending entry distance, equal distances in depth-first order (include/octree.cpp:285-313);
        // `res` may already hold a sorted list: the new leaves are merged in behind equal keys, as repeated insertion would
        void intersectSorted(const Ray& ray, std::vector<std::pair<const Node*, double>>& res, double tmin, double tmax) const
        {
            std::vector<std::pair<const Node*, double>> found;
            std::vector<const Node*> todo = {this};
            while (!todo.empty()) {
                const Node* n = todo.back();
                todo.pop_back();
                double t0, t1;
                if (!n->_bbox.intersect(ray, tmin, tmax, t0, t1)) continue;
                if (n->is_leaf()) { if (!n->_entities.empty()) found.push_back({n, t0}); continue; }
                for (int i = 7; i >= 0; i--) if (n->_children[i]) todo.push_back(n->_children[i].get());
            }
            const size_t old = res.size();
            res.insert(res.end(), found.begin(), found.end());
            std::stable_sort(res.begin() + (std::ptrdiff_t)old, res.end(), [](const std::pair<const Node*, double>& a, const std::pair<const Node*, double>& b) { return a.second < b.second; });
            std::inplace_merge(res.begin(), res.begin() + (std::ptrdiff_t)old, res.end(), [](const std::pair<const Node*, double>& a, const std::pair<const Node*, double>& b) { return a.second < b.second; });
        }
        BoundingBox _bbox;
        std::vector<Entity*> _entities;
        std::array<std::unique_ptr<Node>, 8> _children;
    };

    Octree(gi::dvec3 min = gi::dvec3(0, 0, 0), gi::dvec3 max = gi::dvec3(0, 0, 0)) : _root(BoundingBox(min, max)), _h(gih_scene_create()) {}
    ~Octree() { gih_scene_destroy(_h); }
    Octree(const Octree&) = delete;
    Octree& operator=(const Octree&) = delete;

    std::vector<Light*> lights;
    std::vector<AtmosphereEntity*> at;

    // the caller keeps ownership by convention, as in the reference (raw pointers, never freed)
    void push_back(Entity* object);   // grows _root._bbox by the entity's box, as include/octree.cpp:25-38 (defined below: needs Entity)
    void push_back(Light* light) { lights.push_back(light); valid = false; }
    void push_back(AtmosphereEntity* entity) { at.push_back(entity); valid = false; }

    void rebuild();   // defined below entities (needs triangle / sphere)

    std::vector<Entity*> intersect(const Ray& ray, double tmin, double tmax) const
    {
        std::vector<Entity*> res;
        res.reserve(256);
        _root.intersect(ray, res, tmin, tmax);
        return res;
    }
    std::vector<std::pair<const Node*, double>> intersectSorted(const Ray& ray, double tmin, double tmax) const
    {
        std::vector<std::pair<const Node*, double>> res;
        _root.intersectSorted(ray, res, tmin, tmax);
        return res;
    }
    // density (times the march step), colour and scatter of the atmosphere at a point, include/octree.cpp:214-226
    double atmosphereDensity(const gi::dvec3& pos, gi::dvec3& col, double& scatter)
    {
        (void)scatter;   // the reference leaves it untouched as well
        double d = 0;
        for (AtmosphereEntity* a : at)
            if (a->bbox.contains(pos)) { col = a->col; d += GI_RAYMARCH_STEPSIZE * a->density(pos); }
        return d;
    }
    // clips [mint, maxt] to the atmosphere boxes the ray meets; the lower bound starts at 0, include/octree.cpp:229-251
    bool atmosphereBounds(const Ray& r, double& mint, double& maxt)
    {
        double lo = 0, hi = 0;
        bool any = false;
        for (AtmosphereEntity* a : at) {
            double t0 = 0, t1 = 0;
            if (a->bbox.intersect(r, mint, maxt, t0, t1)) { lo = std::min(lo, t0); hi = std::max(hi, t1); any = true; }
        }
        mint = std::max(mint, lo);
        maxt = std::min(maxt, hi);
        return any;
    }

    bool valid = false;
    Node _root;

    // ---- what the GPU side needs (not part of the reference's surface)
    gih_scene* handle() const { return _h; }
    const std::vector<Entity*>& entities() const { return _all; }   // insertion order = entity index in the flattened tables
    int index_of(const Entity* e) const { auto it = _index.find(e); return it == _index.end() ? -1 : it->second; }

  private:
    gih_scene* _h;
    std::vector<Entity*> _pending;   // pushed since the last rebuild
    std::vector<Entity*> _all;
    std::map<const Entity*, int> _index;
};

#include "entities.h"

// Octree::rebuild, include/octree.cpp:53-119: every entity pushed so far goes to the host builder (light cones + partition), the
// flattened tables stay in the handle for the upload, the tree is mirrored into _root, the lights get their dir / angle.
inline void Octree::rebuild()
{
    for (Entity* e : _pending) { _index[e] = (int)_all.size(); _all.push_back(e); }
    _pending.clear();
    gih_scene* s = gih_scene_create();
    auto fail = [&](const std::string& what) { gih_scene_destroy(s); throw std::runtime_error("Octree::rebuild: " + what); };
    std::map<const texture*, int> tex_id;          // each texture object registered once (include/material.h:10-81)
    std::map<std::vector<double>, int> mat_id;
    auto tex_of = [&](const texture* t) {
        auto it = tex_id.find(t);
        if (it != tex_id.end()) return it->second;
        const int id = t->gi_register(s);
        if (id < 0) fail(gih_last_error(s));
        return tex_id[t] = id;
    };
    for (Entity* e : _all) {
        const int dt = tex_of(e->material.diffuse), et = tex_of(e->material.emissive);
        const std::vector<double> key = {(double)dt, (double)et, e->material.roughness, e->material.opacity, e->material.IOR};
        auto mit = mat_id.find(key);
        const int mi = mit != mat_id.end() ? mit->second : (mat_id[key] = gih_add_material_tex(s, dt, et, e->material.roughness, e->material.opacity, e->material.IOR));
        if (const sphere* sp = dynamic_cast<const sphere*>(e)) {
            const double c[3] = {sp->pos.x, sp->pos.y, sp->pos.z};
            gih_add_sphere(s, c, sp->rad, mi);
            continue;
        }
        const triangle* t = dynamic_cast<const triangle*>(e);
        if (!t) fail("only triangle and sphere entities are on the GPU path (mesh generators push triangles)");
        double pos[9], nrm[9], uv[6];
        t->flat(pos);
        for (int k = 0; k < 3; k++) { gi::put3(nrm + 3 * k, t->vertices[k].norm); uv[k * 2] = t->vertices[k].texCoord.x; uv[k * 2 + 1] = t->vertices[k].texCoord.y; }
        const int32_t mm = mi;
        gih_add_triangles(s, 1, pos, nrm, uv, &mm);
    }
    for (Light* l : lights) {
        const double p[3] = {l->pos.x, l->pos.y, l->pos.z}, c[3] = {l->col.x, l->col.y, l->col.z};
        gih_add_light(s, p, c, l->rad);
    }
    for (AtmosphereEntity* a : at) {
        HeightFog* hf = dynamic_cast<HeightFog*>(a);
        if (!hf) fail("only HeightFog atmosphere entities are on the GPU path");
        const double q[12] = {hf->pos.x, hf->pos.y, hf->pos.z, hf->s.x, hf->s.y, hf->s.z, hf->col.x, hf->col.y, hf->col.z, hf->d, hf->sc, 1.0};
        // the grid the entity was constructed with (its size follows the noise scale given then; nscale itself is 1 afterwards, as in the reference)
        if (gih_add_height_fog_grid(s, q, hf->noiseGrid.data(), (int32_t)hf->noiseGrid.size()) != 0) fail(gih_last_error(s));
    }
    if (gih_build_octree(s) != 0) fail(gih_last_error(s));
    gih_scene_destroy(_h);
    _h = s;
    gi_scene_desc d;
    gih_get_scene_desc(_h, &d);
    // light dir / angle are outputs of rebuild in the reference (include/octree.cpp:81-101)
    for (size_t i = 0; i < lights.size() && (int)i < d.n_light; i++) {
        const double* L = d.lights + i * 11;
        lights[i]->dir = gi::dvec3(L[7], L[8], L[9]);
        lights[i]->angle = L[10];
    }
    // mirror of the builder's pre-order arrays (children 0..7 as Octree::Node::partition numbers them)
    auto box_of = [&](int n) { const double* b = d.node_bbox + (size_t)n * 6; return BoundingBox(gi::dvec3(b[0], b[1], b[2]), gi::dvec3(b[3], b[4], b[5])); };
    auto fill = [&](auto&& self, Node& node, int n) -> void {
        node._bbox = box_of(n);
        node._entities.clear();
        for (int k = d.node_ent_off[n]; k < d.node_ent_off[n + 1]; k++) node._entities.push_back(_all[(size_t)d.node_ent_idx[k]]);
        for (int c = 0; c < 8; c++) {
            const int ch = d.node_child[(size_t)n * 8 + c];
            node._children[c].reset();
            if (ch >= 0) { node._children[c].reset(new Node(box_of(ch))); self(self, *node._children[c], ch); }
        }
    };
    if (d.n_node > 0) fill(fill, _root, 0);
    valid = true;
}

// Octree::push_back(Entity*), include/octree.cpp:25-38: the box of the first entity initialises the root box, every entity widens it -- a
// caller may read _root._bbox before rebuild() (RayTracer::setScene sizes the photon map with it)
inline void Octree::push_back(Entity* object)
{
    const BoundingBox b = object->boundingBox();
    if (_root._entities.empty() && _all.empty()) { _root._bbox.max = b.max; _root._bbox.min = b.min; }
    _root._entities.push_back(object);
    _pending.push_back(object);
    _root._bbox.max = gi::dvec3(std::fmax(_root._bbox.max.x, b.max.x), std::fmax(_root._bbox.max.y, b.max.y), std::fmax(_root._bbox.max.z, b.max.z));
    _root._bbox.min = gi::dvec3(std::fmin(_root._bbox.min.x, b.min.x), std::fmin(_root._bbox.min.y, b.min.y), std::fmin(_root._bbox.min.z, b.min.z));
    valid = false;
}

inline quadMesh::quadMesh(Octree* o, gi::dvec3 v1, gi::dvec3 v2, gi::dvec3 v3, gi::dvec3 v4, const Material& m)
{
    o->push_back(new triangle(vertex(v1), vertex(v2), vertex(v3), m));
    o->push_back(new triangle(vertex(v3), vertex(v2), vertex(v4), m));
}
// the 12 triangles of a box: corners normalize((+-1, +-1, +-1)) * size, rotated by eulerAngleXYZ(rotation), moved to position
// (include/entities.h:740-785); the arithmetic is the host loader's (gih_box_mesh), which the `box` keyword of a .scn goes through as well
inline boxMesh::boxMesh(Octree* o, gi::dvec3 position, gi::dvec3 size, gi::dvec3 rotation, const Material& m)
{
    pos = position;
    const double p[3] = {position.x, position.y, position.z}, s[3] = {size.x, size.y, size.z}, r[3] = {rotation.x, rotation.y, rotation.z};
    double tri[108];
    gih_box_mesh(p, s, r, tri);
    for (int t = 0; t < 12; t++) o->push_back(new triangle(vertex(gi::get3(tri + t * 9)), vertex(gi::get3(tri + t * 9 + 3)), vertex(gi::get3(tri + t * 9 + 6)), m));
}
