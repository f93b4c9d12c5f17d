// tests/cpp/ref_callers_driver.cpp -- the reference's main.cpp flow with the REFERENCE'S OWN loaders (its unmodified sceneLoader.cpp and
// meshLoader.cpp, compiled from where they lie by tests/test_reference_callers.py) on top of the drop-in headers of include/gi/: what a
// maintainer gets after swapping the headers.  Dumps the flattened tables Octree::rebuild leaves for the GPU, and answers host-side
// queries (Octree::intersectSorted / intersect over _root, PhotonMap::rebuild / getInRange) for rays / points the test hands in.
// usage: driver scene.scn io_dir      (io_dir: inputs *.f64 written by the test, outputs written here)
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#include "camera.h"
#include "util.h"
#include "raytracer.h"
#include "meshLoader.h"
#include "sceneLoader.h"

template <class T> static void put(const std::string& dir, const char* name, const T* p, size_t n)
{
    FILE* f = fopen((dir + "/" + name).c_str(), "wb");
    if (n) fwrite(p, sizeof(T), n, f);
    fclose(f);
}
static std::vector<double> get(const std::string& dir, const char* name)
{
    std::vector<double> v;
    FILE* f = fopen((dir + "/" + name).c_str(), "rb");
    if (!f) return v;
    double buf[4096];
    size_t n;
    while ((n = fread(buf, sizeof(double), 4096, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}
static void number(const Octree::Node* n, std::map<const Octree::Node*, int>& id)
{
    const int me = (int)id.size();
    id[n] = me;
    for (int i = 0; i < 8; i++) if (n->_children[i]) number(n->_children[i].get(), id);
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string io = argv[2];
    Camera camera({10, 5, 0}, {0, 0, 0});                          // main.cpp:28
    RayTracer raytracer(camera);                                   // main.cpp:30
    Octree* scene = new Octree();                                  // main.cpp:34
    loadScene(scene, raytracer, argv[1]);                          // main.cpp:36-39 -- the reference's loader
    raytracer.setScene(scene);                                     // main.cpp:41
    RayTracer copy = raytracer;                                    // gui.h:19 / viewer.h:16 pass it by value
    scene->rebuild();
    printf("sizeof(triangle) %zu sizeof(Octree::Node) %zu lights %zu photons %d samples %d..%d\n", sizeof(triangle), sizeof(Octree::Node), scene->lights.size(), copy.photons,
           copy.min_samples, copy.max_samples);

    gi_scene_desc d;
    if (gih_get_scene_desc(scene->handle(), &d) != 0) return 3;
    put(io, "tri_pos.f64", d.tri_pos, (size_t)d.n_tri * 9);
    put(io, "tri_nrm.f64", d.tri_nrm, (size_t)d.n_tri * 9);
    put(io, "tri_uv.f64", d.tri_uv, (size_t)d.n_tri * 6);
    std::vector<double> mat_rows;
    for (int i = 0; i < d.n_tri; i++) mat_rows.insert(mat_rows.end(), d.mats + (size_t)d.tri_mat[i] * 9, d.mats + (size_t)d.tri_mat[i] * 9 + 9);
    put(io, "tri_mat.f64", mat_rows.data(), mat_rows.size());
    put(io, "lights.f64", d.lights, (size_t)d.n_light * 11);
    put(io, "oct_bbox.f64", d.node_bbox, (size_t)d.n_node * 6);
    put(io, "oct_child.i32", d.node_child, (size_t)d.n_node * 8);
    put(io, "oct_ent_off.i32", d.node_ent_off, (size_t)d.n_node + 1);
    put(io, "oct_ent_idx.i32", d.node_ent_idx, (size_t)d.node_ent_off[d.n_node]);

    // ---- the tree as a caller walking _root sees it: boxes and entity counts in pre-order
    std::map<const Octree::Node*, int> id;
    number(&scene->_root, id);
    std::vector<double> rb(id.size() * 6);
    std::vector<int32_t> rn(id.size());
    for (auto& kv : id) {
        const BoundingBox& b = kv.first->_bbox;
        const double v[6] = {b.min.x, b.min.y, b.min.z, b.max.x, b.max.y, b.max.z};
        for (int k = 0; k < 6; k++) rb[(size_t)kv.second * 6 + k] = v[k];
        rn[(size_t)kv.second] = (int32_t)kv.first->_entities.size();
    }
    put(io, "root_bbox.f64", rb.data(), rb.size());
    put(io, "root_nent.i32", rn.data(), rn.size());

    // ---- Octree::intersectSorted(ray, 0, inf) for the rays handed in (origin, unit direction)
    const std::vector<double> rays = get(io, "rays.f64");
    std::vector<int32_t> ln, lo = {0};
    std::vector<double> lt;
    for (size_t i = 0; i + 6 <= rays.size(); i += 6) {
        Ray r(glm::dvec3(rays[i], rays[i + 1], rays[i + 2]), glm::dvec3(rays[i + 3], rays[i + 4], rays[i + 5]));
        r.dir = glm::dvec3(rays[i + 3], rays[i + 4], rays[i + 5]);            // the exact direction of the fixture, not its re-normalisation
        r.invDir = glm::dvec3(1.0 / r.dir);
        for (auto& pr : scene->intersectSorted(r, 0, INFINITY)) { ln.push_back(id.at(pr.first)); lt.push_back(pr.second); }
        lo.push_back((int32_t)ln.size());
    }
    put(io, "leaf_node.i32", ln.data(), ln.size());
    put(io, "leaf_off.i32", lo.data(), lo.size());
    put(io, "leaf_t0.f64", lt.data(), lt.size());

    // ---- Octree::intersect(shadow ray, 0, sqrt(maxt) - SHADOW_BIAS): candidate counts (origin, target pairs)
    const std::vector<double> sq = get(io, "shadow.f64");
    std::vector<int32_t> nc;
    for (size_t i = 0; i + 6 <= sq.size(); i += 6) {
        const glm::dvec3 o(sq[i], sq[i + 1], sq[i + 2]), ld = glm::dvec3(sq[i + 3], sq[i + 4], sq[i + 5]) - o;
        Ray sr(o, ld);
        nc.push_back((int32_t)scene->intersect(sr, 0, sqrt(vecLengthSquared(ld)) - SHADOW_BIAS).size());
    }
    put(io, "shadow_ncand.i32", nc.data(), nc.size());

    // ---- PhotonMap(min, max) + push_back + rebuild + getInRange in the scene's root box
    const std::vector<double> ph = get(io, "photons.f64"), gq = get(io, "gather_q.f64");
    PhotonMap pm(scene->_root._bbox.min, scene->_root._bbox.max);
    pm.reserve((int)(ph.size() / 9));
    for (size_t i = 0; i + 9 <= ph.size(); i += 9) pm.push_back(new Photon(glm::dvec3(ph[i], ph[i + 1], ph[i + 2]), glm::dvec3(ph[i + 3], ph[i + 4], ph[i + 5]), glm::dvec3(ph[i + 6], ph[i + 7], ph[i + 8])));
    pm.rebuild();
    std::vector<int32_t> gc;
    for (size_t i = 0; i + 6 <= gq.size(); i += 6) {
        glm::dvec3 pos(gq[i], gq[i + 1], gq[i + 2]);
        double scale = 0;
        gc.push_back((int32_t)pm.getInRange(pos, scale, 0).size());
    }
    put(io, "gather_ncand.i32", gc.data(), gc.size());
    gi_photon_map_desc pd;
    gih_get_photon_desc(pm.handle(), &pd);
    put(io, "pm_bbox.f64", pd.node_bbox, (size_t)pd.n_node * 6);
    printf("ok: %d entities, %d nodes, %zu photons in %d photon-map nodes\n", d.n_tri, d.n_node, ph.size() / 9, pd.n_node);
    return 0;
}
