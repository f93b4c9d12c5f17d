// include/gi/photonMap.h -- the photon map of the reference (include/photonMap.h:13-49) for its callers: PhotonMap(min, max), reserve,
// push_back(Photon*), rebuild(), getInRange(pos, scale, dist), valid, _root.  rebuild() (include/photonMap.cpp:33-47) runs the host builder
// (PhotonMap::Node::partition, node for node) over the photons pushed so far, keeps the flattened tables for the upload to the GPU and
// mirrors the tree into `_root`.  getInRange and the Node queries are host-side conveniences over `_root` with the reference's candidate
// rule (the leaf that contains pos, grown by EPSILON, and every leaf touching it); the renderer's gather is the HIP kernel.
#pragma once
#include <array>
#include <memory>
#include <stdexcept>
#include <vector>
#include "bbox.h"
#include "photon.h"

class PhotonMap {
  public:
    struct Node {
        explicit Node(const BoundingBox& bbox) : _bbox(bbox) {}
        void partition() {}   // the tree arrives partitioned from PhotonMap::rebuild
        bool is_leaf() const { return _children[0] == nullptr; }   // a split node always has all eight children
        // photons of every leaf under this node whose (closed) box touches `bbox`, include/photonMap.cpp:71-92
        void get(BoundingBox bbox, std::vector<Photon*>& res, double = 0) const
        {
            if (bbox.dx() <= 0) return;
            std::vector<const Node*> todo = {this};
            while (!todo.empty()) {
                const Node* n = todo.back();
                todo.pop_back();
                if (n->is_leaf()) { res.insert(res.end(), n->_entities.begin(), n->_entities.end()); continue; }
                for (int i = 7; i >= 0; i--) if (n->_children[i]->_bbox.intersect(bbox)) todo.push_back(n->_children[i].get());
            }
        }
        // box (grown by EPSILON) of the leaf whose half-open box contains pos; a box of -inf when no child does, include/photonMap.cpp:115-134
        BoundingBox getBounds(gi::dvec3& pos) const
        {
            const Node* n = this;
            while (!n->is_leaf()) {
                int i = 0;
                while (i < 8 && !n->_children[i]->_bbox.contains(pos)) i++;
                if (i == 8) return BoundingBox(gi::dvec3(-INFINITY, -INFINITY, -INFINITY), gi::dvec3(-INFINITY, -INFINITY, -INFINITY));
                n = n->_children[i].get();
            }
            return BoundingBox(n->_bbox.min - gi::dvec3(GI_EPSILON, GI_EPSILON, GI_EPSILON), n->_bbox.max + gi::dvec3(GI_EPSILON, GI_EPSILON, GI_EPSILON));
        }
        void getInRange(gi::dvec3 pos, std::vector<Photon*>& res, double = 0) const   // photons of the leaves whose half-open box contains pos
        {
            if (is_leaf()) { res.insert(res.end(), _entities.begin(), _entities.end()); return; }
            for (int i = 0; i < 8; i++) if (_children[i]->_bbox.contains(pos)) _children[i]->getInRange(pos, res);
        }
        BoundingBox _bbox;
        std::vector<Photon*> _entities;
        std::array<std::unique_ptr<Node>, 8> _children;
    };

    PhotonMap(gi::dvec3 min, gi::dvec3 max) : _root(BoundingBox(min, max)), _h(gih_scene_create()) {}
    ~PhotonMap() { gih_scene_destroy(_h); }
    PhotonMap(const PhotonMap&) = delete;
    PhotonMap& operator=(const PhotonMap&) = delete;

    void reserve(int n) { _root._entities.reserve((size_t)n); }
    void push_back(Photon* p) { _root._entities.push_back(p); _all.push_back(p); }
    int size() const { return (int)_all.size(); }

    void rebuild()
    {
        std::vector<double> flat;
        flat.reserve(_all.size() * 9);
        for (const Photon* p : _all) { const double v[9] = {p->origin.x, p->origin.y, p->origin.z, p->dir.x, p->dir.y, p->dir.z, p->col.x, p->col.y, p->col.z}; flat.insert(flat.end(), v, v + 9); }
        const double box[6] = {_root._bbox.min.x, _root._bbox.min.y, _root._bbox.min.z, _root._bbox.max.x, _root._bbox.max.y, _root._bbox.max.z};
        if (gih_build_photon_map_in_box(_h, box, size(), flat.data()) != 0) throw std::runtime_error(std::string("PhotonMap::rebuild: ") + gih_last_error(_h));
        gi_photon_map_desc d;
        gih_get_photon_desc(_h, &d);
        auto box_of = [&](int n) { const double* b = d.node_bbox + (size_t)n * 6; return BoundingBox(gi::dvec3(b[0], b[1], b[2]), gi::dvec3(b[3], b[4], b[5])); };
        auto fill = [&](auto&& self, Node& node, int n) -> void {
            node._entities.clear();
            for (int k = d.node_off[n]; k < d.node_off[n + 1]; k++) node._entities.push_back(_all[(size_t)d.node_idx[k]]);
            for (int c = 0; c < 8; c++) {
                const int ch = d.node_child[(size_t)n * 8 + c];
                node._children[c].reset();
                if (ch >= 0) { node._children[c].reset(new Node(box_of(ch))); self(self, *node._children[c], ch); }
            }
        };
        if (d.n_node > 0) fill(fill, _root, 0);
        valid = true;
    }

    // candidates of samplePhotons: every photon of the leaves touching the (EPSILON-grown) leaf that contains pos, include/photonMap.cpp:50-66
    std::vector<Photon*> getInRange(gi::dvec3& pos, double& scale, double dist) const
    {
        std::vector<Photon*> res;
        res.reserve(256);
        const BoundingBox bounds = _root.getBounds(pos);
        scale = bounds.dx();
        _root.get(bounds, res, dist);
        return res;
    }

    bool valid = false;
    Node _root;

    gih_scene* handle() const { return _h; }   // flattened tables of the last rebuild (upload to the GPU)

  private:
    gih_scene* _h;
    std::vector<Photon*> _all;
};
