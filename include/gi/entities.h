// include/gi/entities.h -- the entity classes of the reference (include/entities.h:17-49 Entity, :51-142 sphere, :302-327 vertex, :329-558
// triangle, :721-738 quadMesh, :740-796 boxMesh) for the callers that build scenes: the loaders, the GUI, user code.
// Data layout and constructors follow the reference.  The single-ray virtuals (intersect, boundingBox, intersect(BoundingBox)) evaluate the
// kernels' own per-lane functions on the host (detail.h) and the host tree builder's box tests (gi_host.h); rendering never calls them --
// Octree::rebuild flattens the entities into the tables the GPU walks.  `cone`, `sphereMesh` and `coneMesh` are not provided: no scene
// keyword reaches them (include/sceneLoader.cpp).
#pragma once
#include <array>
#include <vector>
#include "bbox.h"
#include "material.h"
#include "ray.h"
#ifdef GI_USE_GLM   // what the reference's entities.h pulls in for its callers (meshLoader.cpp uses glm::eulerAngleXYZ)
#include <glm/gtx/euler_angles.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtx/component_wise.hpp>
#endif
class Octree;

struct Entity {
    Entity() : material(Material(new texture(gi::dvec3(1, 0, 0)), new texture(gi::dvec3(0, 0, 0)), .75, 1)) {}
    Entity(const Material& m) : material(m) {}
    virtual ~Entity() {}
    // hit point, (unnormalised-interpolated) normal and uv of the nearest intersection along the ray
    virtual bool intersect(const Ray& ray, gi::dvec3& intersect, gi::dvec3& normal, gi::dvec2& uv) = 0;
    virtual bool intersect(const Ray& ray, double& t)
    {
        gi::dvec3 hit(0, 0, 0), norm(0, 0, 0);
        gi::dvec2 uv(0, 0);
        const bool i = intersect(ray, hit, norm, uv);
        t = gi::length(gi::to_v3(hit - ray.origin));
        return i;
    }
    virtual bool intersect(BoundingBox) { return false; }
    virtual BoundingBox boundingBox() const = 0;
    gi::dvec3 pos = gi::dvec3(0, 0, 0);
    gi::dvec3 rot = gi::dvec3(0, 0, 0);
    Material material;

  protected:
    // the host tree builder's entity tests (the ones Octree::Node::partition is built from): kind 0 triangle, 1 sphere
    static BoundingBox host_bbox(int kind, const double* pos9)
    {
        double b[6];
        gih_entity_bbox(kind, pos9, b);
        return BoundingBox(gi::dvec3(b[0], b[1], b[2]), gi::dvec3(b[3], b[4], b[5]));
    }
    static bool host_overlaps(int kind, const double* pos9, const BoundingBox& bb)
    {
        const double b[6] = {bb.min.x, bb.min.y, bb.min.z, bb.max.x, bb.max.y, bb.max.z};
        return gih_entity_overlaps_box(kind, pos9, b) > 0;
    }
};

struct sphere : Entity {    // analytic sphere, include/entities.h:51-142
    double rad;
    sphere(gi::dvec3 position, double radius, const Material& m) : Entity(m), rad(radius) { pos = position; }
    bool intersect(const Ray& ray, gi::dvec3& intersect, gi::dvec3& normal, gi::dvec2& uv) override
    {
        gi::TriGeom g;
        flat(g.p0);
        g.e1[0] = rad;
        double u, v;
        gi::V3 hp;
        if (!gi::ent_hit<GI_FEAT_SPHERES>(g, 4u, gi::to_lane_ray(ray), u, v, hp)) return false;
        intersect = gi::from_v3(hp);
        normal = gi::vnormalize(intersect - pos);
        double tu = 0, tv = 0;
        gi::Scene none{};
        gi::ent_uv(none, g, 4u, 0, u, v, hp, tu, tv);
        uv = gi::dvec2(tu, tv);
        return true;
    }
    BoundingBox boundingBox() const override { double p[9]; flat(p); return host_bbox(1, p); }
    bool intersect(BoundingBox bb) override { double p[9]; flat(p); return host_overlaps(1, p, bb); }

  private:
    void flat(double* p) const { p[0] = pos.x; p[1] = pos.y; p[2] = pos.z; p[3] = rad; for (int k = 4; k < 9; k++) p[k] = 0; }
};

struct vertex {
    gi::dvec3 pos = gi::dvec3(0, 0, 0);
    gi::dvec3 norm = gi::dvec3(0, 0, 0);
    gi::dvec2 texCoord = gi::dvec2(0, 0);
    vertex() {}
    vertex(gi::dvec3 position, gi::dvec3 normal, gi::dvec2 uv) : pos(position), norm(gi::vnormalize(normal)), texCoord(uv) {}
    vertex(gi::dvec3 position, gi::dvec3 normal) : pos(position), norm(gi::vnormalize(normal)) {}
    vertex(gi::dvec3 position) : pos(position) {}
};

struct triangle : Entity {
    std::array<vertex, 3> vertices;
    gi::dvec3 norm;          // face normal
    double inv_area;
    triangle(vertex v1, vertex v2, vertex v3, const Material& m) : Entity(m)
    {
        vertices = {{v1, v2, v3}};
        norm = gi::from_v3(gi::normalize(gi::cross(gi::to_v3(v2.pos - v1.pos), gi::to_v3(v3.pos - v1.pos))));
        inv_area = 1.0 / gi::length(gi::cross(gi::to_v3(v1.pos - v2.pos), gi::to_v3(v1.pos - v3.pos)));
    }
    bool smooth() const   // all three vertex normals given (include/entities.h:478)
    {
        return gi::len2(gi::to_v3(vertices[0].norm)) > 0 && gi::len2(gi::to_v3(vertices[1].norm)) > 0 && gi::len2(gi::to_v3(vertices[2].norm)) > 0;
    }
    // Moeller-Trumbore in double, include/entities.h:443-490: the kernels' tri_hit on this triangle's test record
    bool intersect(const Ray& ray, gi::dvec3& intersect, gi::dvec3& normal, gi::dvec2& uv) override
    {
        gi::TriGeom g;
        const gi::V3 p0 = gi::to_v3(vertices[0].pos), e1 = gi::to_v3(vertices[1].pos) - p0, e2 = gi::to_v3(vertices[2].pos) - p0;
        g.p0[0] = p0.x; g.p0[1] = p0.y; g.p0[2] = p0.z;
        g.e1[0] = e1.x; g.e1[1] = e1.y; g.e1[2] = e1.z;
        g.e2[0] = e2.x; g.e2[1] = e2.y; g.e2[2] = e2.z;
        const gi::Ray r = gi::to_lane_ray(ray);
        double u, v, t;
        if (!gi::tri_hit(g, r, u, v, t)) return false;
        intersect = gi::from_v3(r.o + t * r.d);
        if (smooth()) {
            const double w = (1 - u - v);
            normal = gi::from_v3(w * gi::to_v3(vertices[0].norm) + u * gi::to_v3(vertices[1].norm) + v * gi::to_v3(vertices[2].norm));
            uv = gi::dvec2(w * vertices[0].texCoord.x + u * vertices[1].texCoord.x + v * vertices[2].texCoord.x,
                           w * vertices[0].texCoord.y + u * vertices[1].texCoord.y + v * vertices[2].texCoord.y);
        } else
            normal = norm;
        return true;
    }
    using Entity::intersect;
    bool intersect(BoundingBox bb) override { double p[9]; flat(p); return host_overlaps(0, p, bb); }
    BoundingBox boundingBox() const override { double p[9]; flat(p); return host_bbox(0, p); }
    void flat(double* p) const { for (int k = 0; k < 3; k++) gi::put3(p + 3 * k, vertices[k].pos); }
};

// Mesh generators: they only push triangles into the scene (include/entities.h:560).  Defined after Octree (octree.h includes this file).
struct quadMesh : Entity {
    quadMesh(Octree* o, gi::dvec3 v1, gi::dvec3 v2, gi::dvec3 v3, gi::dvec3 v4, const Material& m);
    bool intersect(const Ray&, gi::dvec3&, gi::dvec3&, gi::dvec2&) override { return false; }
    BoundingBox boundingBox() const override { return BoundingBox(gi::dvec3(0, 0, 0), gi::dvec3(1, 1, 1)); }
};
struct boxMesh : Entity {
    boxMesh(Octree* o, gi::dvec3 position, gi::dvec3 size, gi::dvec3 rotation, const Material& m);
    bool intersect(const Ray&, gi::dvec3&, gi::dvec3&, gi::dvec2&) override { return false; }
    BoundingBox boundingBox() const override { return BoundingBox(gi::dvec3(0, 0, 0), gi::dvec3(1, 1, 1)); }
};
#include "octree.h"
