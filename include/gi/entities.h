// include/gi/entities.h -- mirrors include/entities.h of the reference for the entity kinds of this path: Entity (interface
// data), vertex, triangle.  Intersection is not done on the host any more: RayTracer::trace / visible run on the GPU over the
// flattened tables (include/gi_hip.h); cone and the mesh generators are not on this path.
#pragma once
#include <array>
#include "material.h"
#include "vec.h"
struct Entity {
    Entity(const Material& m) : material(m) {}
    virtual ~Entity() {}
    gi::dvec3 pos = gi::dvec3(0, 0, 0);
    gi::dvec3 rot = gi::dvec3(0, 0, 0);
    Material material;
};
struct sphere : Entity {    // include/entities.h:51-142: analytic sphere, intersected on the GPU like the triangles
    double rad;
    sphere(gi::dvec3 position, double radius, const Material& m) : Entity(m), rad(radius) { pos = position; }
};
struct vertex {
    gi::dvec3 pos = gi::dvec3(0, 0, 0);
    gi::dvec3 norm = gi::dvec3(0, 0, 0);
    gi::dvec2 texCoord = gi::dvec2(0, 0);
    vertex() {}
    vertex(gi::dvec3 position, gi::dvec3 normal, gi::dvec2 uv) : pos(position), texCoord(uv) { norm = nrm(normal); }
    vertex(gi::dvec3 position, gi::dvec3 normal) : pos(position) { norm = nrm(normal); }
    vertex(gi::dvec3 position) : pos(position) {}
  private:
    static gi::dvec3 nrm(gi::dvec3 v)
    {
#ifdef GI_USE_GLM
        return glm::normalize(v);
#else
        return gi::normalize(v);
#endif
    }
};
struct triangle : Entity {
    std::array<vertex, 3> vertices;
    triangle(vertex v1, vertex v2, vertex v3, const Material& m) : Entity(m) { vertices = {{v1, v2, v3}}; }
};
