// include/gi/material.h -- mirrors include/material.h:11-29,84-100 of the reference: constant-colour texture and Material.
// checkerboard / imageTexture are outside this path; Octree::rebuild refuses other textures.
#pragma once
#include "vec.h"
struct texture {
    texture(gi::dvec3 col) : color(col) {}
    virtual ~texture() {}
    virtual gi::dvec3 get(gi::dvec2&) { return color; }
    virtual double getAlpha(gi::dvec2&) { return 1; }
    gi::dvec3 color;
};
struct Material {
    Material(texture* dif, texture* em, double r, double o, double i = 1) : diffuse(dif), emissive(em), roughness(r), opacity(o), IOR(i) {}
    double getAlpha(gi::dvec2& uv) { return opacity * diffuse->getAlpha(uv); }
    texture* diffuse;
    texture* emissive;
    double roughness, opacity, IOR;
};
