// include/gi/light.h -- mirrors include/light.h:10-58 of the reference (fields the render path reads; dir / angle are written
// by Octree::rebuild exactly as include/octree.cpp:60-102 does).
#pragma once
#include "vec.h"
struct Light {
    Light(gi::dvec3 position, gi::dvec3 color, double radius) : pos(position), col(color), rad(radius) {}
    Light(gi::dvec3 position, gi::dvec3 target, gi::dvec3 color, double radius) : pos(position), col(color), rad(radius)
    {
#ifdef GI_USE_GLM
        dir = glm::normalize(target - position);
#else
        dir = gi::normalize(target - position);
#endif
    }
    gi::dvec3 dir;
    double angle = .125;
    gi::dvec3 pos, col;
    double rad = 0;
};
