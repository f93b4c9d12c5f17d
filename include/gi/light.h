// include/gi/light.h -- mirrors include/light.h:10-58 of the reference: position, colour, radius, and the emission cone (dir, angle) that
// Octree::rebuild writes (include/octree.cpp:60-102).  getPoint(x, y) / getPointInRange(x, y) are the samplers the render path uses
// (the kernels' own functions); the reference's table of 250 sub-random points (getPoint() / getPoint(i)) is not on that path and is left out.
#pragma once
#include <vector>
#include "detail.h"
struct Light {
    std::vector<gi::dvec3> points;   // kept for source compatibility; empty
    gi::dvec3 dir;
    int count = 0;
    double angle = .125;
    Light(gi::dvec3 position, gi::dvec3 color, double radius) : pos(position), col(color), rad(radius) {}
    Light(gi::dvec3 position, gi::dvec3 target, gi::dvec3 color, double radius) : pos(position), col(color), rad(radius) { dir = gi::vnormalize(target - position); }
    gi::dvec3 getPoint(double x, double y) { return pos + rad * gi::from_v3(gi::random_unit_vec(x, y)); }
    gi::dvec3 getPointInRange(double x, double y)
    {
        if (angle < 1) return pos + rad * gi::from_v3(gi::sphere_cap_cos(gi::to_v3(dir), (float)x, (float)y, 1, angle));
        return pos + rad * gi::from_v3(gi::random_unit_vec(x, y));
    }
    gi::dvec3 pos = gi::dvec3(0, 0, 0);
    gi::dvec3 col = gi::dvec3(0, 0, 0);
    double rad = 0;
};
