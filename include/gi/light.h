// include/gi/light.h -- mirrors include/light.h:10-58 of the reference: position, colour, radius, and the emission cone (dir, angle) that
// Octree::rebuild writes (include/octree.cpp:60-102).  getPoint(x, y) / getPointInRange(x, y) are the samplers the render path uses
// (the kernels' own functions).  getPoint() / getPoint(i) pick from the table of 250 directions the reference's constructors fill (include/light.h:17-40,
// include/util.cpp:129-155: the Hammersley point (i / n, radical inverse of i) turned into a direction; the additive-recurrence state that function
// also sets up never reaches its output) -- not on the render path, here for callers that use them.  Like the reference they draw from rand().
#pragma once
#include <cstdlib>
#include <vector>
#include "detail.h"
struct Light {
    std::vector<gi::dvec3> points;   // 250 directions on the unit sphere
    gi::dvec3 dir;
    int count = 0;
    double angle = .125;
    Light(gi::dvec3 position, gi::dvec3 color, double radius) : pos(position), col(color), rad(radius) { fill_points(250); }
    Light(gi::dvec3 position, gi::dvec3 target, gi::dvec3 color, double radius) : pos(position), col(color), rad(radius) { fill_points(250); dir = gi::vnormalize(target - position); }
    gi::dvec3 getPoint() { return pos + rad * points[(size_t)rand() % points.size()]; }
    // (the reference's index, rand() + 1097 * i % size, runs past the table for most values of rand(): undefined there; the sum is reduced to the table here)
    gi::dvec3 getPoint(int i) { return pos + rad * points[((size_t)rand() + (size_t)(1097 * i) % points.size()) % points.size()]; }
    gi::dvec3 getPoint(double x, double y) { return pos + rad * gi::from_v3(gi::random_unit_vec(x, y)); }
    gi::dvec3 getPointInRange(double x, double y)
    {
        if (angle < 1) return pos + rad * gi::from_v3(gi::sphere_cap_cos(gi::to_v3(dir), (float)x, (float)y, 1, angle));
        return pos + rad * gi::from_v3(gi::random_unit_vec(x, y));
    }
    gi::dvec3 pos = gi::dvec3(0, 0, 0);
    gi::dvec3 col = gi::dvec3(0, 0, 0);
    double rad = 0;
private:
    void fill_points(int n)
    {
        points.clear();
        points.reserve((size_t)n);
        for (int i = 0; i < n; i++) {
            uint32_t b = (uint32_t)i, rev = 0;
            for (int k = 0; k < 32; k++) { rev = (rev << 1) | (b & 1u); b >>= 1; }          // van der Corput, base 2
            const double x = (double)((float)i / (float)n), y = (double)(float)((double)(float)rev * 2.3283064365386963e-10);
            const double theta = std::acos(2 * y - 1);
            points.push_back(gi::dvec3(std::sin(theta) * std::cos(2 * x * M_PI), std::sin(theta) * std::sin(2 * x * M_PI), std::cos(theta)));
        }
    }
};
