// include/gi/bbox.h -- mirrors include/bbox.h:11-139 of the reference: axis-aligned box with the closed box-box test, the half-open point test
// and the two slab tests; the slab arithmetic is the kernels' own (gi::box_range / gi::box_hit), compiled for the host.
#pragma once
#include <cassert>
#include "ray.h"
struct BoundingBox {
    BoundingBox(gi::dvec3 lo, gi::dvec3 hi) : min(lo), max(hi) { assert(lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z); }
    double dx() const { return max.x - min.x; }
    double dy() const { return max.y - min.y; }
    double dz() const { return max.z - min.z; }
    gi::dvec3 center() const { return min + 0.5 * (max - min); }
    gi::dvec3 size() const { return max - min; }
    bool intersect(const BoundingBox& o) const
    {
        const double a[6] = {min.x, min.y, min.z, max.x, max.y, max.z}, b[6] = {o.min.x, o.min.y, o.min.z, o.max.x, o.max.y, o.max.z};
        return gi::boxes_touch(a, a + 3, b, b + 3);
    }
    bool contains(gi::dvec3 p) const
    {
        const double a[6] = {min.x, min.y, min.z, max.x, max.y, max.z};
        return gi::box_contains(a, a + 3, gi::to_v3(p));
    }
    bool intersect(const Ray& ray, double tmin, double tmax, double& toutmin, double& toutmax) const
    {
        const double a[6] = {min.x, min.y, min.z, max.x, max.y, max.z};
        return gi::box_range(a, a + 3, gi::to_lane_ray(ray), tmin, tmax, toutmin, toutmax);
    }
    bool intersectSimple(const Ray& ray, double tmin, double tmax) const
    {
        const double a[6] = {min.x, min.y, min.z, max.x, max.y, max.z};
        return gi::box_hit(a, a + 3, gi::to_lane_ray(ray), tmin, tmax);
    }
    gi::dvec3 min, max;
};
