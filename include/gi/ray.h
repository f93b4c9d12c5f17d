// include/gi/ray.h -- mirrors include/ray.h:5-37 of the reference: origin, normalised direction, its component-wise inverse.
#pragma once
#include "detail.h"
struct Ray {
    Ray(gi::dvec3 o, gi::dvec3 d) : origin(o), dir(d) { setDir(d); }
    void setDir(const gi::dvec3& d)
    {
        const gi::Ray r = gi::make_ray(gi::to_v3(origin), gi::to_v3(d));
        dir = gi::from_v3(r.d);
        invDir = gi::from_v3(r.inv);
    }
    gi::dvec3 origin, dir, invDir;
};
namespace gi {
inline Ray to_lane_ray(const ::Ray& r) { Ray q; q.o = to_v3(r.origin); q.d = to_v3(r.dir); q.inv = to_v3(r.invDir); return q; }
}
