// include/gi/camera.h -- mirrors include/camera.h:7-31 of the reference (same members, same construction arithmetic).
#pragma once
#include "vec.h"
#define FOCAL_DIST 240
struct Camera {
    explicit Camera(gi::dvec3 pos) : Camera(pos, gi::dvec3(0, 0, 0)) {}
    Camera(gi::dvec3 pos_, gi::dvec3 lookAt) : pos(pos_) { setDir(lookAt - pos_); }
    void setDir(gi::dvec3 dir)
    {
        using namespace gi;
#ifdef GI_USE_GLM
        using namespace glm;
#endif
        forward = normalize(dir);
        up = gi::dvec3(0, 1.0, 0);
        right = normalize(cross(up, forward));
        up = cross(forward, right);
    }
    gi::dvec3 pos, up, forward, right;
    double sensorDiag = 0.035 * FOCAL_DIST * 2;
    double focalDist = 0.04 * FOCAL_DIST;
};
