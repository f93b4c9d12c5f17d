// include/gi/photon.h -- mirrors include/photon.h:5-15 of the reference.
#pragma once
#include "vec.h"
struct Photon {
    Photon(gi::dvec3 o, gi::dvec3 d, gi::dvec3 c) : origin(o), dir(d), col(c) {}
    gi::dvec3 origin, dir, col;
};
