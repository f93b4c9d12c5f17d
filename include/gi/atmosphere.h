// include/gi/atmosphere.h -- mirrors include/atmosphere.h:11-83 of the reference: AtmosphereEntity / HeightFog as data holders.
// density() is evaluated inside the ray-march of the HIP kernels; the noise grid is filled when the Octree flattens the entity
// (from the counter RNG instead of the reference's time-seeded drand()).
#pragma once
#include "vec.h"
struct AtmosphereEntity {
    AtmosphereEntity(gi::dvec3 position, gi::dvec3 size_, gi::dvec3 color, double scatter) : pos(position), col(color), size(size_), sc(scatter) {}
    virtual ~AtmosphereEntity() {}
    gi::dvec3 pos, col, size;
    double sc = 0;
};
struct HeightFog : AtmosphereEntity {
    double d;
    int nscale;
    gi::dvec3 s;
    HeightFog(gi::dvec3 position, gi::dvec3 size_, gi::dvec3 color, double density, double scatter, int noiseScale)
        : AtmosphereEntity(position, size_, color, scatter), d(density), nscale(noiseScale), s(size_) {}
};
