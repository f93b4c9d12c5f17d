// include/gi/atmosphere.h -- mirrors include/atmosphere.h:11-83 of the reference: AtmosphereEntity (position, colour, box, scatter) and
// HeightFog.  The noise grid of a HeightFog is filled from the counter RNG when the entity is created (the reference fills it with its
// time-seeded drand(), include/atmosphere.h:41-44) and handed to the kernels as it is; density() evaluates the kernels' fog_density on it.
#pragma once
#include <vector>
#include "bbox.h"
struct AtmosphereEntity {
    AtmosphereEntity(gi::dvec3 position, gi::dvec3 size, gi::dvec3 color, double scatter) : pos(position), col(color), bbox(position - .5 * size, position + .5 * size), sc(scatter) {}
    virtual ~AtmosphereEntity() {}
    virtual double density(const gi::dvec3&) { return 0; }   // intersection probability per unit length
    gi::dvec3 pos = gi::dvec3(0, 0, 0);
    gi::dvec3 col = gi::dvec3(0, 0, 0);
    BoundingBox bbox;
    double sc = 0;
};
struct HeightFog : AtmosphereEntity {
    double d;
    int nscale;
    std::vector<double> noiseGrid;
    gi::dvec3 s;
    HeightFog(gi::dvec3 position, gi::dvec3 size, gi::dvec3 color, double density_, double scatter, int noiseScale, uint64_t seed = 0x9E3779B97F4A7C15ull)
        : AtmosphereEntity(position, size, color, scatter), d(density_), nscale(noiseScale), s(size)
    {
        const double q[12] = {position.x, position.y, position.z, size.x, size.y, size.z, color.x, color.y, color.z, density_, scatter, (double)noiseScale};
        noiseGrid.resize((size_t)gih_fog_grid(q, seed, nullptr, 0));
        gih_fog_grid(q, seed, noiseGrid.data(), (int32_t)noiseGrid.size());
        nscale = 1;   // as the reference's constructor leaves it (include/atmosphere.h:46)
    }
    double density(const gi::dvec3& p) override
    {
        gi::FogD f;
        memset(&f, 0, sizeof f);
        gi::put3(f.pos, pos); gi::put3(f.size, s); gi::put3(f.col, col);
        f.d = d; f.sc = sc;
        gi::put3(f.bmin, bbox.min); gi::put3(f.bmax, bbox.max);
        f.grid_off = 0; f.grid_n = (int32_t)noiseGrid.size();
        gi::Scene S{};
        S.fog_grid = noiseGrid.data();
        return gi::fog_density(S, f, gi::to_v3(p));
    }
};
