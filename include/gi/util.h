// include/gi/util.h -- the constants of the reference's include/util.h:12-31 that its callers (loaders, GUI, main.cpp) may name, and gamma().
// The samplers, pow approximations and the RNG of that header are part of the render path and live in the kernels (gi_device.h).
#pragma once
#include <cmath>
#include "detail.h"
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define MAX_ENTITIES_PER_LEAF 16
#define MAX_PHOTONS_PER_LEAF 16
#define MIN_LEAF_SIZE .0015
#define MAX_SUBDIV_RATIO 0.75
#define EPSILON GI_EPSILON
#define SHADOW_BIAS GI_SHADOW_BIAS
#define MIN_DEPTH GI_MIN_DEPTH
#define MAX_DEPTH GI_MAX_DEPTH
#define NOISE_THRESH 0.0015
#define MIN_SAMPLES 8
#define SAMPLES 32
#define PHOTONS 75000
#define PHOTON_DEPTH 5
#define RAYMARCH_STEPSIZE GI_RAYMARCH_STEPSIZE
#define FOCAL_BLUR 0
#define GAMMA 2.2
inline double vecLengthSquared(const gi::dvec3& v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline double compMax(const gi::dvec3& v) { return std::fmax(std::fmax(v.x, v.y), v.z); }
inline gi::dvec3 gamma(const gi::dvec3& col, double g) { return gi::dvec3(std::pow(col.x, 1.0 / g), std::pow(col.y, 1.0 / g), std::pow(col.z, 1.0 / g)); }
inline double fastPow(double a, double b) { return gi::fast_pow(a, b); }
inline double fastPrecisePow(double a, double b) { return gi::fast_precise_pow(a, b); }
inline gi::dvec3 randomUnitVec(double x, double y) { return gi::from_v3(gi::random_unit_vec(x, y)); }
inline gi::dvec3 hemisphereSample_cos(gi::dvec3 n, float u, float v, double power) { return gi::from_v3(gi::hemi_cos_n(gi::to_v3(n), u, v, power)); }
inline gi::dvec3 sphereCapSample_cos(gi::dvec3 n, float u, float v, double power, double frac) { return gi::from_v3(gi::sphere_cap_cos(gi::to_v3(n), u, v, power, frac)); }
inline gi::dvec3 sample_phong(const gi::dvec3& outdir, const gi::dvec3&, double power, double sx, double sy) { return gi::from_v3(gi::sample_phong(gi::to_v3(outdir), power, sx, sy)); }
inline gi::dvec3 refr(gi::dvec3 inc, gi::dvec3 norm, double eta) { return gi::from_v3(gi::refr(gi::to_v3(inc), gi::to_v3(norm), eta)); }
