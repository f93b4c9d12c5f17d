// include/gi/detail.h -- what the C++ drop-in headers share: the vector types, conversions to the flat arrays of the C ABI, and a host
// build of the per-lane functions of the kernels (gi_raytracer_amd/csrc/gi_device.h is written so that the same text compiles for the
// device and for the host).  The host build serves the single-object conveniences of the reference's API (Entity::intersect,
// BoundingBox::intersect, Light::getPoint, Halton_sampler::sample ...): the arithmetic a caller sees there is the arithmetic the GPU runs.
// Rendering itself -- RayTracer::run, trace, visible, samplePhotons, radiance, tracePhotons -- never comes this way: it goes through the
// C ABI (include/gi_hip.h) to the GPU and fails loudly without one.
#pragma once
#include <cmath>
#include <cstring>
#include "vec.h"
#if !defined(__HIPCC__) && !defined(GI_HD)
#define GI_HD static inline
#define GI_HDM inline
#endif
#include "../../gi_raytracer_amd/csrc/gi_device.h"
#include "../../gi_raytracer_amd/csrc/gi_host.h"

namespace gi {
inline V3 to_v3(const dvec3& v) { return v3(v.x, v.y, v.z); }
inline dvec3 from_v3(const V3& v) { return dvec3(v.x, v.y, v.z); }
inline void put3(double* dst, const dvec3& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
inline dvec3 get3(const double* p) { return dvec3(p[0], p[1], p[2]); }
inline dvec3 vnormalize(const dvec3& v) { return from_v3(normalize(to_v3(v))); }   // glm::normalize = v * inversesqrt(dot(v, v))
}  // namespace gi
