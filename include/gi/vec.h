// include/gi/vec.h -- vector types of the drop-in C++ API.
// With -DGI_USE_GLM the reference's own glm::dvec3 / glm::dvec2 are used (so the reference's GUI, loaders and main.cpp compile
// unchanged against these headers); otherwise a minimal stand-alone pair with the same member names.
#pragma once
#ifdef GI_USE_GLM
#include <glm/glm.hpp>
namespace gi { typedef glm::dvec3 dvec3; typedef glm::dvec2 dvec2; }
#else
#include <cmath>
namespace gi {
struct dvec2 { double x, y; dvec2() : x(0), y(0) {} dvec2(double a, double b) : x(a), y(b) {} };
struct dvec3 {
    union { double x; double r; };
    union { double y; double g; };
    union { double z; double b; };
    dvec3() : x(0), y(0), z(0) {}
    dvec3(double a, double b_, double c) : x(a), y(b_), z(c) {}
    double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline dvec3 operator+(dvec3 a, dvec3 b) { return dvec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline dvec3 operator-(dvec3 a, dvec3 b) { return dvec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline dvec3 operator*(dvec3 a, double s) { return dvec3(a.x * s, a.y * s, a.z * s); }
inline dvec3 operator*(double s, dvec3 a) { return dvec3(s * a.x, s * a.y, s * a.z); }
inline double dot(dvec3 a, dvec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline dvec3 cross(dvec3 x, dvec3 y) { return dvec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
inline dvec3 normalize(dvec3 v) { return v * (1.0 / std::sqrt(dot(v, v))); }
}  // namespace gi
#endif
