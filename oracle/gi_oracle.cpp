// oracle/gi_oracle.cpp -- CPU oracle: double-precision restatement of the reference render hot path.
//
// TEST INFRASTRUCTURE ONLY (see gi_oracle.h).  Every function cites the reference file:line it follows
// (paths relative to the reference repository moepforfreedom/GI_Raytracer).  Arithmetic is written in the
// reference's operand order (glm 0.9.8 semantics: normalize = v * (1/sqrt(dot)), dot = (x+y)+z, reflect =
// I - N*dot(N,I)*2, mix = x + a*(y-x)) so that results are bit-identical to the compiled reference wherever
// the reference itself is deterministic.  Build with -ffp-contract=off.
//
// Parity pin: tests/test_oracle_vs_reference.py checks this file against tests/golden/*.npz, which were
// captured from the unmodified reference by oracle/ref/ref_driver.cpp (function tables: Halton, samplers,
// octree, trace, visible, photon octree, gather; whole frames: the reference's own RayTracer::run on a pinned
// xorshift chain, reproduced here by GIO_RNG_CHAIN).
#include "gi_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

// ----------------------------------------------------------------------------- constants (include/util.h:14-31)
const int MAX_ENTITIES_PER_LEAF = 16;
const int MAX_PHOTONS_PER_LEAF = 16;
const double MIN_LEAF_SIZE = .0015;
const double MAX_SUBDIV_RATIO = 0.75;
const double EPSILON = 0.00001;
const double SHADOW_BIAS = 0.0001;
const int MIN_DEPTH = 2;
const int MAX_DEPTH = 64;
const double RAYMARCH_STEPSIZE = 0.04;
const double GAMMA_ = 2.2;                  // include/util.h:31
const double PI = 3.14159265358979323846;  // glibc M_PI (util.h's fallback #define is not taken)

// ----------------------------------------------------------------------------- vec3 with glm operand order
struct V3 { double x, y, z; };
struct V2 { double x, y; };
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 operator+(V3 a, double s) { return V3{a.x + s, a.y + s, a.z + s}; }
inline V3 operator-(V3 a, double s) { return V3{a.x - s, a.y - s, a.z - s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 x, V3 y) { return V3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline V3 normalize(V3 v) { return v * (1.0 / std::sqrt(dot(v, v))); }
inline double length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0; }
inline V3 mix(V3 x, V3 y, double a) { return x + a * (y - x); }
inline double len2(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }  // vecLengthSquared, util.h:35
inline double comp_max(V3 v) { return std::max(std::max(v.x, v.y), v.z); }  // util.h:47
inline double idx3(const V3& v, int i) { return (&v.x)[i]; }

// ----------------------------------------------------------------------------- RNG
// splitmix64 finaliser
inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27; z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
inline double counter_rand(uint64_t seed, uint32_t stream, uint32_t depth, uint32_t purpose, uint32_t a, uint32_t b)
{
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)stream + 1));
    h = mix64(h ^ (((uint64_t)depth << 32) | purpose));
    h = mix64(h ^ (((uint64_t)a << 32) | b));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
// purposes of the counter RNG contract (low byte; light index in bits 8..)
enum {
    P_FOG_CAMERA = 0, P_FOG_SHADOW = 1, P_FOG_PHOTON = 2,   // `b` key of P_FOG draws: which march of the vertex (shadow: + light index)
};
enum {
    P_TRACE_ALPHA = 0, P_SHADOW_ALPHA = 1, P_LIGHT_X = 2, P_LIGHT_Y = 3, P_TYPE_OPACITY = 4, P_TYPE_FRESNEL = 5,
    P_RR = 6, P_FOG = 7, P_TRACE_GUARD = 8,
    P_PH_DIR_U = 16, P_PH_DIR_V = 17, P_PH_SEC_U = 18, P_PH_SEC_V = 19, P_PH_TRACE0_ALPHA = 20, P_PH_FOG_U = 21, P_PH_FOG_V = 22
};
const uint64_t PHOTON_SEED_XOR = 0x5048544f4e5eed00ull;

struct Rng {
    int mode = GIO_RNG_COUNTER;
    uint64_t seed = 0;
    uint64_t state = 0;  // chain
    uint32_t stream = 0, depth = 0;
    // xorshift64*, include/util.h:52-74
    double chain_next()
    {
        uint64_t& x = state;
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        return (double)(x * 2685821657736338717ull) / (double)18446744073709551615ull;
    }
    double draw(uint32_t purpose, uint32_t a = 0, uint32_t b = 0)
    {
        if (mode == GIO_RNG_CHAIN) return chain_next();
        return counter_rand(seed, stream, depth, purpose, a, b);
    }
};

// ----------------------------------------------------------------------------- util.h / util.cpp
// include/util.h:100-111
inline double fast_pow(double a, double b)
{
    union { double d; int32_t x[2]; } u = {a};
    u.x[1] = (int32_t)(b * (u.x[1] - 1072632447) + 1072632447);
    u.x[0] = 0;
    return u.d;
}
// include/util.h:113-136
inline double fast_precise_pow(double a, double b)
{
    int e = (int)b;
    union { double d; int32_t x[2]; } u = {a};
    u.x[1] = (int32_t)((b - e) * (u.x[1] - 1072632447) + 1072632447);
    u.x[0] = 0;
    double r = 1.0;
    while (e) {
        if (e & 1) r *= a;
        a *= a;
        e >>= 1;
    }
    return r * u.d;
}
// local frame used by the samplers, include/util.cpp:37-42 (glm column-major dmat3x3 times vector)
inline V3 frame_mul(V3 n, double z, V3 r)
{
    double k = (1.0 / (1 + z));
    V3 c0 = v3(z + k * -n.y * -n.y, k * (n.x * -n.y), -n.x);
    V3 c1 = v3(k * (n.x * -n.y), z + k * -n.x * -n.x, -n.y);
    V3 c2 = v3(n.x, n.y, z);
    return v3(c0.x * r.x + c1.x * r.y + c2.x * r.z, c0.y * r.x + c1.y * r.y + c2.y * r.z, c0.z * r.x + c1.z * r.y + c2.z * r.z);
}
// include/util.cpp:27-33
inline V3 hemi_cos(float u, float v, double power)
{
    float phi = v * 2.0f * PI;
    float cosTheta = fast_precise_pow(1.0f - u, (1.0f / power));
    float sinTheta = std::sqrt((double)(1.0f - cosTheta * cosTheta));
    return v3(std::cos((double)phi) * sinTheta, std::sin((double)phi) * sinTheta, cosTheta);
}
// include/util.cpp:35-58
inline V3 hemi_cos_n(V3 normal, float u, float v, double power)
{
    double z = std::abs(normal.z);
    float phi = v * 2.0f * PI;
    float cosTheta = fast_precise_pow(1.0f - u, (1.0f / power));
    float sinTheta = std::sqrt((double)(1.0f - cosTheta * cosTheta));
    V3 res = v3(std::cos((double)phi) * sinTheta, std::sin((double)phi) * sinTheta, cosTheta);
    res = frame_mul(normal, z, res);
    if (normal.z < 0) res.z *= -1.0;
    return res;
}
// include/util.cpp:60-83
inline V3 sphere_cap_cos(V3 normal, float u, float v, double power, double frac)
{
    double z = std::abs(normal.z);
    float phi = v * 2.0f * PI;
    float cosTheta = frac * fast_precise_pow(1.0f - u, (1.0f / power)) + (1 - frac);
    float sinTheta = std::sqrt((double)(1.0f - cosTheta * cosTheta));
    V3 res = v3(std::cos((double)phi) * sinTheta, std::sin((double)phi) * sinTheta, cosTheta);
    res = frame_mul(normal, z, res);
    if (normal.z < 0) res.z *= -1.0;
    return res;
}
// include/util.cpp:91-107
inline V3 sample_phong(V3 outdir, V3 /*n*/, double power, double sx, double sy)
{
    double z = std::abs(outdir.z);
    V3 out = frame_mul(outdir, z, hemi_cos((float)sx, (float)sy, power));
    if (outdir.z < 0) out.z *= -1.0;
    return out;
}
// include/util.h:183-188
inline V3 random_unit_vec(double x, double y)
{
    double theta = std::acos(2 * y - 1);
    return v3(std::sin(theta) * std::cos(2 * x * PI), std::sin(theta) * std::sin(2 * x * PI), std::cos(theta));
}
// include/util.h:173-181
inline V3 refr(V3 inc, V3 norm, double eta)
{
    double d = dot(norm, inc);
    double k = 1.0 - eta * eta * (1.0 - d * d);
    if (k < EPSILON) return reflect(inc, norm);
    return eta * inc - (eta * d + std::sqrt(k)) * norm;
}

// include/util.cpp:186-207 planeBoxOverlap (float d, double vectors)
inline int plane_box_overlap(V3 normal, float d, V3 maxbox)
{
    V3 vmin, vmax;
    for (int q = 0; q <= 2; q++) {
        if (idx3(normal, q) > 0.0f) { (&vmin.x)[q] = -idx3(maxbox, q); (&vmax.x)[q] = idx3(maxbox, q); }
        else { (&vmin.x)[q] = idx3(maxbox, q); (&vmax.x)[q] = -idx3(maxbox, q); }
    }
    if (dot(normal, vmin) + d > 0.0f) return 0;
    if (dot(normal, vmax) + d >= 0.0f) return 1;
    return 0;
}
// include/util.cpp:257-330 triBoxOverlap (Akenine-Moller): note the float temporaries min,max,d,p0,p1,p2,rad,fe*
inline bool tri_box_overlap(V3 boxcenter, V3 bh, const V3 tv[3])
{
    V3 v0 = tv[0] - boxcenter, v1 = tv[1] - boxcenter, v2 = tv[2] - boxcenter;
    V3 e0 = v1 - v0, e1 = v2 - v1, e2 = v0 - v2;
    float mn, mx, d, p0, p1, p2, rad, fex, fey, fez;
#define AX_X01(a, b, fa, fb) p0 = a * v0.y - b * v0.z; p2 = a * v2.y - b * v2.z; if (p0 < p2) { mn = p0; mx = p2; } else { mn = p2; mx = p0; } rad = fa * bh.y + fb * bh.z; if (mn > rad || mx < -rad) return false;
#define AX_X2(a, b, fa, fb) p0 = a * v0.y - b * v0.z; p1 = a * v1.y - b * v1.z; if (p0 < p1) { mn = p0; mx = p1; } else { mn = p1; mx = p0; } rad = fa * bh.y + fb * bh.z; if (mn > rad || mx < -rad) return false;
#define AX_Y02(a, b, fa, fb) p0 = -a * v0.x + b * v0.z; p2 = -a * v2.x + b * v2.z; if (p0 < p2) { mn = p0; mx = p2; } else { mn = p2; mx = p0; } rad = fa * bh.x + fb * bh.z; if (mn > rad || mx < -rad) return false;
#define AX_Y1(a, b, fa, fb) p0 = -a * v0.x + b * v0.z; p1 = -a * v1.x + b * v1.z; if (p0 < p1) { mn = p0; mx = p1; } else { mn = p1; mx = p0; } rad = fa * bh.x + fb * bh.z; if (mn > rad || mx < -rad) return false;
#define AX_Z12(a, b, fa, fb) p1 = a * v1.x - b * v1.y; p2 = a * v2.x - b * v2.y; if (p2 < p1) { mn = p2; mx = p1; } else { mn = p1; mx = p2; } rad = fa * bh.x + fb * bh.y; if (mn > rad || mx < -rad) return false;
#define AX_Z0(a, b, fa, fb) p0 = a * v0.x - b * v0.y; p1 = a * v1.x - b * v1.y; if (p0 < p1) { mn = p0; mx = p1; } else { mn = p1; mx = p0; } rad = fa * bh.x + fb * bh.y; if (mn > rad || mx < -rad) return false;
    fex = std::abs(e0.x); fey = std::abs(e0.y); fez = std::abs(e0.z);
    AX_X01(e0.z, e0.y, fez, fey); AX_Y02(e0.z, e0.x, fez, fex); AX_Z12(e0.y, e0.x, fey, fex);
    fex = std::abs(e1.x); fey = std::abs(e1.y); fez = std::abs(e1.z);
    AX_X01(e1.z, e1.y, fez, fey); AX_Y02(e1.z, e1.x, fez, fex); AX_Z0(e1.y, e1.x, fey, fex);
    fex = std::abs(e2.x); fey = std::abs(e2.y); fez = std::abs(e2.z);
    AX_X2(e2.z, e2.y, fez, fey); AX_Y1(e2.z, e2.x, fez, fex); AX_Z12(e2.y, e2.x, fey, fex);
#undef AX_X01
#undef AX_X2
#undef AX_Y02
#undef AX_Y1
#undef AX_Z12
#undef AX_Z0
#define FMM(x0, x1, x2) mn = mx = x0; if (x1 < mn) mn = x1; if (x1 > mx) mx = x1; if (x2 < mn) mn = x2; if (x2 > mx) mx = x2;
    FMM(v0.x, v1.x, v2.x); if (mn > bh.x || mx < -bh.x) return false;
    FMM(v0.y, v1.y, v2.y); if (mn > bh.y || mx < -bh.y) return false;
    FMM(v0.z, v1.z, v2.z); if (mn > bh.z || mx < -bh.z) return false;
#undef FMM
    V3 normal = cross(e0, e1);
    d = -dot(normal, v0);
    if (!plane_box_overlap(normal, d, bh)) return false;
    return true;
}

// ----------------------------------------------------------------------------- Halton (halton_enum.h, halton_sampler.h)
struct HaltonEnum {
    unsigned p2 = 0, p3 = 0, m_x = 0, m_y = 0, inc = 1;
    float scale_x = 1, scale_y = 1;
    static void ext_euclid(int a, int b, int& s, int& t)  // halton_enum.h:126-134
    {
        if (!b) { s = 1; t = 0; return; }
        int q = a / b, r = a % b, s1, t1;
        ext_euclid(b, r, s1, t1);
        s = t1; t = s1 - q * t1;
    }
    void init(unsigned width, unsigned height)  // halton_enum.h:69-104
    {
        p2 = 0; unsigned w = 1;
        while (w < width) { ++p2; w *= 2; }
        scale_x = float(w);
        p3 = 0; unsigned h = 1;
        while (h < height) { ++p3; h *= 3; }
        scale_y = float(h);
        inc = w * h;
        int s, t;
        ext_euclid((int)h, (int)w, s, t);
        unsigned inv2 = (s < 0) ? (s + w) : (s % w);
        unsigned inv3 = (t < 0) ? (t + h) : (t % h);
        m_x = h * inv2;
        m_y = w * inv3;
    }
    static unsigned inv2(unsigned index, unsigned digits)  // halton_enum.h:136-144
    {
        index = (index << 16) | (index >> 16);
        index = ((index & 0x00ff00ff) << 8) | ((index & 0xff00ff00) >> 8);
        index = ((index & 0x0f0f0f0f) << 4) | ((index & 0xf0f0f0f0) >> 4);
        index = ((index & 0x33333333) << 2) | ((index & 0xcccccccc) >> 2);
        index = ((index & 0x55555555) << 1) | ((index & 0xaaaaaaaa) >> 1);
        return digits ? index >> (32 - digits) : 0;   // digits == 0 only for width 1: x is 0 anyway
    }
    static unsigned inv3(unsigned index, unsigned digits)  // halton_enum.h:146-155
    {
        unsigned result = 0;
        for (unsigned d = 0; d < digits; ++d) { result = result * 3 + index % 3; index /= 3; }
        return result;
    }
    unsigned get_index(unsigned i, unsigned x, unsigned y) const  // halton_enum.h:106-114
    {
        const unsigned long long hx = inv2(x, p2);
        const unsigned long long hy = inv3(y, p3);
        const unsigned offset = (unsigned)((hx * m_x + hy * m_y) % inc);
        return offset + i * inc;
    }
};

// Faure-permuted Halton sampler, 256 dimensions (halton_sampler.h).  The reference file is generated code with one
// function per base; its rule is: per base b, a table over g digits (largest b^g <= 500) of the digit-permuted radical
// inverse, n groups with (b^g)^n < 2^32, value = (sum_k table[(index / P^k) % P] * P^(n-1-k)) * float(0.9999998807907104 / P^n).
struct HaltonSampler {
    struct Dim { unsigned base, P, n; float scale; unsigned off; };
    Dim dims[256];
    std::vector<uint16_t> table;
    bool ready = false;
    void init_faure()  // halton_sampler.h:573-603 (perms), :890-900 (invert), :902-1414 (tables)
    {
        const unsigned max_base = 1619u;
        std::vector<std::vector<uint16_t>> perms(max_base + 1);
        for (unsigned k = 1; k <= 3; ++k) { perms[k].resize(k); for (unsigned i = 0; i < k; ++i) perms[k][i] = i; }
        for (unsigned base = 4; base <= max_base; ++base) {
            perms[base].resize(base);
            const unsigned b = base / 2;
            if (base & 1) {
                for (unsigned i = 0; i < base - 1; ++i) perms[base][i + (i >= b)] = perms[base - 1][i] + (perms[base - 1][i] >= b);
                perms[base][b] = b;
            } else {
                for (unsigned i = 0; i < b; ++i) { perms[base][i] = 2 * perms[b][i]; perms[base][b + i] = 2 * perms[b][i] + 1; }
            }
        }
        // first 256 primes
        std::vector<unsigned> primes;
        for (unsigned c = 2; primes.size() < 256; c++) {
            bool ok = true;
            for (unsigned p : primes) { if (p * p > c) break; if (c % p == 0) { ok = false; break; } }
            if (ok) primes.push_back(c);
        }
        table.clear();
        for (unsigned d = 0; d < 256; d++) {
            unsigned b = primes[d];
            unsigned g = 1, P = b;
            while ((unsigned long long)P * b <= 500ull) { P *= b; g++; }
            unsigned n = 1; unsigned long long Q = P;
            while (Q * P < 4294967296ull) { Q *= P; n++; }
            dims[d].base = b; dims[d].P = P; dims[d].n = n;
            dims[d].scale = float(0.9999998807907104 / (double)Q);
            dims[d].off = (unsigned)table.size();
            for (unsigned i = 0; i < P; i++) {
                unsigned short result = 0, index = (unsigned short)i;
                for (unsigned k = 0; k < g; ++k) { result = result * b + perms[b][index % b]; index /= b; }
                table.push_back(result);
            }
        }
        ready = true;
    }
    float sample(unsigned dimension, unsigned index) const  // halton_sampler.h:626-888, :1417-3286
    {
        if (dimension == 0) {  // :1417-1431 bit reversal into the mantissa
            index = (index << 16) | (index >> 16);
            index = ((index & 0x00ff00ff) << 8) | ((index & 0xff00ff00) >> 8);
            index = ((index & 0x0f0f0f0f) << 4) | ((index & 0xf0f0f0f0) >> 4);
            index = ((index & 0x33333333) << 2) | ((index & 0xcccccccc) >> 2);
            index = ((index & 0x55555555) << 1) | ((index & 0xaaaaaaaa) >> 1);
            union { unsigned u; float f; } r;
            r.u = 0x3f800000u | (index >> 9);
            return r.f - 1.f;
        }
        const Dim& D = dims[dimension];
        unsigned sum = 0, idx = index;
        unsigned w = 1;
        for (unsigned k = 1; k < D.n; k++) w *= D.P;  // P^(n-1)
        for (unsigned k = 0; k < D.n; k++) {
            sum += table[D.off + idx % D.P] * w;
            idx /= D.P;
            w /= D.P;
        }
        return sum * D.scale;
    }
};
HaltonSampler g_sampler;
const HaltonSampler& sampler()
{
    if (!g_sampler.ready) {
#pragma omp critical(gio_sampler_init)
        if (!g_sampler.ready) g_sampler.init_faure();
    }
    return g_sampler;
}

// ----------------------------------------------------------------------------- geometry
struct Ray {
    V3 origin, dir, invDir;
    Ray() {}
    Ray(V3 o, V3 d) : origin(o) { set_dir(d); }
    void set_dir(V3 d) { dir = normalize(d); invDir = v3(1.0 / dir.x, 1.0 / dir.y, 1.0 / dir.z); }  // include/ray.h:12-17
    void set_dir_exact(V3 d) { dir = d; invDir = v3(1.0 / dir.x, 1.0 / dir.y, 1.0 / dir.z); }
};

struct Box {
    V3 min, max;
    double dx() const { return max.x - min.x; }
    double dy() const { return max.y - min.y; }
    double dz() const { return max.z - min.z; }
    V3 center() const { return min + 0.5 * (max - min); }
    // include/bbox.h:33-38
    bool intersect(const Box& o) const
    {
        return (min.x <= o.max.x && max.x >= o.min.x) && (min.y <= o.max.y && max.y >= o.min.y) && (min.z <= o.max.z && max.z >= o.min.z);
    }
    // include/bbox.h:41-44
    bool contains(V3 p) const { return p.x >= min.x && p.y >= min.y && p.z >= min.z && p.x < max.x && p.y < max.y && p.z < max.z; }
    // include/bbox.h:47-73
    bool intersect(const Ray& ray, double tmin, double tmax, double& toutmin, double& toutmax) const
    {
        for (int i = 0; i < 3; i++) {
            double t0 = (idx3(min, i) - idx3(ray.origin, i)) * idx3(ray.invDir, i);
            double t1 = (idx3(max, i) - idx3(ray.origin, i)) * idx3(ray.invDir, i);
            if (idx3(ray.invDir, i) < 0.0) { double tmp = t0; t0 = t1; t1 = tmp; }
            tmin = t0 > tmin ? t0 : tmin;
            tmax = t1 < tmax ? t1 : tmax;
            if (tmax <= tmin) { toutmin = INFINITY; toutmax = -INFINITY; return false; }
        }
        toutmin = tmin; toutmax = tmax;
        return true;
    }
    // include/bbox.h:117-138
    bool intersect_simple(const Ray& ray, double tmin, double tmax) const
    {
        for (int i = 0; i < 3; i++) {
            double t0 = (idx3(min, i) - idx3(ray.origin, i)) * idx3(ray.invDir, i);
            double t1 = (idx3(max, i) - idx3(ray.origin, i)) * idx3(ray.invDir, i);
            if (idx3(ray.invDir, i) < 0.0) { double tmp = t0; t0 = t1; t1 = tmp; }
            tmin = t0 > tmin ? t0 : tmin;
            tmax = t1 < tmax ? t1 : tmax;
            if (tmax <= tmin) return false;
        }
        return true;
    }
};

struct Material { double roughness, opacity, IOR; V3 diffuse, emissive; int dtex = -1, etex = -1; };  // include/material.h:84-100; diffuse / emissive = texture::color
struct Texture { int kind; double p[8]; };   // 0 texture(col), 1 checkerboard, 2 imageTexture (include/material.h:10-81); parameters as gi_scene_desc::tex_param
struct Entity {
    int kind;       // 0 triangle, 1 sphere
    V3 p[3], n[3];  // triangle vertices / normals; sphere: p[0] centre, p[1].x radius
    V2 t[3];
    V3 fnorm;       // include/entities.h:339
    int mat;
    bool smooth;    // all three vertex normals non-zero, include/entities.h:478
};
struct Light { V3 pos, col; double rad; V3 dir; double angle; };
struct Fog { V3 pos, size, col; double d, sc; int nscale; std::vector<double> grid; };  // HeightFog, include/atmosphere.h:30-83

struct Hit { V3 pos, norm; V2 uv; };

// include/entities.h:443-490 (triangle), :60-101 (sphere)
inline bool ent_intersect(const Entity& e, const Ray& ray, Hit& h)
{
    if (e.kind == 0) {
        V3 edge1 = e.p[1] - e.p[0];
        V3 edge2 = e.p[2] - e.p[0];
        V3 p = cross(ray.dir, edge2);
        double det = dot(edge1, p);
        if (det < EPSILON && det > -EPSILON) return false;
        double inv_det = 1.0 / det;
        V3 tvec = ray.origin - e.p[0];
        double u = dot(tvec, p) * inv_det;
        if (u < 0 || u > 1) return false;
        V3 q = cross(tvec, edge1);
        double v = dot(ray.dir, q) * inv_det;
        if (v < 0 || u + v > 1) return false;
        double t = dot(edge2, q) * inv_det;
        if (t <= 0) return false;
        h.pos = ray.origin + t * ray.dir;
        if (e.smooth) {
            h.norm = (1 - u - v) * e.n[0] + u * e.n[1] + v * e.n[2];
            V2 a = e.t[0], b = e.t[1], c = e.t[2];
            double w = (1 - u - v);
            h.uv = V2{w * a.x + u * b.x + v * c.x, w * a.y + u * b.y + v * c.y};
        } else
            h.norm = e.fnorm;  // uv left untouched by the reference
        return true;
    }
    // sphere
    V3 pos = e.p[0];
    double rad = e.p[1].x;
    double d = dot(ray.dir, (ray.origin - pos));
    double r = (std::pow(d, 2) - len2(ray.origin - pos) + std::pow(rad, 2));
    if (r < 0) return false;
    double sr = std::sqrt(r);
    double t_1 = -1 * d - sr;
    double t_2 = -1 * d + sr;
    if (t_1 < 0 && t_2 < 0) return false;
    if ((t_1 < t_2 && t_1 > 0) || t_2 < 0) h.pos = ray.origin + ray.dir * t_1;
    else h.pos = ray.origin + ray.dir * t_2;
    h.norm = normalize(h.pos - pos);
    V3 dd = (pos - h.pos) / rad;
    double vv = .5 + std::asin(dd.y) / PI;
    double uu = .5 + std::atan2(dd.z, dd.x) / (2 * PI);
    h.uv = V2{uu, vv};
    return true;
}
// include/entities.h:530-557 (triangle: note max += EPSILON after EVERY vertex), :103-106 (sphere)
inline Box ent_bbox(const Entity& e)
{
    if (e.kind == 0) {
        V3 mn = v3(INFINITY, INFINITY, INFINITY), mx = v3(-INFINITY, -INFINITY, -INFINITY);
        for (int i = 0; i < 3; i++) {
            V3 vp = e.p[i];
            if (vp.x < mn.x) mn.x = vp.x;
            if (vp.x > mx.x) mx.x = vp.x;
            if (vp.y < mn.y) mn.y = vp.y;
            if (vp.y > mx.y) mx.y = vp.y;
            if (vp.z < mn.z) mn.z = vp.z;
            if (vp.z > mx.z) mx.z = vp.z;
            mx.x += EPSILON; mx.y += EPSILON; mx.z += EPSILON;
        }
        return Box{mn, mx};
    }
    double rad = e.p[1].x;
    return Box{e.p[0] + rad * v3(-1, -1, -1), e.p[0] + rad * v3(1, 1, 1)};
}
// include/entities.h:522-528 (triangle), :108-141 (sphere)
inline bool ent_overlaps_box(const Entity& e, const Box& b)
{
    if (e.kind == 0) {
        Box tb{b.min - EPSILON, b.max + EPSILON};
        V3 verts[3] = {e.p[0], e.p[1], e.p[2]};
        return tri_box_overlap(tb.center(), v3(tb.dx() / 2, tb.dy() / 2, tb.dz() / 2), verts);
    }
    auto check = [](double pn, double bmin, double bmax) {
        double out = 0, v = pn;
        if (v < bmin) { double val = (bmin - v); out += val * val; }
        if (v > bmax) { double val = (v - bmax); out += val * val; }
        return out;
    };
    double sq = 0.0;
    sq += check(e.p[0].x, b.min.x, b.max.x);
    sq += check(e.p[0].y, b.min.y, b.max.y);
    sq += check(e.p[0].z, b.min.z, b.max.z);
    double rad = e.p[1].x;
    return sq <= (rad * rad);
}

// ----------------------------------------------------------------------------- octrees (pre-order node arrays)
struct ONode { Box box; int child[8]; std::vector<int> ents; bool leaf() const { for (int i = 0; i < 8; i++) if (child[i] >= 0) return false; return true; } };
struct PNode { Box box; int first_child; std::vector<int> ph; };
struct Photon { V3 origin, dir, col; };

struct Counters { int64_t v_trace = 0, v_shadow = 0, tri = 0, shaded = 0, pcand = 0, traces = 0, shadows = 0, gathers = 0, tri_shadow = 0; };

}  // namespace

struct gio_ctx {
    std::vector<Entity> ents;
    std::vector<Material> mats;
    std::vector<Texture> texs;
    std::vector<unsigned char> tex_pixels;
    std::vector<Light> lights;
    std::vector<Fog> fogs;
    V3 ambient = {0, 0, 0};
    V3 cam_pos = {10, 5, 0}, cam_up = {0, 1, 0}, cam_fwd = {-1, 0, 0};
    double sensorDiag = 0.035 * 240 * 2, focalDist = 0.04 * 240;
    std::vector<ONode> onodes;
    bool octree_valid = false;
    std::vector<Photon> photons;
    std::vector<PNode> pnodes;
    bool pmap_valid = false;
    Box pmap_root_box;
    uint64_t chain_state = 0;  // GIO_RNG_CHAIN: one xorshift64* chain across emit_photons and render calls

    // ---- octree build: Octree::Node::partition, include/octree.cpp:316-384
    void partition(int ni)
    {
        Box bb = onodes[ni].box;
        V3 mid = mix(bb.min, bb.max, .5);
        Box cb[8];
        cb[0] = Box{bb.min, mid};
        cb[1] = Box{v3(bb.min.x + .5 * bb.dx(), bb.min.y, bb.min.z), v3(mid.x + .5 * bb.dx(), mid.y, mid.z)};
        cb[2] = Box{v3(bb.min.x, bb.min.y, bb.min.z + .5 * bb.dz()), v3(mid.x, mid.y, mid.z + .5 * bb.dz())};
        cb[3] = Box{v3(bb.min.x + .5 * bb.dx(), bb.min.y, bb.min.z + .5 * bb.dz()), v3(mid.x + .5 * bb.dx(), mid.y, mid.z + .5 * bb.dz())};
        cb[4] = Box{v3(bb.min.x, bb.min.y + .5 * bb.dy(), bb.min.z), v3(mid.x, mid.y + .5 * bb.dy(), mid.z)};
        cb[5] = Box{v3(bb.min.x + .5 * bb.dx(), bb.min.y + .5 * bb.dy(), bb.min.z), v3(mid.x + .5 * bb.dx(), mid.y + .5 * bb.dy(), mid.z)};
        cb[6] = Box{v3(bb.min.x, bb.min.y + .5 * bb.dy(), bb.min.z + .5 * bb.dz()), v3(mid.x, mid.y + .5 * bb.dy(), mid.z + .5 * bb.dz())};
        cb[7] = Box{mid, bb.max};
        std::vector<int> lists[8];
        for (int e : onodes[ni].ents) {
            Box eb = ent_bbox(ents[e]);
            for (int i = 0; i < 8; i++)
                if (cb[i].intersect(eb) && ent_overlaps_box(ents[e], cb[i]) && eb.dx() > EPSILON) lists[i].push_back(e);
        }
        double avg = 0;
        for (int i = 0; i < 8; i++) if (!lists[i].empty()) avg += (double)lists[i].size();
        avg /= 8;
        size_t parent_count = onodes[ni].ents.size();
        onodes[ni].ents.clear();
        bool stop = avg > MAX_SUBDIV_RATIO * parent_count;
        // children are appended in pre-order: child i and its whole sub-tree before child i+1
        for (int i = 0; i < 8; i++) {
            if (lists[i].empty()) continue;
            int ci = (int)onodes.size();
            onodes.push_back(ONode());
            onodes[ci].box = cb[i];
            for (int k = 0; k < 8; k++) onodes[ci].child[k] = -1;
            onodes[ci].ents = lists[i];
            onodes[ni].child[i] = ci;
            if (!stop && (int)lists[i].size() > MAX_ENTITIES_PER_LEAF && cb[i].dx() > MIN_LEAF_SIZE) partition(ci);
        }
    }

    // Octree::push_back + Octree::rebuild, include/octree.cpp:25-38, :53-119
    void build_octree()
    {
        onodes.clear();
        onodes.push_back(ONode());
        ONode& root = onodes[0];
        for (int k = 0; k < 8; k++) root.child[k] = -1;
        root.box = Box{v3(0, 0, 0), v3(0, 0, 0)};
        for (size_t i = 0; i < ents.size(); i++) {
            Box b = ent_bbox(ents[i]);
            if (root.ents.empty()) { root.box.max = b.max; root.box.min = b.min; }
            root.ents.push_back((int)i);
            root.box.max = v3(std::max(root.box.max.x, b.max.x), std::max(root.box.max.y, b.max.y), std::max(root.box.max.z, b.max.z));
            root.box.min = v3(std::min(root.box.min.x, b.min.x), std::min(root.box.min.y, b.min.y), std::min(root.box.min.z, b.min.z));
        }
        pmap_root_box = root.box;  // RayTracer::setScene, include/raytracer.h:38
        // light cone precompute, include/octree.cpp:60-102
        V3 avgPos = v3(0, 0, 0);
        double count = 0;
        for (const Entity& e : ents)
            if (mats[e.mat].roughness < 0.1) { avgPos = avgPos + ent_bbox(e).center(); count++; }
        if (count > 0) avgPos = avgPos / count;
        for (Light& l : lights) {
            double maxAngle = 0;
            l.dir = normalize(avgPos - l.pos);
            for (const Entity& e : ents)
                if (mats[e.mat].roughness < 0.1) {
                    Box b = ent_bbox(e);
                    avgPos = avgPos + b.center();
                    double angle = 1.0 - std::acos(dot(l.dir, normalize(l.pos - b.min))) / PI;
                    maxAngle = std::max(maxAngle, angle);
                    count++;
                }
            l.angle = maxAngle;
        }
        if ((int)onodes[0].ents.size() > MAX_ENTITIES_PER_LEAF) partition(0);
        octree_valid = true;
    }

    // Octree::Node::intersectSorted, include/octree.cpp:285-313
    void intersect_sorted(int ni, const Ray& ray, std::vector<std::pair<int, double>>& res, double tmin, double tmax, Counters* c) const
    {
        const ONode& n = onodes[ni];
        double t0, t1;
        if (c) c->v_trace++;
        if (n.box.intersect(ray, tmin, tmax, t0, t1)) {
            if (n.leaf()) {
                if (n.ents.size() > 0) {
                    auto it = std::partition_point(res.begin(), res.end(), [&t0](const std::pair<int, double>& q) { return t0 >= q.second; });
                    res.insert(it, {ni, t0});
                }
            } else
                for (int i = 0; i < 8; i++)
                    if (n.child[i] >= 0) intersect_sorted(n.child[i], ray, res, tmin, tmax, c);
        }
    }
    // Octree::Node::intersect, include/octree.cpp:256-282: (leaf id, entity) pairs of all touched leaves, DFS order
    void intersect_all(int ni, const Ray& ray, std::vector<std::pair<int, int>>& res, double tmin, double tmax, Counters* c) const
    {
        const ONode& n = onodes[ni];
        if (c) c->v_shadow++;
        if (n.box.intersect_simple(ray, tmin, tmax)) {
            if (n.leaf()) { for (int e : n.ents) res.push_back({ni, e}); }
            else
                for (int i = 0; i < 8; i++)
                    if (n.child[i] >= 0) intersect_all(n.child[i], ray, res, tmin, tmax, c);
        }
    }

    // ---- atmosphere (HeightFog), include/atmosphere.h:50-81, include/octree.cpp:214-251, include/raytracer.h:509-529
    Box fog_box(const Fog& f) const { return Box{f.pos - .5 * f.size, f.pos + .5 * f.size}; }
    double fog_density(const Fog& f, V3 p) const
    {
        const Box bb = fog_box(f);
        const double nscale = (double)f.nscale;
        double ymax = f.pos.y + .5 * f.size.y;
        V3 rel = nscale * (p - bb.min);
        double dx = nscale * (rel.x - (int)rel.x), dy = nscale * (rel.y - (int)rel.y), dz = nscale * (rel.z - (int)rel.z);
        auto g = [&](double idx) { size_t i = (size_t)idx; return f.grid[i < f.grid.size() ? i : f.grid.size() - 1]; };   // the reference reads out of bounds there
        const int rx = (int)rel.x, ry = (int)rel.y, rz = (int)rel.z;
        double c00 = (1 - dx) * g((rx * nscale * f.size.x + ry) * nscale * f.size.z + rz) + dx * g(((rx + 1) * nscale * f.size.x + ry) * nscale * f.size.z + rz);
        double c01 = (1 - dx) * g((rx * nscale * f.size.x + ry) * nscale * f.size.z + rz + 1) + dx * g(((rx + 1) * nscale * f.size.x + ry) * nscale * f.size.z + rz + 1);
        double c10 = (1 - dx) * g((rx * nscale * f.size.x + (ry + 1)) * nscale * f.size.z + rz) + dx * g(((rx + 1) * nscale * f.size.x + (ry + 1)) * nscale * f.size.z + rz);
        double c11 = (1 - dx) * g((rx * nscale * f.size.x + (ry + 1)) * nscale * f.size.z + rz + 1) + dx * g(((rx + 1) * nscale * f.size.x + (ry + 1)) * nscale * f.size.z + rz + 1);
        double c0 = c00 * (1 - dy) + c10 * dy;
        double c1 = c01 * (1 - dy) + c11 * dy;
        double noise = fast_pow((1 - dz) * c0 + dz * c1, 7);
        return f.d * noise * fast_pow((ymax - p.y) / f.size.y, 2);
    }
    double atmosphere_density(V3 pos, V3& col) const   // Octree::atmosphereDensity, include/octree.cpp:214-226
    {
        double d = 0;
        for (const Fog& f : fogs)
            if (fog_box(f).contains(pos)) { col = f.col; d += RAYMARCH_STEPSIZE * fog_density(f, pos); }
        return d;
    }
    bool atmosphere_bounds(const Ray& r, double& mint, double& maxt) const   // Octree::atmosphereBounds, include/octree.cpp:229-251
    {
        double mn = 0, mx = 0;
        bool intersected = false;
        for (const Fog& f : fogs) {
            double tmpmin = 0, tmpmax = 0;
            if (fog_box(f).intersect(r, mint, maxt, tmpmin, tmpmax)) { mn = std::min(mn, tmpmin); mx = std::max(mx, tmpmax); intersected = true; }
        }
        mint = std::max(mint, mn);
        maxt = std::min(maxt, mx);
        return intersected;
    }
    bool raymarch(const Ray& r, V3& hit, V3& col, double mint, double maxt, Rng& rng, uint32_t which) const   // include/raytracer.h:509-529
    {
        double t = mint + SHADOW_BIAS;
        V3 current = r.origin + mint * r.dir;
        uint32_t step = 0;
        while (t < maxt) {
            if (rng.draw(P_FOG, step, which) < atmosphere_density(current, col)) { hit = current; return true; }
            current = current + RAYMARCH_STEPSIZE * r.dir;
            t += RAYMARCH_STEPSIZE;
            step++;
        }
        return false;
    }

    // texture::get / checkerboard::get / imageTexture::get (include/material.h:18-21,38-44,63-68)
    V3 tex_get(int t, V3 constant, V2 uv) const
    {
        if (t < 0) return constant;
        const Texture& x = texs[(size_t)t];
        if (x.kind == 0) return v3(x.p[0], x.p[1], x.p[2]);
        if (x.kind == 1) {
            const int tiles = (int)x.p[6];
            if (((int)(uv.x * tiles) % 2 == 0) ^ ((int)(uv.y * tiles) % 2 == 0)) return v3(x.p[0], x.p[1], x.p[2]);
            return v3(x.p[3], x.p[4], x.p[5]);
        }
        const unsigned char* px = tex_pixel(x, uv);
        // gamma({r/255, g/255, b/255}, 1.0/GAMMA) = pow(c, 1.0 / (1.0/GAMMA)), include/util.h:94-97
        const double g = 1.0 / GAMMA_;
        return v3(std::pow(px[0] / 255.0, 1.0 / g), std::pow(px[1] / 255.0, 1.0 / g), std::pow(px[2] / 255.0, 1.0 / g));
    }
    // texture::getAlpha / imageTexture::getAlpha (include/material.h:23-26,70-78)
    double tex_alpha(int t, V2 uv) const
    {
        if (t < 0) return 1;
        const Texture& x = texs[(size_t)t];
        if (x.kind != 2 || x.p[4] == 0) return 1;
        return tex_pixel(x, uv)[3] / 255.0;
    }
    const unsigned char* tex_pixel(const Texture& x, V2 uv) const   // image.pixelColor(...), include/material.h:65
    {
        const int w = (int)x.p[2], h = (int)x.p[3];
        const int px = std::abs((int)(uv.x * w * x.p[0]) % w);
        const int py = h - std::abs((int)(uv.y * h * x.p[1]) % h) - 1;
        return &tex_pixels[(size_t)x.p[5] + ((size_t)py * w + px) * 4];
    }
    double mat_alpha(const Material& m, V2 uv) const { return m.opacity * tex_alpha(m.dtex, uv); }  // Material::getAlpha, material.h:90-93

    // RayTracer::trace, include/raytracer.h:382-478.  alpha_purpose selects the counter-RNG purpose of the alpha draws.
    bool trace(const Ray& ray, Hit& minHit, int& obj, Rng& rng, Counters* c, int* n_leaves = nullptr, uint32_t alpha_purpose = P_TRACE_ALPHA) const
    {
        Hit h;
        h.uv = V2{0, 0};
        bool intersected = false;
        std::vector<std::pair<int, double>> nodes;
        intersect_sorted(0, ray, nodes, 0, INFINITY, c);
        if (n_leaves) *n_leaves = (int)nodes.size();
        if (c) c->traces++;
        (void)rng.draw(P_TRACE_GUARD);  // the debug-print guard draw, raytracer.h:438 (value unused)
        int current = -1;
        bool term = false;
        size_t nd = 0;
        while (nd != nodes.size() && !term) {
            const ONode& cur = onodes[nodes[nd].first];
            for (int ei : cur.ents) {
                const Entity& e = ents[ei];
                if (c) c->tri++;
                if (ent_intersect(e, ray, h) && (rng.draw(alpha_purpose, (uint32_t)nodes[nd].first, (uint32_t)ei) < mat_alpha(mats[e.mat], h.uv) || mats[e.mat].IOR != 1)) {
                    if (!intersected || len2(h.pos - ray.origin) < len2(minHit.pos - ray.origin)) {
                        current = ei;
                        minHit = h;
                        intersected = true;
                        if (cur.box.contains(h.pos)) term = true;
                    }
                }
            }
            ++nd;
        }
        if (intersected) obj = current;
        return intersected;
    }

    // RayTracer::visible, include/raytracer.h:280-319 (no atmosphere entities in this oracle yet)
    bool visible(const Ray& ray, double mt, Rng& rng, Counters* c, uint32_t light_index, int* n_cand = nullptr) const
    {
        bool hit = false;
        std::vector<std::pair<int, int>> cand;
        cand.reserve(256);
        intersect_all(0, ray, cand, 0, std::sqrt(mt) - SHADOW_BIAS, c);
        if (n_cand) *n_cand = (int)cand.size();
        if (c) c->shadows++;
        size_t k = 0;
        while (!hit && k != cand.size()) {
            const Entity& e = ents[cand[k].second];
            Hit h;
            h.uv = V2{0, 0};   // the reference's `glm::dvec2 uv;` is uninitialised here; only a flat-shaded triangle reads it before writing
            if (c) { c->tri++; c->tri_shadow++; }
            if (ent_intersect(e, ray, h) && (rng.draw(P_SHADOW_ALPHA | (light_index << 8), (uint32_t)cand[k].first, (uint32_t)cand[k].second) < mat_alpha(mats[e.mat], h.uv) || mats[e.mat].IOR != 1)) {
                double t_shadow = len2(h.pos - ray.origin);
                hit = (t_shadow < mt) && (t_shadow > 0);
            }
            ++k;
        }
        if (hit) return false;
        double tmin = 0, tmax = mt;   // the reference passes the SQUARED length here (include/raytracer.h:308-309)
        if (!fogs.empty() && atmosphere_bounds(ray, tmin, tmax)) {
            V3 fh, fc;
            if (raymarch(ray, fh, fc, tmin, tmax, rng, P_FOG_SHADOW + 16 * light_index)) return false;
        }
        return true;
    }

    // RayTracer::rayType, include/raytracer.h:481-506
    int ray_type(const Material& m, const Ray& ray, V3 norm, V2 uv, Rng& rng) const
    {
        int type = 2;
        double IOR = m.IOR;
        double opacity = tex_alpha(m.dtex, uv) * m.opacity;
        double r0 = std::pow((1 - IOR) / (1 + IOR), 2);
        double fs = r0 + (1 - r0) * std::pow(1 - dot(reflect(ray.dir, norm), norm), 5);
        if (m.roughness < .001) type = 0;
        if (rng.draw(P_TYPE_OPACITY) > opacity) {
            if (rng.draw(P_TYPE_FRESNEL) < fs) type = 0;
            else type = 1;
        }
        return type;
    }

    // RayTracer::secondaryRay, include/raytracer.h:321-379
    void secondary_ray(const Ray& ray, const Material& m, V3& norm, V2 uv, double sx, double sy, V3& refDir, V3& f, double& roughness, V3& contrib, double& offset, Rng& rng) const
    {
        bool backface = false;
        if (dot(norm, ray.dir) > 0) { norm = norm * -1.0; backface = true; }
        V3 color = tex_get(m.dtex, m.diffuse, uv);
        roughness = m.roughness;
        int type = ray_type(m, ray, norm, uv, rng);
        if (type == 1) {
            if (backface) refDir = refr(ray.dir, norm, m.IOR);
            else refDir = refr(ray.dir, norm, 1.0 / m.IOR);
            offset *= -1;
            contrib = v3(1, 1, 1);
            f = 1.0 * color;
        } else if (type == 0) {
            refDir = reflect(ray.dir, norm);
            contrib = v3(1, 1, 1);
            f = 1.0 * color;
        } else {
            refDir = hemi_cos_n(norm, (float)sx, (float)sy, 2);
            if (m.roughness < .9) {
                refDir = sample_phong(reflect(ray.dir, norm), norm, (1.0 / (m.roughness)) + 1, sx, sy);
                if (dot(refDir, norm) < 0) refDir = reflect(refDir, norm);
            }
            f = 1.0 * color;
            V3 inf = color;
            contrib = contrib * inf;
            contrib = mix(contrib, inf, 0.5);
        }
    }

    // ---- photon octree: PhotonMap::Node::partition, include/photonMap.cpp:137-192
    void ppartition(int ni)
    {
        Box bb = pnodes[ni].box;
        V3 mid = mix(bb.min, bb.max, .5);
        Box cb[8];
        cb[0] = Box{bb.min, mid};
        cb[1] = Box{v3(bb.min.x + .5 * bb.dx(), bb.min.y, bb.min.z), v3(mid.x + .5 * bb.dx(), mid.y, mid.z)};
        cb[2] = Box{v3(bb.min.x, bb.min.y, bb.min.z + .5 * bb.dz()), v3(mid.x, mid.y, mid.z + .5 * bb.dz())};
        cb[3] = Box{v3(bb.min.x + .5 * bb.dx(), bb.min.y, bb.min.z + .5 * bb.dz()), v3(mid.x + .5 * bb.dx(), mid.y, mid.z + .5 * bb.dz())};
        cb[4] = Box{v3(bb.min.x, bb.min.y + .5 * bb.dy(), bb.min.z), v3(mid.x, mid.y + .5 * bb.dy(), mid.z)};
        cb[5] = Box{v3(bb.min.x + .5 * bb.dx(), bb.min.y + .5 * bb.dy(), bb.min.z), v3(mid.x + .5 * bb.dx(), mid.y + .5 * bb.dy(), mid.z)};
        cb[6] = Box{v3(bb.min.x, bb.min.y + .5 * bb.dy(), bb.min.z + .5 * bb.dz()), v3(mid.x, mid.y + .5 * bb.dy(), mid.z + .5 * bb.dz())};
        cb[7] = Box{mid, bb.max};
        std::vector<int> lists[8];
        for (int p : pnodes[ni].ph)
            for (int i = 0; i < 8; i++)
                if (cb[i].contains(photons[p].origin)) lists[i].push_back(p);
        double avg = 0;
        for (int i = 0; i < 8; i++) avg += (double)lists[i].size();
        avg /= 8;
        size_t parent_count = pnodes[ni].ph.size();
        pnodes[ni].ph.clear();
        bool stop = avg > MAX_SUBDIV_RATIO * parent_count;
        // all 8 children always exist; pre-order: child i with its sub-tree before child i+1.  first_child is the
        // index of child 0; later children are found by skipping sub-trees, so we also record them explicitly.
        int first = -1;
        std::vector<int> kids(8);
        for (int i = 0; i < 8; i++) {
            int ci = (int)pnodes.size();
            if (i == 0) first = ci;
            kids[i] = ci;
            pnodes.push_back(PNode());
            pnodes[ci].box = cb[i];
            pnodes[ci].first_child = -1;
            pnodes[ci].ph = lists[i];
            if (!stop && (int)lists[i].size() > MAX_PHOTONS_PER_LEAF) ppartition(ci);
        }
        pnodes[ni].first_child = first;
        pkids.resize(pnodes.size() * 8, -1);
        for (int i = 0; i < 8; i++) pkids[(size_t)ni * 8 + i] = kids[i];
    }
    std::vector<int> pkids;  // explicit child table [node][8]

    // PhotonMap::rebuild, include/photonMap.cpp:33-47
    void build_pmap()
    {
        pnodes.clear();
        pkids.clear();
        pnodes.push_back(PNode());
        pnodes[0].box = pmap_root_box;
        pnodes[0].first_child = -1;
        for (size_t i = 0; i < photons.size(); i++) pnodes[0].ph.push_back((int)i);
        pkids.resize(8, -1);
        if ((int)photons.size() > MAX_PHOTONS_PER_LEAF) ppartition(0);
        pkids.resize(pnodes.size() * 8, -1);
        pmap_valid = true;
    }
    // PhotonMap::Node::getBounds, include/photonMap.cpp:115-134
    Box pm_bounds(int ni, V3 pos) const
    {
        const PNode& n = pnodes[ni];
        if (n.first_child < 0) return Box{n.box.min - EPSILON, n.box.max + EPSILON};
        int i = 0;
        while (i < 8 && !pnodes[pkids[(size_t)ni * 8 + i]].box.contains(pos)) i++;
        if (i < 8) return pm_bounds(pkids[(size_t)ni * 8 + i], pos);
        return Box{v3(-INFINITY, -INFINITY, -INFINITY), v3(-INFINITY, -INFINITY, -INFINITY)};
    }
    // PhotonMap::Node::get, include/photonMap.cpp:71-92
    void pm_get(int ni, const Box& bbox, std::vector<int>& res) const
    {
        if (bbox.dx() <= 0) return;
        const PNode& n = pnodes[ni];
        if (n.first_child < 0) { res.insert(res.end(), n.ph.begin(), n.ph.end()); }
        else
            for (int i = 0; i < 8; i++) {
                int ci = pkids[(size_t)ni * 8 + i];
                if (pnodes[ci].box.intersect(bbox)) pm_get(ci, bbox, res);
            }
    }
    // RayTracer::samplePhotons, include/raytracer.h:532-579 (+ PhotonMap::getInRange, photonMap.cpp:50-66)
    V3 sample_photons(V3 pos, V3 dir, int count, Counters* c, int* n_cand = nullptr) const
    {
        V3 res = v3(0, 0, 0);
        std::vector<int> cand;
        cand.reserve(256);
        if (!pnodes.empty()) {
            Box bounds = pm_bounds(0, pos);
            pm_get(0, bounds, cand);
        }
        if (n_cand) *n_cand = (int)cand.size();
        if (c) { c->gathers++; c->pcand += (int64_t)cand.size(); }
        count = std::min(count, (int)cand.size());
        std::partial_sort(cand.begin(), cand.begin() + count, cand.end(), [&](int l, int r) { return len2(photons[l].origin - pos) < len2(photons[r].origin - pos); });
        for (int i = 0; i < count; i++) {
            const Photon& p = photons[cand[i]];
            res = res + p.col * dot(p.dir, dir);
        }
        if (cand.size() > 0) {
            double maxDist = len2(photons[cand[count - 1]].origin - pos);
            res = res / (PI * maxDist);
        }
        return res;
    }

    // Light::getPoint(x, y), include/light.h:42-45
    V3 light_point(const Light& l, double x, double y) const { return l.pos + l.rad * random_unit_vec(x, y); }
    // Light::getPointInRange, include/light.h:47-53
    V3 light_point_in_range(const Light& l, double x, double y) const
    {
        if (l.angle < 1) return l.pos + l.rad * sphere_cap_cos(l.dir, (float)x, (float)y, 1, l.angle);
        return l.pos + l.rad * random_unit_vec(x, y);
    }

    // RayTracer::radiance, include/raytracer.h:167-276
    V3 radiance(const Ray& ray, int depth, uint32_t sample, V3 contrib, Rng& rng, Counters* c) const
    {
        if (depth > MAX_DEPTH) return v3(0, 0, 0);
        rng.depth = (uint32_t)depth;
        float sx = sampler().sample(2 + 2 * depth, sample);
        float sy = sampler().sample(3 + 2 * depth, sample);
        double offset = SHADOW_BIAS;
        Hit mh;
        mh.uv = V2{0, 0};
        int current = -1;
        bool intersected = trace(ray, mh, current, rng, c);
        if (intersected) {
            if (c) c->shaded++;
            const Material& m = mats[ents[current].mat];
            V3 i = v3(0, 0, 0);
            V3 refDir;
            V3 color = tex_get(m.dtex, m.diffuse, mh.uv);
            double roughness = m.roughness;
            V3 f = v3(1, 1, 1);
            secondary_ray(ray, m, mh.norm, mh.uv, sx, sy, refDir, f, roughness, contrib, offset, rng);
            {   // include/raytracer.h:209-228
                double tmin = 0, tmax = length(mh.pos - ray.origin);
                if (!fogs.empty() && atmosphere_bounds(ray, tmin, tmax)) {
                    V3 fh, col;
                    if (raymarch(ray, fh, col, tmin, tmax, rng, P_FOG_CAMERA)) {
                        mh.pos = fh;
                        refDir = random_unit_vec(sx, sy);
                        f = 1.0 * col;
                        color = col;
                        contrib = col;
                        roughness = 1;
                    }
                }
            }
            uint32_t li = 0;
            for (const Light& light : lights) {
                bool shadow = false;
                // argument evaluation order of light->getPoint(drand(), drand()) under g++ is right-to-left: y first
                double ry = rng.draw(P_LIGHT_Y | (li << 8));
                double rx = rng.draw(P_LIGHT_X | (li << 8));
                V3 lightDir = light_point(light, rx, ry) - (mh.pos + SHADOW_BIAS * mh.norm);
                double maxt = len2(lightDir);
                double hfrac = 1 / (PI * len2(light.pos - mh.pos));
                Ray shadow_ray(mh.pos + SHADOW_BIAS * mh.norm, lightDir);
                shadow = !visible(shadow_ray, maxt, rng, c, li);
                if (!shadow) {
                    double d = dot(mh.norm, normalize(light.pos - mh.pos));
                    if (d < 0) d = 0;
                    double l = std::pow(d, (1.0 / roughness));
                    i = light.col * l * hfrac;
                }
                li++;
            }
            V3 caustic = depth <= 10 ? sample_photons(mh.pos, refDir, 32, c) : v3(0, 0, 0);
            double q = comp_max(contrib);
            if (depth <= MIN_DEPTH || rng.draw(P_RR) < q) {
                f = f * (depth <= MIN_DEPTH ? 1.0 : (1.0 / q));
                V3 next = radiance(Ray(mh.pos + offset * mh.norm, refDir), depth + 1, sample, contrib, rng, c);
                return color * i + f * next + tex_get(m.etex, m.emissive, mh.uv) + color * caustic;
            }
            return color * i;
        }
        return ambient;
    }

    // camera set-up, include/raytracer.h:74-78
    struct Cam { V3 screenCenter, right; double sw, sh; };
    Cam make_cam(int w, int h) const
    {
        Cam m;
        m.sw = (sensorDiag * w) / (std::sqrt((double)w * w + h * h));
        m.sh = m.sw * ((double)h / w);
        m.screenCenter = cam_pos + focalDist * cam_fwd;
        m.right = normalize(cross(cam_fwd, cam_up));
        return m;
    }
    // primary ray, include/raytracer.h:112-129 (FOCAL_BLUR == 0)
    Ray primary(const Cam& m, const HaltonEnum& he, int w, int h, int s, int x, int y, uint32_t& idx) const
    {
        idx = he.get_index(s, x, y);
        double xr = sampler().sample(0, idx);
        double yr = sampler().sample(1, idx);
        double dx = (float)((float)xr * he.scale_x);
        double dy = (float)((float)yr * he.scale_y);
        V3 pixelPos = m.screenCenter + (m.sw * (dx / w - .5)) * m.right - (m.sh * (dy / h - .5)) * cam_up;
        V3 eyePos = cam_pos + 0 * (xr - .5) * m.right + 0 * (yr - .5) * cam_up;
        return Ray(eyePos, normalize(pixelPos - eyePos));
    }

    // one pixel of RayTracer::run, include/raytracer.h:100-157
    void pixel(const Cam& cam, const HaltonEnum& he, int w, int h, int x, int y, int min_s, int max_s, double thresh, Rng& rng, Counters* c, V3& lin, int& taken) const
    {
        V3 color = v3(0.5, 0.5, 0.5), lastCol = v3(0, 0, 0);
        double var = 0;
        int samps = 0, s = 0;
        while (s < max_s && samps < min_s) {
            lastCol = color;
            uint32_t idx;
            Ray ray = primary(cam, he, w, h, s, x, y, idx);
            rng.stream = idx;
            V3 L = radiance(ray, 0, idx, v3(1, 1, 1), rng, c);
            if (s == 0) color = L;
            else color = (1.0 * s * color + L) * (1.0 / (s + 1));
            if (s > 0) var = (1.0 * 5 * var + length(color - lastCol)) * (1.0 / (5 + 1));
            if (s > 0 && var > thresh) samps -= 2;
            s++;
            samps++;
        }
        lin = color;
        taken = s;
    }

    // RayTracer::tracePhotons, include/raytracer.h:582-715 (single light list, sequential i; identical set for any thread count
    // under the counter RNG because every draw is keyed by (i, light, tries, depth))
    int64_t emit_photons(int count, int maxDepth, Rng& rng)
    {
        photons.clear();
        int64_t total = 0;
        Counters* c = nullptr;
        for (int i = 0; i < count; i++) {
            uint32_t li = 0;
            for (const Light& l : lights) {
                int tries = 0;
                bool stored = false;
                rng.stream = (uint32_t)i * (uint32_t)lights.size() + li;
                while (!stored && tries < 500) {
                    rng.depth = (uint32_t)tries * 16u;
                    float sx = sampler().sample(0, i * 500 + tries);
                    float sy = sampler().sample(1, i * 500 + tries);
                    V3 pos = light_point_in_range(l, sx, sy);
                    // g++ evaluates the arguments right-to-left: the `13*i` draw comes first
                    double d13 = rng.draw(P_PH_DIR_V);
                    double d5 = rng.draw(P_PH_DIR_U);
                    V3 dir = sphere_cap_cos(normalize(pos - l.pos), (float)std::fmod(d5 + 5 * i, 1), (float)std::fmod(d13 + 13 * i, 1), 2, l.angle);
                    Ray r(pos, dir);
                    Hit h;
                    h.uv = V2{0, 0};
                    V3 col = (1.0 / count) * .5 * l.angle * l.col;
                    int current = -1;
                    int depth = 0;
                    bool term = false;
                    bool isCaustic = false;
                    if (!trace(r, h, current, rng, c, nullptr, P_PH_TRACE0_ALPHA)) { tries++; continue; }
                    while (depth < maxDepth && !term) {
                        rng.depth = (uint32_t)tries * 16u + (uint32_t)depth + 1u;
                        double roughness = mats[ents[current].mat].roughness;
                        if (roughness < 0.1) {
                            if (!trace(r, h, current, rng, c)) { term = true; continue; }
                            const Material& m = mats[ents[current].mat];
                            roughness = m.roughness;
                            V3 refDir, f, contrib = v3(0, 0, 0);
                            double offset = SHADOW_BIAS;
                            double e13 = rng.draw(P_PH_SEC_V);
                            double e5 = rng.draw(P_PH_SEC_U);
                            secondary_ray(r, m, h.norm, h.uv, std::fmod(e5 + 5 * i, 1), std::fmod(e13 + 13 * i, 1), refDir, f, roughness, contrib, offset, rng);
                            {   // include/raytracer.h:658-675
                                double tmin = 0, tmax = length(h.pos - r.origin);
                                if (!fogs.empty() && atmosphere_bounds(r, tmin, tmax)) {
                                    V3 ahit, color;
                                    if (raymarch(r, ahit, color, tmin, tmax, rng, P_FOG_PHOTON)) {
                                        h.pos = ahit;
                                        // randomUnitVec(fmod(drand()+13*i,1), fmod(drand()+7*i,1)): g++ draws the second argument first
                                        double g7 = rng.draw(P_PH_FOG_V);
                                        double g13 = rng.draw(P_PH_FOG_U);
                                        refDir = random_unit_vec(std::fmod(g13 + 13 * i, 1), std::fmod(g7 + 7 * i, 1));
                                        f = 1.0 * color;
                                        roughness = 1;
                                    }
                                }
                            }
                            col = col * f;
                            r.origin = h.pos + offset * h.norm;
                            r.set_dir(refDir);
                            isCaustic = true;
                        }
                        if (depth > 0 && isCaustic && roughness >= 0.1) {
                            photons.push_back(Photon{h.pos, r.dir, col});
                            term = true;
                            stored = true;
                        }
                        depth++;
                    }
                    tries++;
                }
                total += tries;
                li++;
            }
        }
        pmap_valid = false;
        return total;
    }
};

// ============================================================================= C ABI
extern "C" {

gio_ctx* gio_create(void) { return new gio_ctx(); }
void gio_destroy(gio_ctx* c) { delete c; }

int gio_set_scene(gio_ctx* c, int n_ent, const int32_t* ent_kind, const double* pos, const double* nrm, const double* uv,
                  const int32_t* mat_idx, int n_mat, const double* mats, int n_light, const double* lights, const double* ambient3)
{
    c->ents.resize(n_ent);
    for (int i = 0; i < n_ent; i++) {
        Entity& e = c->ents[i];
        e.kind = ent_kind ? ent_kind[i] : 0;
        for (int k = 0; k < 3; k++) {
            e.p[k] = v3(pos[i * 9 + k * 3], pos[i * 9 + k * 3 + 1], pos[i * 9 + k * 3 + 2]);
            e.n[k] = v3(nrm[i * 9 + k * 3], nrm[i * 9 + k * 3 + 1], nrm[i * 9 + k * 3 + 2]);
            e.t[k] = V2{uv[i * 6 + k * 2], uv[i * 6 + k * 2 + 1]};
        }
        e.mat = mat_idx[i];
        if (e.mat < 0 || e.mat >= n_mat) return -1;
        // triangle ctor, include/entities.h:335-342
        e.fnorm = normalize(cross((e.p[1] - e.p[0]), (e.p[2] - e.p[0])));
        e.smooth = len2(e.n[0]) > 0 && len2(e.n[1]) > 0 && len2(e.n[2]) > 0;
    }
    c->mats.resize(n_mat);
    for (int i = 0; i < n_mat; i++) {
        const double* m = mats + i * 9;
        c->mats[i] = Material{m[0], m[1], m[2], v3(m[3], m[4], m[5]), v3(m[6], m[7], m[8])};
    }
    c->lights.resize(n_light);
    for (int i = 0; i < n_light; i++) {
        const double* l = lights + i * 7;
        c->lights[i] = Light{v3(l[0], l[1], l[2]), v3(l[3], l[4], l[5]), l[6], v3(0, 0, 0), .125};
    }
    if (ambient3) c->ambient = v3(ambient3[0], ambient3[1], ambient3[2]);
    c->octree_valid = false;
    c->pmap_valid = false;
    return 0;
}

int gio_set_fog(gio_ctx* c, int n, const double* params12, const int32_t* grid_off, const double* grid)
{
    c->fogs.resize(n);
    for (int i = 0; i < n; i++) {
        const double* q = params12 + (size_t)i * 12;
        Fog& f = c->fogs[i];
        f.pos = v3(q[0], q[1], q[2]); f.size = v3(q[3], q[4], q[5]); f.col = v3(q[6], q[7], q[8]);
        f.d = q[9]; f.sc = q[10]; f.nscale = 1;   // the constructor forces nscale = 1 after sizing the grid (include/atmosphere.h:46)
        f.grid.assign(grid + grid_off[i], grid + grid_off[i + 1]);
        if (f.grid.empty()) return -1;
    }
    return 0;
}
// textures as in gi_scene_desc (include/gi_hip.h): call after gio_set_scene
int gio_set_textures(gio_ctx* c, int n_tex, const int32_t* kind, const double* param8, const int32_t* mat_tex, const uint8_t* pixels, int64_t n_bytes)
{
    c->texs.resize((size_t)n_tex);
    for (int i = 0; i < n_tex; i++) { c->texs[i].kind = kind[i]; for (int k = 0; k < 8; k++) c->texs[i].p[k] = param8[(size_t)i * 8 + k]; }
    c->tex_pixels.assign(pixels, pixels + n_bytes);
    for (size_t m = 0; m < c->mats.size(); m++) { c->mats[m].dtex = n_tex ? mat_tex[m * 2] : -1; c->mats[m].etex = n_tex ? mat_tex[m * 2 + 1] : -1; }
    return 0;
}
// known answers of texture::get / getAlpha: out [n][4] = rgb, alpha
int gio_tex_eval(gio_ctx* c, int tex, int n, const double* uv, double* out)
{
    if (tex < 0 || tex >= (int)c->texs.size()) return -1;
    for (int i = 0; i < n; i++) {
        V2 q = V2{uv[i * 2], uv[i * 2 + 1]};
        V3 g = c->tex_get(tex, v3(0, 0, 0), q);
        out[i * 4] = g.x; out[i * 4 + 1] = g.y; out[i * 4 + 2] = g.z; out[i * 4 + 3] = c->tex_alpha(tex, q);
    }
    return 0;
}
int gio_chain_discard(gio_ctx* c, int64_t n)
{
    Rng r; r.state = c->chain_state;
    for (int64_t i = 0; i < n; i++) r.chain_next();
    c->chain_state = r.state;
    return 0;
}

int gio_set_camera(gio_ctx* c, const double* m)
{
    c->cam_pos = v3(m[0], m[1], m[2]);
    c->cam_up = v3(m[3], m[4], m[5]);
    c->cam_fwd = v3(m[6], m[7], m[8]);
    c->sensorDiag = m[9];
    c->focalDist = m[10];
    return 0;
}

int gio_chain_seed(gio_ctx* c, uint64_t seed) { c->chain_state = seed; return 0; }
int gio_build_octree(gio_ctx* c) { c->build_octree(); return 0; }
int gio_octree_counts(gio_ctx* c, int32_t* n_nodes, int32_t* n_refs)
{
    size_t r = 0;
    for (auto& n : c->onodes) r += n.ents.size();
    *n_nodes = (int32_t)c->onodes.size();
    *n_refs = (int32_t)r;
    return 0;
}
int gio_octree_dump(gio_ctx* c, double* bbox, int32_t* child, int32_t* ent_off, int32_t* ent_idx)
{
    int off = 0;
    ent_off[0] = 0;
    for (size_t i = 0; i < c->onodes.size(); i++) {
        const ONode& n = c->onodes[i];
        bbox[i * 6 + 0] = n.box.min.x; bbox[i * 6 + 1] = n.box.min.y; bbox[i * 6 + 2] = n.box.min.z;
        bbox[i * 6 + 3] = n.box.max.x; bbox[i * 6 + 4] = n.box.max.y; bbox[i * 6 + 5] = n.box.max.z;
        for (int k = 0; k < 8; k++) child[i * 8 + k] = n.child[k];
        for (int e : n.ents) ent_idx[off++] = e;
        ent_off[i + 1] = off;
    }
    return 0;
}
int gio_get_lights(gio_ctx* c, double* da)
{
    for (size_t i = 0; i < c->lights.size(); i++) {
        da[i * 4 + 0] = c->lights[i].dir.x; da[i * 4 + 1] = c->lights[i].dir.y; da[i * 4 + 2] = c->lights[i].dir.z; da[i * 4 + 3] = c->lights[i].angle;
    }
    return 0;
}
int gio_ent_bbox(gio_ctx* c, double* bbox)
{
    for (size_t i = 0; i < c->ents.size(); i++) {
        Box b = ent_bbox(c->ents[i]);
        bbox[i * 6 + 0] = b.min.x; bbox[i * 6 + 1] = b.min.y; bbox[i * 6 + 2] = b.min.z;
        bbox[i * 6 + 3] = b.max.x; bbox[i * 6 + 4] = b.max.y; bbox[i * 6 + 5] = b.max.z;
    }
    return 0;
}

static Ray ray_from6(const double* r)
{
    Ray ray;
    ray.origin = v3(r[0], r[1], r[2]);
    ray.set_dir_exact(v3(r[3], r[4], r[5]));
    return ray;
}

int gio_trace(gio_ctx* c, int n, const double* rays, int32_t* hit, int32_t* ent, double* res, int32_t* n_leaves)
{
    if (!c->octree_valid) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        Ray ray = ray_from6(rays + (size_t)i * 6);
        Rng rng;  // counter mode, seed 0: alpha draws only matter for 0 < opacity < 1
        rng.stream = (uint32_t)i;
        Hit h;
        h.pos = v3(0, 0, 0); h.norm = v3(0, 0, 0); h.uv = V2{0, 0};
        int obj = -1, nl = 0;
        bool ok = c->trace(ray, h, obj, rng, nullptr, &nl);
        hit[i] = ok;
        ent[i] = ok ? obj : -1;
        double* o = res + (size_t)i * 8;
        if (ok) { o[0] = h.pos.x; o[1] = h.pos.y; o[2] = h.pos.z; o[3] = h.norm.x; o[4] = h.norm.y; o[5] = h.norm.z; o[6] = h.uv.x; o[7] = h.uv.y; }
        else for (int k = 0; k < 8; k++) o[k] = 0;
        if (n_leaves) n_leaves[i] = nl;
    }
    return 0;
}

// gamma (util.h:94-97), glm::clamp, Image::setPixel (image.h:14-16) for one channel.  pow of a negative value is NaN; the reference's clamp
// passes it on and its (int) cast yields INT_MIN on x86, i.e. byte 0 (pinned by kat_pixel, captured from the reference's own Image).
static uint8_t pixel8(double lin)
{
    const double g = std::pow(lin, 1.0 / 2.2);
    if (g != g) return 0;
    return (uint8_t)(int)(255 * std::min(std::max(g, 0.0), 1.0));
}
void gio_pixel8(int n, const double* lin, uint8_t* out) { for (int i = 0; i < n; i++) out[i] = pixel8(lin[i]); }

int gio_leaf_order(gio_ctx* c, const double* ray6, int cap, int32_t* node, double* t0)
{
    if (!c->octree_valid) return -1;
    Ray ray = ray_from6(ray6);
    std::vector<std::pair<int, double>> v;
    c->intersect_sorted(0, ray, v, 0, INFINITY, nullptr);
    for (int i = 0; i < (int)v.size() && i < cap; i++) { node[i] = v[i].first; t0[i] = v[i].second; }
    return (int)v.size();
}

int gio_visible(gio_ctx* c, int n, const double* q, int32_t* vis, int32_t* n_cand)
{
    if (!c->octree_valid) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        V3 o = v3(q[i * 6], q[i * 6 + 1], q[i * 6 + 2]), t = v3(q[i * 6 + 3], q[i * 6 + 4], q[i * 6 + 5]);
        V3 ld = t - o;
        double maxt = len2(ld);
        Ray sr(o, ld);
        Rng rng;
        rng.stream = (uint32_t)i;
        int nc = 0;
        vis[i] = c->visible(sr, maxt, rng, nullptr, 0, &nc);
        if (n_cand) n_cand[i] = nc;
    }
    return 0;
}

int gio_set_photons(gio_ctx* c, int n, const double* p)
{
    c->photons.resize(n);
    for (int i = 0; i < n; i++)
        c->photons[i] = Photon{v3(p[i * 9], p[i * 9 + 1], p[i * 9 + 2]), v3(p[i * 9 + 3], p[i * 9 + 4], p[i * 9 + 5]), v3(p[i * 9 + 6], p[i * 9 + 7], p[i * 9 + 8])};
    c->pmap_valid = false;
    return 0;
}
int gio_photon_count(gio_ctx* c) { return (int)c->photons.size(); }
int gio_get_photons(gio_ctx* c, double* p)
{
    for (size_t i = 0; i < c->photons.size(); i++) {
        const Photon& q = c->photons[i];
        double v[9] = {q.origin.x, q.origin.y, q.origin.z, q.dir.x, q.dir.y, q.dir.z, q.col.x, q.col.y, q.col.z};
        memcpy(p + i * 9, v, sizeof v);
    }
    return 0;
}
int gio_emit_photons(gio_ctx* c, int count, int max_depth, int rng_mode, uint64_t seed, int64_t* tries_out)
{
    if (!c->octree_valid) return -1;
    Rng rng;
    rng.mode = rng_mode;
    rng.seed = seed ^ PHOTON_SEED_XOR;
    rng.state = c->chain_state;
    int64_t t = c->emit_photons(count, max_depth, rng);
    c->chain_state = rng.state;
    if (tries_out) *tries_out = t;
    return (int)c->photons.size();
}
int gio_build_photon_map(gio_ctx* c)
{
    if (!c->octree_valid) return -1;
    c->build_pmap();
    return 0;
}
int gio_pmap_counts(gio_ctx* c, int32_t* n_nodes, int32_t* n_refs)
{
    size_t r = 0;
    for (auto& n : c->pnodes) r += n.ph.size();
    *n_nodes = (int32_t)c->pnodes.size();
    *n_refs = (int32_t)r;
    return 0;
}
int gio_pmap_dump(gio_ctx* c, double* bbox, int32_t* first_child, int32_t* off, int32_t* idx)
{
    int o = 0;
    off[0] = 0;
    for (size_t i = 0; i < c->pnodes.size(); i++) {
        const PNode& n = c->pnodes[i];
        bbox[i * 6 + 0] = n.box.min.x; bbox[i * 6 + 1] = n.box.min.y; bbox[i * 6 + 2] = n.box.min.z;
        bbox[i * 6 + 3] = n.box.max.x; bbox[i * 6 + 4] = n.box.max.y; bbox[i * 6 + 5] = n.box.max.z;
        first_child[i] = n.first_child;
        for (int p : n.ph) idx[o++] = p;
        off[i + 1] = o;
    }
    return 0;
}
int gio_gather(gio_ctx* c, int n, const double* q, double* res3, int32_t* n_cand)
{
    if (!c->pmap_valid) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        int nc = 0;
        V3 r = c->sample_photons(v3(q[i * 6], q[i * 6 + 1], q[i * 6 + 2]), v3(q[i * 6 + 3], q[i * 6 + 4], q[i * 6 + 5]), 32, nullptr, &nc);
        res3[i * 3] = r.x; res3[i * 3 + 1] = r.y; res3[i * 3 + 2] = r.z;
        if (n_cand) n_cand[i] = nc;
    }
    return 0;
}

static int render_rows(gio_ctx* c, int w, int h, int n_rows, const int32_t* rows, int min_samples, int max_samples, double noise_thresh,
                       int rng_mode, uint64_t seed, int chain_predraws, int n_threads,
                       double* out_lin, uint8_t* out_u8, int32_t* out_spp, int64_t* counters)
{
    if (!c->octree_valid) return -1;
    HaltonEnum he;
    he.init(w, h);
    gio_ctx::Cam cam = c->make_cam(w, h);
    (void)sampler();
    Counters total;
    if (rng_mode == GIO_RNG_CHAIN) n_threads = 1;
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    Rng chain;
    chain.mode = GIO_RNG_CHAIN;
    chain.state = c->chain_state;
    for (int i = 0; i < chain_predraws; i++) chain.chain_next();
#pragma omp parallel num_threads(n_threads)
    {
        Counters local;
        Rng rng_local;
        rng_local.mode = GIO_RNG_COUNTER;
        rng_local.seed = seed;
        // the reference deals rows in chunks of 10 (raytracer.h:93); chunks of 1 keep every core busy on a short row list
#pragma omp for schedule(dynamic, 1)
        for (int ri = 0; ri < n_rows; ri++) {
            const int y = rows[ri];
            for (int x = 0; x < w; x++) {
                Rng& rng = (rng_mode == GIO_RNG_CHAIN) ? chain : rng_local;
                V3 lin;
                int taken = 0;
                c->pixel(cam, he, w, h, x, y, min_samples, max_samples, noise_thresh, rng, counters ? &local : nullptr, lin, taken);
                size_t o = ((size_t)y * w + x) * 3;
                if (out_lin) { out_lin[o] = lin.x; out_lin[o + 1] = lin.y; out_lin[o + 2] = lin.z; }
                if (out_spp) out_spp[(size_t)y * w + x] = taken;
                if (out_u8) {
                    // gamma (util.h:94-97), glm::clamp, Image::setPixel (image.h:14-16)
                    const double l3[3] = {lin.x, lin.y, lin.z};
                    for (int k = 0; k < 3; k++) out_u8[o + k] = pixel8(l3[k]);
                }
            }
        }
#pragma omp critical(gio_counters)
        {
            total.v_trace += local.v_trace; total.v_shadow += local.v_shadow; total.tri += local.tri; total.shaded += local.shaded;
            total.pcand += local.pcand; total.traces += local.traces; total.shadows += local.shadows; total.gathers += local.gathers; total.tri_shadow += local.tri_shadow;
        }
    }
    if (rng_mode == GIO_RNG_CHAIN) c->chain_state = chain.state;
    if (counters) {
        counters[0] = total.v_trace; counters[1] = total.v_shadow; counters[2] = total.tri; counters[3] = total.shaded;
        counters[4] = total.pcand; counters[5] = total.traces; counters[6] = total.shadows; counters[7] = total.gathers; counters[8] = total.tri_shadow;
    }
    return 0;
}

int gio_render(gio_ctx* c, int w, int h, int y0, int y1, int min_samples, int max_samples, double noise_thresh,
               int rng_mode, uint64_t seed, int chain_predraws, int n_threads,
               double* out_lin, uint8_t* out_u8, int32_t* out_spp, int64_t* counters)
{
    std::vector<int32_t> rows;
    for (int y = y0; y < y1; y++) rows.push_back(y);
    return render_rows(c, w, h, (int)rows.size(), rows.data(), min_samples, max_samples, noise_thresh, rng_mode, seed, chain_predraws, n_threads, out_lin, out_u8, out_spp, counters);
}

int gio_render_rows(gio_ctx* c, int w, int h, int n_rows, const int32_t* rows, int min_samples, int max_samples, double noise_thresh,
                    uint64_t seed, int n_threads, double* out_lin, int64_t* counters)
{
    for (int i = 0; i < n_rows; i++) if (rows[i] < 0 || rows[i] >= h) return -2;
    return render_rows(c, w, h, n_rows, rows, min_samples, max_samples, noise_thresh, GIO_RNG_COUNTER, seed, 0, n_threads, out_lin, nullptr, nullptr, counters);
}

int gio_radiance(gio_ctx* c, int n, const double* rays, const uint32_t* stream, uint64_t seed, double* out3)
{
    if (!c->octree_valid) return -1;
    (void)sampler();
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; i++) {
        Ray ray = ray_from6(rays + (size_t)i * 6);
        Rng rng;
        rng.seed = seed;
        rng.stream = stream[i];
        V3 L = c->radiance(ray, 0, stream[i], v3(1, 1, 1), rng, nullptr);
        out3[i * 3] = L.x; out3[i * 3 + 1] = L.y; out3[i * 3 + 2] = L.z;
    }
    return 0;
}

// ---- known-answer helpers
int gio_halton_enum_params(int w, int h, uint32_t* out5)
{
    HaltonEnum e;
    e.init(w, h);
    out5[0] = e.p2; out5[1] = e.p3; out5[2] = e.m_x; out5[3] = e.m_y; out5[4] = e.inc;
    return 0;
}
uint32_t gio_halton_index(int w, int h, uint32_t s, uint32_t x, uint32_t y) { HaltonEnum e; e.init(w, h); return e.get_index(s, x, y); }
float gio_halton_scale(int w, int h, int axis, float v) { HaltonEnum e; e.init(w, h); return axis == 0 ? v * e.scale_x : v * e.scale_y; }
float gio_halton_sample(uint32_t dim, uint32_t index) { return sampler().sample(dim, index); }
double gio_fast_pow(double a, double b) { return fast_pow(a, b); }
double gio_fast_precise_pow(double a, double b) { return fast_precise_pow(a, b); }
static void put3(double* o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
static V3 get3(const double* p) { return v3(p[0], p[1], p[2]); }
void gio_hemi_cos_n(const double* n3, float u, float v, double power, double* out3) { put3(out3, hemi_cos_n(get3(n3), u, v, power)); }
void gio_hemi_cos(float u, float v, double power, double* out3) { put3(out3, hemi_cos(u, v, power)); }
void gio_sample_phong(const double* o3, const double* n3, double power, double sx, double sy, double* out3) { put3(out3, sample_phong(get3(o3), get3(n3), power, sx, sy)); }
void gio_sphere_cap(const double* n3, float u, float v, double power, double frac, double* out3) { put3(out3, sphere_cap_cos(get3(n3), u, v, power, frac)); }
void gio_unit_vec(double x, double y, double* out3) { put3(out3, random_unit_vec(x, y)); }
void gio_refr(const double* inc3, const double* n3, double eta, double* out3) { put3(out3, refr(get3(inc3), get3(n3), eta)); }
void gio_reflect(const double* inc3, const double* n3, double* out3) { put3(out3, reflect(get3(inc3), get3(n3))); }
int gio_tri_box_overlap(const double* c3, const double* h3, const double* v9)
{
    V3 tv[3] = {get3(v9), get3(v9 + 3), get3(v9 + 6)};
    return tri_box_overlap(get3(c3), get3(h3), tv) ? 1 : 0;
}
double gio_chain_drand(uint64_t* state) { Rng r; r.state = *state; double v = r.chain_next(); *state = r.state; return v; }
double gio_counter_rand(uint64_t seed, uint32_t stream, uint32_t depth, uint32_t purpose, uint32_t a, uint32_t b) { return counter_rand(seed, stream, depth, purpose, a, b); }
uint32_t gio_primary_ray(gio_ctx* c, int w, int h, int s, int x, int y, double* ray6)
{
    HaltonEnum he;
    he.init(w, h);
    gio_ctx::Cam cam = c->make_cam(w, h);
    uint32_t idx;
    Ray r = c->primary(cam, he, w, h, s, x, y, idx);
    ray6[0] = r.origin.x; ray6[1] = r.origin.y; ray6[2] = r.origin.z; ray6[3] = r.dir.x; ray6[4] = r.dir.y; ray6[5] = r.dir.z;
    return idx;
}

}  // extern "C"
