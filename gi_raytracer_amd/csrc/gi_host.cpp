// gi_host.cpp -- host-side feeders of the hot path: .scn/.obj ingestion, scene octree and photon octree builders,
// flattening into the C-ABI tables of include/gi_hip.h.  See gi_host.h.  Host code by design (the reference builds its
// trees on the host as well); the trees produced here are node-for-node those of the reference (tests/test_host_builders.py
// compares them with dumps of the reference's own trees).
#include "gi_host.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------- PNG -> RGBA8 as QImage::pixelColor presents it
// 8-bit, non-interlaced PNGs of colour type 0 (grey), 2 (RGB), 3 (palette, tRNS), 4 (grey + alpha), 6 (RGBA).  QImage converts none of these
// (no gamma / colour management unless asked), so pixelColor() returns the stored channel values; 16-bit and interlaced files are
// refused.  has_alpha = what QImage::hasAlphaChannel() reports: an alpha channel in the file or a tRNS chunk.
inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
bool decode_png(const std::string& path, int& w, int& h, bool& has_alpha, std::vector<uint8_t>& rgba, std::string& err)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open image: " + path; return false; }
    std::vector<unsigned char> file;
    unsigned char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file: " + path; return false; }
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    w = h = 0;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(&file[pos]);
        const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
        if (pos + 12 + (size_t)len > file.size()) { err = "truncated PNG: " + path; return false; }
        const unsigned char* data = &file[pos + 8];
        if (memcmp(type, "IHDR", 4) == 0 && len >= 13) { w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (memcmp(type, "PLTE", 4) == 0) plte.assign(data, data + len);
        else if (memcmp(type, "tRNS", 4) == 0) trns.assign(data, data + len);
        else if (memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), data, data + len);
        else if (memcmp(type, "IEND", 4) == 0) break;
        pos += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768) { err = "PNG without a usable IHDR: " + path; return false; }
    if (depth != 8 || interlace != 0) { err = "PNG must be 8 bits per channel and not interlaced: " + path; return false; }
    int ch = 0;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: err = "PNG colour type not supported: " + path; return false; }
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { err = "PNG data does not inflate: " + path; return false; }
    std::vector<unsigned char> img(stride * (size_t)h);
    for (int y = 0; y < h; y++) {                       // undo the scanline filters (PNG spec 9.2)
        const unsigned char* in = &raw[(stride + 1) * (size_t)y];
        unsigned char* out = &img[stride * (size_t)y];
        const unsigned char* up = y ? out - stride : nullptr;
        const int ft = in[0];
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= (size_t)ch ? out[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) { err = "PNG filter type out of range: " + path; return false; }
            out[x] = (unsigned char)(in[1 + x] + pred);
        }
    }
    has_alpha = ctype == 4 || ctype == 6 || !trns.empty();
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        unsigned char r = 0, g = 0, b = 0, a = 255;
        const unsigned char* px = &img[i * ch];
        if (ctype == 0) { r = g = b = px[0]; if (trns.size() >= 2 && px[0] == trns[1]) a = 0; }
        else if (ctype == 2) { r = px[0]; g = px[1]; b = px[2]; if (trns.size() >= 6 && r == trns[1] && g == trns[3] && b == trns[5]) a = 0; }
        else if (ctype == 3) { const size_t k = px[0]; if (k * 3 + 2 < plte.size()) { r = plte[k * 3]; g = plte[k * 3 + 1]; b = plte[k * 3 + 2]; } if (k < trns.size()) a = trns[k]; }
        else if (ctype == 4) { r = g = b = px[0]; a = px[1]; }
        else { r = px[0]; g = px[1]; b = px[2]; a = px[3]; }
        rgba[i * 4] = r; rgba[i * 4 + 1] = g; rgba[i * 4 + 2] = b; rgba[i * 4 + 3] = a;
    }
    return true;
}

const int kMaxEntitiesPerLeaf = 16;    // include/util.h:14
const int kMaxPhotonsPerLeaf = 16;     // include/util.h:15
const double kMinLeafSize = .0015;     // include/util.h:16
const double kMaxSubdivRatio = 0.75;   // include/util.h:17
const double kEps = 0.00001;           // include/util.h:18
const double kPi = 3.14159265358979323846;

struct D3 { double x, y, z; };
inline D3 mk(double x, double y, double z) { D3 r = {x, y, z}; return r; }
inline D3 sub(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline D3 add(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline D3 scl(D3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
inline double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 cross3(D3 x, D3 y) { return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
inline D3 unit(D3 v) { return scl(v, 1.0 / std::sqrt(dot3(v, v))); }   // glm::normalize = v * inversesqrt(dot)
inline double comp(const D3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

struct Aabb { D3 lo, hi; };
inline double ext_x(const Aabb& b) { return b.hi.x - b.lo.x; }
inline double ext_y(const Aabb& b) { return b.hi.y - b.lo.y; }
inline double ext_z(const Aabb& b) { return b.hi.z - b.lo.z; }
inline D3 centre(const Aabb& b) { return add(b.lo, scl(sub(b.hi, b.lo), 0.5)); }   // min + 0.5*(max-min), include/bbox.h:25
inline bool closed_overlap(const Aabb& a, const Aabb& o)                             // include/bbox.h:33-38
{
    return (a.lo.x <= o.hi.x && a.hi.x >= o.lo.x) && (a.lo.y <= o.hi.y && a.hi.y >= o.lo.y) && (a.lo.z <= o.hi.z && a.hi.z >= o.lo.z);
}
inline bool half_open_contains(const Aabb& b, D3 p)                                  // include/bbox.h:41-44
{
    return p.x >= b.lo.x && p.y >= b.lo.y && p.z >= b.lo.z && p.x < b.hi.x && p.y < b.hi.y && p.z < b.hi.z;
}

// the 8 octants in the reference's child numbering and with its exact corner expressions (include/octree.cpp:318-328,
// include/photonMap.cpp:139-149): x is bit 0, z is bit 1, y is bit 2; upper corners are mid + .5*extent, not the parent max
void octants(const Aabb& b, Aabb out[8])
{
    const D3 mid = add(b.lo, scl(sub(b.hi, b.lo), .5));   // glm::mix(min, max, .5) = min + .5*(max - min)
    const double hx = .5 * ext_x(b), hy = .5 * ext_y(b), hz = .5 * ext_z(b);
    for (int i = 0; i < 8; i++) {
        const bool ux = i & 1, uz = i & 2, uy = i & 4;
        out[i].lo = mk(ux ? b.lo.x + hx : b.lo.x, uy ? b.lo.y + hy : b.lo.y, uz ? b.lo.z + hz : b.lo.z);
        out[i].hi = mk(ux ? mid.x + hx : mid.x, uy ? mid.y + hy : mid.y, uz ? mid.z + hz : mid.z);
    }
    out[0].lo = b.lo; out[0].hi = mid;
    out[7].lo = mid; out[7].hi = b.hi;
}

// Akenine-Moeller triangle/box overlap as the reference compiles it (include/util.cpp:186-330): the projections, radii,
// extents and the plane offset are held in *float* variables while the vectors stay double.
struct TriBox {
    D3 v0, v1, v2, h;
    float lo, hi, rad;
    bool sep(double a, double b, float r) { if (a < b) { lo = (float)a; hi = (float)b; } else { lo = (float)b; hi = (float)a; } rad = r; return lo > rad || hi < -rad; }
};
bool tri_overlaps_box(D3 c, D3 half, D3 t0, D3 t1, D3 t2)
{
    TriBox T;
    T.v0 = sub(t0, c); T.v1 = sub(t1, c); T.v2 = sub(t2, c); T.h = half;
    const D3 e0 = sub(T.v1, T.v0), e1 = sub(T.v2, T.v1), e2 = sub(T.v0, T.v2);
    const D3 &v0 = T.v0, &v1 = T.v1, &v2 = T.v2;
    float fx, fy, fz, p, q;
    // edge e0: X01, Y02, Z12
    fx = (float)std::fabs(e0.x); fy = (float)std::fabs(e0.y); fz = (float)std::fabs(e0.z);
    p = (float)(e0.z * v0.y - e0.y * v0.z); q = (float)(e0.z * v2.y - e0.y * v2.z);
    if (T.sep(p, q, (float)(fz * half.y + fy * half.z))) return false;
    p = (float)(-e0.z * v0.x + e0.x * v0.z); q = (float)(-e0.z * v2.x + e0.x * v2.z);
    if (T.sep(p, q, (float)(fz * half.x + fx * half.z))) return false;
    p = (float)(e0.y * v1.x - e0.x * v1.y); q = (float)(e0.y * v2.x - e0.x * v2.y);
    if (T.sep(q, p, (float)(fy * half.x + fx * half.y))) return false;
    // edge e1: X01, Y02, Z0
    fx = (float)std::fabs(e1.x); fy = (float)std::fabs(e1.y); fz = (float)std::fabs(e1.z);
    p = (float)(e1.z * v0.y - e1.y * v0.z); q = (float)(e1.z * v2.y - e1.y * v2.z);
    if (T.sep(p, q, (float)(fz * half.y + fy * half.z))) return false;
    p = (float)(-e1.z * v0.x + e1.x * v0.z); q = (float)(-e1.z * v2.x + e1.x * v2.z);
    if (T.sep(p, q, (float)(fz * half.x + fx * half.z))) return false;
    p = (float)(e1.y * v0.x - e1.x * v0.y); q = (float)(e1.y * v1.x - e1.x * v1.y);
    if (T.sep(p, q, (float)(fy * half.x + fx * half.y))) return false;
    // edge e2: X2, Y1, Z12
    fx = (float)std::fabs(e2.x); fy = (float)std::fabs(e2.y); fz = (float)std::fabs(e2.z);
    p = (float)(e2.z * v0.y - e2.y * v0.z); q = (float)(e2.z * v1.y - e2.y * v1.z);
    if (T.sep(p, q, (float)(fz * half.y + fy * half.z))) return false;
    p = (float)(-e2.z * v0.x + e2.x * v0.z); q = (float)(-e2.z * v1.x + e2.x * v1.z);
    if (T.sep(p, q, (float)(fz * half.x + fx * half.z))) return false;
    p = (float)(e2.y * v1.x - e2.x * v1.y); q = (float)(e2.y * v2.x - e2.x * v2.y);
    if (T.sep(q, p, (float)(fy * half.x + fx * half.y))) return false;
    // the three box axes: min/max of the vertex coordinates kept in float, compared with the double half size
    for (int ax = 0; ax < 3; ax++) {
        const double a = comp(v0, ax), b = comp(v1, ax), cc = comp(v2, ax);
        float mn = (float)a, mx = (float)a;
        if (b < mn) mn = (float)b;
        if (b > mx) mx = (float)b;
        if (cc < mn) mn = (float)cc;
        if (cc > mx) mx = (float)cc;
        if (mn > comp(half, ax) || mx < -comp(half, ax)) return false;
    }
    // triangle plane against the box
    const D3 n = cross3(e0, e1);
    const float d = (float)(-dot3(n, v0));
    D3 vmin, vmax;
    vmin.x = n.x > 0.0f ? -half.x : half.x; vmax.x = n.x > 0.0f ? half.x : -half.x;
    vmin.y = n.y > 0.0f ? -half.y : half.y; vmax.y = n.y > 0.0f ? half.y : -half.y;
    vmin.z = n.z > 0.0f ? -half.z : half.z; vmax.z = n.z > 0.0f ? half.z : -half.z;
    if (dot3(n, vmin) + d > 0.0f) return false;
    if (dot3(n, vmax) + d >= 0.0f) return true;
    return false;
}

// glm::eulerAngleXYZ upper 3x3 (3rd_party/glm/gtx/euler_angles.inl:135-167), column-major m[col][row]
// Entity::boundingBox and Entity::intersect(BoundingBox) of the two entity kinds (0 triangle: three vertices; 1 sphere: centre = v[0], radius = v[1].x)
const double kEntEps = 0.00001;   // EPSILON, include/util.h:18
// triangle::boundingBox, include/entities.h:530-557: the upper corner grows by EPSILON after EVERY vertex; sphere::boundingBox, :103-106
Aabb entity_box(int kind, const D3 v[3])
{
    Aabb b;
    if (kind == 1) {
        const D3 c = v[0];
        const double rad = v[1].x;
        b.lo = mk(c.x + rad * -1, c.y + rad * -1, c.z + rad * -1);
        b.hi = mk(c.x + rad * 1, c.y + rad * 1, c.z + rad * 1);
        return b;
    }
    b.lo = mk(INFINITY, INFINITY, INFINITY); b.hi = mk(-INFINITY, -INFINITY, -INFINITY);
    for (int k = 0; k < 3; k++) {
        const D3 p = v[k];
        if (p.x < b.lo.x) b.lo.x = p.x;
        if (p.x > b.hi.x) b.hi.x = p.x;
        if (p.y < b.lo.y) b.lo.y = p.y;
        if (p.y > b.hi.y) b.hi.y = p.y;
        if (p.z < b.lo.z) b.lo.z = p.z;
        if (p.z > b.hi.z) b.hi.z = p.z;
        b.hi.x += kEntEps; b.hi.y += kEntEps; b.hi.z += kEntEps;
    }
    return b;
}
// triangle::intersect(BoundingBox), include/entities.h:522-528; sphere::intersect(BoundingBox), :108-141 (squared distance centre-box <= rad^2)
bool entity_in_cell(int kind, const D3 v[3], const Aabb& cell)
{
    if (kind == 1) {
        const D3 c = v[0];
        const double rad = v[1].x;
        double sq = 0.0;
        for (int ax = 0; ax < 3; ax++) {
            const double q = comp(c, ax), lo = comp(cell.lo, ax), hi = comp(cell.hi, ax);
            double out = 0;
            if (q < lo) { const double val = (lo - q); out += val * val; }
            if (q > hi) { const double val = (q - hi); out += val * val; }
            sq += out;
        }
        return sq <= (rad * rad);
    }
    Aabb g;
    g.lo = mk(cell.lo.x - kEntEps, cell.lo.y - kEntEps, cell.lo.z - kEntEps);
    g.hi = mk(cell.hi.x + kEntEps, cell.hi.y + kEntEps, cell.hi.z + kEntEps);
    return tri_overlaps_box(centre(g), mk(ext_x(g) / 2, ext_y(g) / 2, ext_z(g) / 2), v[0], v[1], v[2]);
}

struct M3 { double m[3][3]; };
M3 euler_xyz(double t1, double t2, double t3)
{
    const double c1 = std::cos(-t1), c2 = std::cos(-t2), c3 = std::cos(-t3);
    const double s1 = std::sin(-t1), s2 = std::sin(-t2), s3 = std::sin(-t3);
    M3 r;
    r.m[0][0] = c2 * c3;  r.m[0][1] = -c1 * s3 + s1 * s2 * c3;  r.m[0][2] = s1 * s3 + c1 * s2 * c3;
    r.m[1][0] = c2 * s3;  r.m[1][1] = c1 * c3 + s1 * s2 * s3;   r.m[1][2] = -s1 * c3 + c1 * s2 * s3;
    r.m[2][0] = -s2;      r.m[2][1] = s1 * c2;                  r.m[2][2] = c1 * c2;
    return r;
}
inline D3 mul(const M3& r, D3 v)
{
    return mk(r.m[0][0] * v.x + r.m[1][0] * v.y + r.m[2][0] * v.z, r.m[0][1] * v.x + r.m[1][1] * v.y + r.m[2][1] * v.z,
              r.m[0][2] * v.x + r.m[1][2] * v.y + r.m[2][2] * v.z);
}

}  // namespace

struct gih_scene {
    std::string err;
    // entities (triangles), insertion order = Octree::_root._entities order
    std::vector<double> tri_pos, tri_nrm, tri_uv;
    std::vector<int32_t> tri_mat;
    std::vector<int32_t> ent_kind;   // 0 triangle, 1 sphere (centre = vertex 0, radius = vertex 1 x)
    std::vector<double> mats;     // 9 per material
    // textures (include/material.h): kind, 8 parameters, RGBA8 pixel pool; mat_tex = (diffuse, emissive) texture per material, -1 for a
    // material given as plain colours (gih_add_material)
    std::vector<int32_t> tex_kind, mat_tex;
    std::vector<double> tex_param;
    std::vector<uint8_t> tex_pixels;
    bool textured() const { for (int32_t k : tex_kind) if (k != 0) return true; return false; }
    int add_texture(int kind, const double* p8, const uint8_t* rgba, long long n_bytes)
    {
        double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (kind < 0 || kind > 2 || !p8) { err = "texture: unknown kind"; return -2; }
        for (int k = 0; k < 8; k++) q[k] = p8[k];
        if (kind == 2) {
            const long long w = (long long)q[2], h = (long long)q[3];
            if (w <= 0 || h <= 0 || !rgba || n_bytes != w * h * 4) { err = "image texture: width x height x 4 bytes of RGBA expected"; return -2; }
            q[5] = (double)tex_pixels.size();
            tex_pixels.insert(tex_pixels.end(), rgba, rgba + n_bytes);
        }
        tex_kind.push_back(kind);
        tex_param.insert(tex_param.end(), q, q + 8);
        tree_valid = false;
        return (int)tex_kind.size() - 1;
    }
    int add_material_tex(int dif, int em, double r, double o, double ior)
    {
        const int nt = (int)tex_kind.size();
        if (dif < 0 || em < 0 || dif >= nt || em >= nt) { err = "mat: texture index out of range"; return -2; }
        // the colour fields hold texture::color: the colour of a colorTex, (0, 0, 0) for the others (include/material.h:34,57)
        const double* pd = &tex_param[(size_t)dif * 8];
        const double* pe = &tex_param[(size_t)em * 8];
        const bool cd = tex_kind[dif] == 0, ce = tex_kind[em] == 0;
        const double m[9] = {r, o, ior, cd ? pd[0] : 0, cd ? pd[1] : 0, cd ? pd[2] : 0, ce ? pe[0] : 0, ce ? pe[1] : 0, ce ? pe[2] : 0};
        mats.insert(mats.end(), m, m + 9);
        while (mat_tex.size() < (mats.size() / 9 - 1) * 2) mat_tex.push_back(-1);
        mat_tex.push_back(dif); mat_tex.push_back(em);
        tree_valid = false;
        return (int)(mats.size() / 9) - 1;
    }
    std::vector<double> lights;   // 11 per light
    std::vector<double> fog, fog_grid;   // 12 per HeightFog; concatenated noise grids
    std::vector<int32_t> fog_grid_off = std::vector<int32_t>(1, 0);
    gih_settings st;
    // octree
    bool tree_valid = false;
    std::vector<double> node_bbox;
    std::vector<int32_t> node_child, node_ent_off, node_ent_idx;
    Aabb root_box;
    // photon map
    std::vector<double> photons;
    std::vector<double> pm_bbox;
    std::vector<int32_t> pm_child, pm_off, pm_idx;

    std::vector<Aabb> tri_box;    // cached triangle::boundingBox()

    gih_scene()
    {
        memset(&st, 0, sizeof st);
        st.photons = 75000; st.photon_depth = 5; st.min_samples = 8; st.max_samples = 32; st.noise_thresh = 0.0015;   // include/util.h:24-29
        const double pos[3] = {10, 5, 0}, look[3] = {0, 0, 0};                                                          // main.cpp:28
        st.sensor_diag = 0.035 * 240 * 2; st.focal_dist = 0.04 * 240;                                                   // include/camera.h:4,29-30
        set_camera(pos, look);
        root_box.lo = mk(0, 0, 0); root_box.hi = mk(0, 0, 0);
    }
    int n_tri() const { return (int)tri_mat.size(); }

    void set_camera(const double* pos, const double* look)   // Camera ctor, include/camera.h:9-15
    {
        D3 p = mk(pos[0], pos[1], pos[2]);
        D3 f = unit(sub(mk(look[0], look[1], look[2]), p));
        D3 up = mk(0, 1.0, 0);
        D3 right = unit(cross3(up, f));
        up = cross3(f, right);
        st.cam_pos[0] = p.x; st.cam_pos[1] = p.y; st.cam_pos[2] = p.z;
        st.cam_forward[0] = f.x; st.cam_forward[1] = f.y; st.cam_forward[2] = f.z;
        st.cam_up[0] = up.x; st.cam_up[1] = up.y; st.cam_up[2] = up.z;
    }

    D3 vert(int t, int k) const { const double* p = &tri_pos[(size_t)t * 9 + k * 3]; return mk(p[0], p[1], p[2]); }

    Aabb triangle_box(int t) const { const D3 v[3] = {vert(t, 0), vert(t, 1), vert(t, 2)}; return entity_box(ent_kind[t], v); }
    bool triangle_in_cell(int t, const Aabb& cell) const { const D3 v[3] = {vert(t, 0), vert(t, 1), vert(t, 2)}; return entity_in_cell(ent_kind[t], v, cell); }

    void push_triangle(const D3 p[3], const D3 n[3], const double uv[6], int mat)
    {
        for (int k = 0; k < 3; k++) { tri_pos.push_back(p[k].x); tri_pos.push_back(p[k].y); tri_pos.push_back(p[k].z); }
        for (int k = 0; k < 3; k++) { tri_nrm.push_back(n[k].x); tri_nrm.push_back(n[k].y); tri_nrm.push_back(n[k].z); }
        for (int k = 0; k < 6; k++) tri_uv.push_back(uv[k]);
        tri_mat.push_back(mat);
        ent_kind.push_back(0);
        tree_valid = false;
    }
    void push_sphere(D3 c, double rad, int mat)   // new sphere(pos, rad, mat), include/entities.h:55-58
    {
        const D3 p[3] = {c, mk(rad, 0, 0), mk(0, 0, 0)}, z[3] = {mk(0, 0, 0), mk(0, 0, 0), mk(0, 0, 0)};
        const double uv[6] = {0, 0, 0, 0, 0, 0};
        push_triangle(p, z, uv, mat);
        ent_kind.back() = 1;
    }

    // ---------------------------------------------------------------- Octree::Node::partition as an emitter of pre-order arrays
    int new_node(const Aabb& b)
    {
        const int id = (int)(node_bbox.size() / 6);
        const double v[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
        node_bbox.insert(node_bbox.end(), v, v + 6);
        for (int k = 0; k < 8; k++) node_child.push_back(-1);
        node_ent_off.push_back((int32_t)node_ent_idx.size());   // start; the end is the next node's start
        return id;
    }
    // emits node `id` (already created, box `cell`) holding `items`; split decides whether it is partitioned
    void emit_node(int id, const Aabb& cell, const std::vector<int>& items, bool split)
    {
        if (!split) {   // leaf: owns its items
            node_ent_idx.insert(node_ent_idx.end(), items.begin(), items.end());
            return;
        }
        Aabb oc[8];
        octants(cell, oc);
        std::vector<int> part[8];
        for (int t : items) {
            const Aabb& tb = tri_box[t];
            for (int i = 0; i < 8; i++)
                if (closed_overlap(oc[i], tb) && triangle_in_cell(t, oc[i]) && ext_x(tb) > kEps) part[i].push_back(t);
        }
        double mean = 0;
        for (int i = 0; i < 8; i++) mean += (double)part[i].size();
        mean /= 8;
        const bool frozen = mean > kMaxSubdivRatio * (double)items.size();   // "skipped" subdivision: children stay leaves
        for (int i = 0; i < 8; i++) {
            if (part[i].empty()) continue;   // empty children are null
            const int ch = new_node(oc[i]);
            node_child[(size_t)id * 8 + i] = ch;
            const bool again = !frozen && (int)part[i].size() > kMaxEntitiesPerLeaf && ext_x(oc[i]) > kMinLeafSize;
            emit_node(ch, oc[i], part[i], again);
        }
    }

    int build_octree()
    {
        const int T = n_tri();
        tri_box.resize((size_t)T);
        for (int t = 0; t < T; t++) tri_box[t] = triangle_box(t);
        // Octree::push_back, include/octree.cpp:25-38: root box = union of entity boxes (root starts at (0,0,0)-(0,0,0))
        root_box.lo = mk(0, 0, 0); root_box.hi = mk(0, 0, 0);
        for (int t = 0; t < T; t++) {
            const Aabb& b = tri_box[t];
            if (t == 0) root_box = b;
            root_box.hi = mk(std::fmax(root_box.hi.x, b.hi.x), std::fmax(root_box.hi.y, b.hi.y), std::fmax(root_box.hi.z, b.hi.z));
            root_box.lo = mk(std::fmin(root_box.lo.x, b.lo.x), std::fmin(root_box.lo.y, b.lo.y), std::fmin(root_box.lo.z, b.lo.z));
        }
        // light cones, include/octree.cpp:60-102 (the running avgPos keeps growing inside the light loop, as in the reference)
        D3 avg = mk(0, 0, 0);
        double cnt = 0;
        for (int t = 0; t < T; t++)
            if (mats[(size_t)tri_mat[t] * 9] < 0.1) { avg = add(avg, centre(tri_box[t])); cnt++; }
        if (cnt > 0) avg = mk(avg.x / cnt, avg.y / cnt, avg.z / cnt);
        for (size_t l = 0; l < lights.size() / 11; l++) {
            double* L = &lights[l * 11];
            const D3 lp = mk(L[0], L[1], L[2]);
            const D3 dir = unit(sub(avg, lp));
            double widest = 0;
            for (int t = 0; t < T; t++)
                if (mats[(size_t)tri_mat[t] * 9] < 0.1) {
                    avg = add(avg, centre(tri_box[t]));
                    const double ang = 1.0 - std::acos(dot3(dir, unit(sub(lp, tri_box[t].lo)))) / kPi;
                    widest = std::fmax(widest, ang);
                    cnt++;
                }
            L[7] = dir.x; L[8] = dir.y; L[9] = dir.z; L[10] = widest;
        }
        node_bbox.clear(); node_child.clear(); node_ent_off.clear(); node_ent_idx.clear();
        std::vector<int> all((size_t)T);
        for (int t = 0; t < T; t++) all[t] = t;
        const int root = new_node(root_box);
        emit_node(root, root_box, all, T > kMaxEntitiesPerLeaf);
        node_ent_off.push_back((int32_t)node_ent_idx.size());
        tree_valid = true;
        return 0;
    }

    // ---------------------------------------------------------------- PhotonMap::Node::partition, include/photonMap.cpp:137-192
    int pm_new_node(const Aabb& b)
    {
        const int id = (int)(pm_bbox.size() / 6);
        const double v[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
        pm_bbox.insert(pm_bbox.end(), v, v + 6);
        for (int k = 0; k < 8; k++) pm_child.push_back(-1);
        pm_off.push_back((int32_t)pm_idx.size());
        return id;
    }
    void pm_emit(int id, const Aabb& cell, const std::vector<int>& items, bool split)
    {
        if (!split) { pm_idx.insert(pm_idx.end(), items.begin(), items.end()); return; }
        Aabb oc[8];
        octants(cell, oc);
        std::vector<int> part[8];
        for (int p : items) {
            const D3 o = mk(photons[(size_t)p * 9], photons[(size_t)p * 9 + 1], photons[(size_t)p * 9 + 2]);
            for (int i = 0; i < 8; i++)
                if (half_open_contains(oc[i], o)) part[i].push_back(p);
        }
        double mean = 0;
        for (int i = 0; i < 8; i++) mean += (double)part[i].size();
        mean /= 8;
        const bool frozen = mean > kMaxSubdivRatio * (double)items.size();
        for (int i = 0; i < 8; i++) {   // all 8 children always exist
            const int ch = pm_new_node(oc[i]);
            pm_child[(size_t)id * 8 + i] = ch;
            pm_emit(ch, oc[i], part[i], !frozen && (int)part[i].size() > kMaxPhotonsPerLeaf);
        }
    }
    int build_photon_map(int n, const double* ph)
    {
        photons.assign(ph, ph + (size_t)n * 9);
        pm_bbox.clear(); pm_child.clear(); pm_off.clear(); pm_idx.clear();
        std::vector<int> all((size_t)n);
        for (int i = 0; i < n; i++) all[i] = i;
        const int root = pm_new_node(root_box);
        pm_emit(root, root_box, all, n > kMaxPhotonsPerLeaf);
        pm_off.push_back((int32_t)pm_idx.size());
        return 0;
    }

    // ---------------------------------------------------------------- loaders
    int load_obj(const std::string& path, D3 pos, D3 rot_angles, int mat)   // include/meshLoader.cpp:18-99
    {
        std::vector<float> v, vt, vn;   // the reference keeps these as glm::vec3 / vec2 (float)
        const M3 R = euler_xyz(rot_angles.x, rot_angles.y, rot_angles.z);
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return 1;   // missing mesh: skipped, as in the reference
        char word[128];
        for (;;) {
            if (fscanf(f, "%127s", word) == EOF) break;
            if (strcmp(word, "v") == 0) {
                D3 p;
                fscanf(f, "%lf %lf %lf\n", &p.x, &p.y, &p.z);
                const D3 w = add(mul(R, p), pos);
                v.push_back((float)w.x); v.push_back((float)w.y); v.push_back((float)w.z);
            } else if (strcmp(word, "vt") == 0) {
                double a, b;
                fscanf(f, "%lf %lf\n", &a, &b);
                vt.push_back((float)a); vt.push_back((float)b);
            } else if (strcmp(word, "vn") == 0) {
                D3 p;
                fscanf(f, "%lf %lf %lf\n", &p.x, &p.y, &p.z);
                const D3 w = mul(R, p);
                vn.push_back((float)w.x); vn.push_back((float)w.y); vn.push_back((float)w.z);
            } else if (strcmp(word, "f") == 0) {
                unsigned vi[3], ti[3], ni[3];
                const int got = fscanf(f, "%u%*[/]%u%*[/]%u %u%*[/]%u%*[/]%u %u%*[/]%u%*[/]%u\n", &vi[0], &ti[0], &ni[0], &vi[1], &ti[1], &ni[1], &vi[2], &ti[2], &ni[2]);
                if (got != 9) { fclose(f); return 2; }   // "error while reading faces": the reference stops reading this mesh
                D3 p[3], n[3];
                double uv[6];
                for (int k = 0; k < 3; k++) {
                    if (vi[k] == 0 || (size_t)vi[k] * 3 > v.size() || ni[k] == 0 || (size_t)ni[k] * 3 > vn.size() || ti[k] == 0 || (size_t)ti[k] * 2 > vt.size()) { fclose(f); return 2; }
                    p[k] = mk(v[(vi[k] - 1) * 3], v[(vi[k] - 1) * 3 + 1], v[(vi[k] - 1) * 3 + 2]);
                    n[k] = unit(mk(vn[(ni[k] - 1) * 3], vn[(ni[k] - 1) * 3 + 1], vn[(ni[k] - 1) * 3 + 2]));   // vertex ctor normalises, include/entities.h:311-316
                    uv[k * 2] = vt[(ti[k] - 1) * 2]; uv[k * 2 + 1] = vt[(ti[k] - 1) * 2 + 1];
                }
                push_triangle(p, n, uv, mat);
            }
        }
        fclose(f);
        return 0;
    }

    void add_box(D3 pos, D3 size, D3 rot_angles, int mat)   // boxMesh, include/entities.h:740-785
    {
        static const double c[12][3][3] = {
            {{-1, -1, -1}, {-1, 1, -1}, {1, -1, -1}}, {{-1, 1, -1}, {1, 1, -1}, {1, -1, -1}},
            {{-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}}, {{-1, -1, 1}, {-1, 1, 1}, {-1, 1, -1}},
            {{-1, -1, -1}, {1, -1, -1}, {-1, -1, 1}}, {{-1, -1, 1}, {1, -1, -1}, {1, -1, 1}},
            {{-1, -1, 1}, {1, -1, 1}, {-1, 1, 1}},    {{1, -1, 1}, {1, 1, 1}, {-1, 1, 1}},
            {{-1, 1, 1}, {1, 1, 1}, {1, 1, -1}},      {{-1, 1, 1}, {1, 1, -1}, {-1, 1, -1}},
            {{1, -1, -1}, {1, 1, -1}, {1, -1, 1}},    {{1, -1, 1}, {1, 1, -1}, {1, 1, 1}}};
        const M3 R = euler_xyz(rot_angles.x, rot_angles.y, rot_angles.z);
        const double uv[6] = {0, 0, 0, 0, 0, 0};
        const D3 zero[3] = {mk(0, 0, 0), mk(0, 0, 0), mk(0, 0, 0)};
        for (int t = 0; t < 12; t++) {
            D3 p[3];
            for (int k = 0; k < 3; k++) {
                const D3 u = unit(mk(c[t][k][0], c[t][k][1], c[t][k][2]));
                p[k] = add(mul(R, mk(u.x * size.x, u.y * size.y, u.z * size.z)), pos);
            }
            push_triangle(p, zero, uv, mat);
        }
    }

    int load_scn(const char* path)   // include/sceneLoader.cpp:12-185
    {
        const std::string full = path;
        const std::string dir = full.substr(0, full.find_last_of("/"));
        FILE* f = fopen(path, "r");
        if (!f) { err = std::string("cannot open scene file: ") + path; return -1; }
        std::vector<int> tex;   // the scene file's texture list -> indices of this scene's texture table
        const int mat_base = (int)(mats.size() / 9);
        int n_mats_here = 0;
        char word[128];
        int rc = 0;
        for (;;) {
            if (fscanf(f, "%127s", word) == EOF) break;
            if (strcmp(word, "colorTex") == 0) {
                double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                fscanf(f, "%lf %lf %lf\n", &q[0], &q[1], &q[2]);
                tex.push_back(add_texture(0, q, nullptr, 0));
            } else if (strcmp(word, "checkerboardTex") == 0) {   // include/sceneLoader.cpp:57-64
                double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int tiles = 0;
                fscanf(f, "%lf %lf %lf %lf %lf %lf %d\n", &q[0], &q[1], &q[2], &q[3], &q[4], &q[5], &tiles);
                q[6] = tiles;
                tex.push_back(add_texture(1, q, nullptr, 0));
            } else if (strcmp(word, "imTex") == 0) {             // include/sceneLoader.cpp:47-56
                char fn[100];
                int utile = 0, vtile = 0;
                fscanf(f, "%99s %d %d\n", fn, &utile, &vtile);
                int w = 0, h = 0;
                bool alpha = false;
                std::vector<uint8_t> px;
                if (!decode_png(dir + "/" + fn, w, h, alpha, px, err)) { rc = -2; break; }
                const double q[8] = {(double)utile, (double)vtile, (double)w, (double)h, alpha ? 1.0 : 0.0, 0, 0, 0};
                tex.push_back(add_texture(2, q, px.data(), (long long)px.size()));
            } else if (strcmp(word, "mat") == 0) {
                int dif = 0, em = 0;
                double r = 0, o = 0, ior = 1.0;   // reference: IOR uninitialised when the 5th number is missing
                fscanf(f, "%d %d %lf %lf %lf\n", &dif, &em, &r, &o, &ior);
                if (dif < 0 || em < 0 || dif >= (int)tex.size() || em >= (int)tex.size()) { err = "mat: texture index out of range"; rc = -2; break; }
                if (add_material_tex(tex[dif], tex[em], r, o, ior) < 0) { rc = -2; break; }
                n_mats_here++;
            } else if (strcmp(word, "multiMat") == 0) {
                char s[128];
                int length = 0;
                fscanf(f, "%127[0123456789 ]%n\n", s, &length);   // parsed and ignored: meshes take their single `mat` index
            } else if (strcmp(word, "mesh") == 0) {
                char fn[100];
                D3 pos, rot;
                int mat = 0;
                fscanf(f, "%99s %lf %lf %lf %lf %lf %lf %d\n", fn, &pos.x, &pos.y, &pos.z, &rot.x, &rot.y, &rot.z, &mat);
                if (mat < 0 || mat >= n_mats_here) { err = "mesh: material index out of range"; rc = -2; break; }
                load_obj(dir + "/" + fn, pos, rot, mat_base + mat);
            } else if (strcmp(word, "box") == 0) {
                D3 pos, size, rot;
                int mat = 0;
                fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %d\n", &pos.x, &pos.y, &pos.z, &size.x, &size.y, &size.z, &rot.x, &rot.y, &rot.z, &mat);
                if (mat < 0 || mat >= n_mats_here) { err = "box: material index out of range"; rc = -2; break; }
                add_box(pos, size, rot, mat_base + mat);
            } else if (strcmp(word, "sphere") == 0) {
                D3 pos;
                double rad = 0;
                int mat = 0;
                fscanf(f, "%lf %lf %lf %lf %d\n", &pos.x, &pos.y, &pos.z, &rad, &mat);
                if (mat < 0 || mat >= n_mats_here) { err = "sphere: material index out of range"; rc = -2; break; }
                push_sphere(pos, rad, mat_base + mat);
            } else if (strcmp(word, "heightFog") == 0) {
                double q[12] = {0};
                fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf %lf\n", &q[0], &q[1], &q[2], &q[3], &q[4], &q[5], &q[6], &q[7], &q[8], &q[9], &q[10], &q[11]);
                if (add_height_fog(q, nullptr, 0, 0x9E3779B97F4A7C15ull) != 0) { rc = -2; break; }
            } else if (strcmp(word, "light") == 0) {
                double p[3], c[3], rad = 0;
                fscanf(f, "%lf %lf %lf %lf %lf %lf %lf\n", &p[0], &p[1], &p[2], &c[0], &c[1], &c[2], &rad);
                add_light(p, c, rad);
            } else if (strcmp(word, "photons") == 0) {
                fscanf(f, "%d %d\n", &st.photons, &st.photon_depth);
            } else if (strcmp(word, "samples") == 0) {
                fscanf(f, "%d %d %lf\n", &st.min_samples, &st.max_samples, &st.noise_thresh);
            } else if (strcmp(word, "ambient") == 0) {
                fscanf(f, "%lf %lf %lf\n", &st.ambient[0], &st.ambient[1], &st.ambient[2]);
            } else if (strcmp(word, "camera") == 0) {
                double p[3], l[3];
                fscanf(f, "%lf %lf %lf %lf %lf %lf\n", &p[0], &p[1], &p[2], &l[0], &l[1], &l[2]);
                set_camera(p, l);   // Camera::setDir(lookAt - pos): same basis as the constructor
            }
        }
        fclose(f);
        return rc;
    }

    // new HeightFog(pos, size, col, density, scatter, noiseScale), include/atmosphere.h:37-47: the noise grid has
    // (sx+1)(sy+1)(sz+1) noiseScale^3 values; the reference draws them from time-seeded drand(), here they are given or come
    // from the counter RNG (splitmix64 of seed and index)
    int add_height_fog(const double* q, const double* grid, int n_grid, unsigned long long seed, bool any_size = false)
    {
        const int ns = (int)q[11];
        const double want = (q[3] + 1) * (q[4] + 1) * (q[5] + 1) * std::pow((double)ns, 3);
        int n = 0;
        for (int i = 0; i < want; i++) n++;   // the constructor's loop: for (int i = 0; i < (double)N; i++)
        if (grid && any_size) n = n_grid;     // an entity that carries its grid (the C++ HeightFog): taken as it is
        if (n <= 0) { err = "heightFog: empty noise grid"; return -2; }
        if (grid && n_grid != n) { err = "heightFog: noise grid size does not match (sx+1)(sy+1)(sz+1) scale^3"; return -2; }
        fog.insert(fog.end(), q, q + 12);
        for (int i = 0; i < n; i++) {
            if (grid) { fog_grid.push_back(grid[i]); continue; }
            unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)fog_grid.size() + 1);
            z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 27; z *= 0x94d049bb133111ebull; z ^= z >> 31;
            fog_grid.push_back((double)(z >> 11) * (1.0 / 9007199254740992.0));
        }
        fog_grid_off.push_back((int32_t)fog_grid.size());
        tree_valid = false;
        return 0;
    }

    void add_light(const double* p, const double* c, double rad)
    {
        // Light(pos, target = 0, col, rad): dir = normalize(target - pos), angle = .125 until Octree::rebuild (include/light.h:16,26-31)
        const D3 d = unit(sub(mk(0, 0, 0), mk(p[0], p[1], p[2])));
        const double L[11] = {p[0], p[1], p[2], c[0], c[1], c[2], rad, d.x, d.y, d.z, .125};
        lights.insert(lights.end(), L, L + 11);
        tree_valid = false;
    }
};

extern "C" {

gih_scene* gih_scene_create(void) { return new gih_scene(); }
void gih_scene_destroy(gih_scene* s) { delete s; }
const char* gih_last_error(const gih_scene* s) { return s ? s->err.c_str() : "null scene"; }

int gih_load_scn(gih_scene* s, const char* path)
{
    if (!s || !path) return -1;
    return s->load_scn(path);
}

int gih_load_obj(gih_scene* s, const char* path, const double* pos3, const double* rot3, int32_t mat_idx)
{
    if (!s || !path || !pos3 || !rot3) return -1;
    if (mat_idx < 0 || mat_idx >= (int)(s->mats.size() / 9)) { s->err = "load_obj: material index out of range"; return -2; }
    return s->load_obj(path, mk(pos3[0], pos3[1], pos3[2]), mk(rot3[0], rot3[1], rot3[2]), mat_idx);
}

int gih_add_material(gih_scene* s, const double* m)
{
    if (!s || !m) return -1;
    s->mats.insert(s->mats.end(), m, m + 9);
    while (s->mat_tex.size() < s->mats.size() / 9 * 2) s->mat_tex.push_back(-1);   // plain colours: no texture record
    s->tree_valid = false;
    return (int)(s->mats.size() / 9) - 1;
}

int gih_add_texture(gih_scene* s, int32_t kind, const double* params8, const uint8_t* rgba, int64_t n_bytes)
{
    if (!s) return -1;
    return s->add_texture(kind, params8, rgba, n_bytes);
}
int gih_add_material_tex(gih_scene* s, int32_t dif, int32_t em, double roughness, double opacity, double ior)
{
    if (!s) return -1;
    return s->add_material_tex(dif, em, roughness, opacity, ior);
}
int gih_load_png(const char* path, int32_t* width, int32_t* height, int32_t* has_alpha, uint8_t** rgba_out, char* errbuf, int32_t err_len)
{
    if (!path || !width || !height || !has_alpha || !rgba_out) return -1;
    int w = 0, h = 0;
    bool a = false;
    std::vector<uint8_t> px;
    std::string err;
    if (!decode_png(path, w, h, a, px, err)) {
        if (errbuf && err_len > 0) { strncpy(errbuf, err.c_str(), (size_t)err_len - 1); errbuf[err_len - 1] = 0; }
        return -2;
    }
    *width = w; *height = h; *has_alpha = a ? 1 : 0;
    *rgba_out = (uint8_t*)malloc(px.size());
    if (!*rgba_out) return -3;
    memcpy(*rgba_out, px.data(), px.size());
    return 0;
}
void gih_free(void* p) { free(p); }

int gih_add_triangles(gih_scene* s, int32_t n, const double* pos, const double* nrm, const double* uv, const int32_t* mat_idx)
{
    if (!s || n < 0 || (n && (!pos || !mat_idx))) return -1;
    const int nm = (int)(s->mats.size() / 9);
    for (int i = 0; i < n; i++) {
        if (mat_idx[i] < 0 || mat_idx[i] >= nm) { s->err = "add_triangles: material index out of range"; return -2; }
        D3 p[3], nn[3];
        double t[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 3; k++) {
            p[k] = mk(pos[(size_t)i * 9 + k * 3], pos[(size_t)i * 9 + k * 3 + 1], pos[(size_t)i * 9 + k * 3 + 2]);
            nn[k] = nrm ? mk(nrm[(size_t)i * 9 + k * 3], nrm[(size_t)i * 9 + k * 3 + 1], nrm[(size_t)i * 9 + k * 3 + 2]) : mk(0, 0, 0);
        }
        if (uv) for (int k = 0; k < 6; k++) t[k] = uv[(size_t)i * 6 + k];
        s->push_triangle(p, nn, t, mat_idx[i]);
    }
    return 0;
}

int gih_add_sphere(gih_scene* s, const double* centre3, double radius, int32_t mat_idx)
{
    if (!s || !centre3) return -1;
    if (mat_idx < 0 || mat_idx >= (int)(s->mats.size() / 9)) { s->err = "add_sphere: material index out of range"; return -2; }
    s->push_sphere(mk(centre3[0], centre3[1], centre3[2]), radius, mat_idx);
    return 0;
}

int gih_add_height_fog(gih_scene* s, const double* params12, const double* grid, int32_t n_grid, uint64_t seed)
{
    if (!s || !params12) return -1;
    return s->add_height_fog(params12, grid, n_grid, seed);
}

int gih_add_height_fog_grid(gih_scene* s, const double* params12, const double* grid, int32_t n_grid)
{
    if (!s || !params12 || !grid || n_grid <= 0) return -1;
    return s->add_height_fog(params12, grid, n_grid, 0, true);
}

int gih_add_light(gih_scene* s, const double* pos3, const double* col3, double rad)
{
    if (!s || !pos3 || !col3) return -1;
    s->add_light(pos3, col3, rad);
    return 0;
}

int gih_set_ambient(gih_scene* s, const double* rgb)
{
    if (!s || !rgb) return -1;
    for (int k = 0; k < 3; k++) s->st.ambient[k] = rgb[k];
    return 0;
}

int gih_get_settings(const gih_scene* s, gih_settings* out)
{
    if (!s || !out) return -1;
    *out = s->st;
    return 0;
}

int gih_set_camera(gih_scene* s, const double* pos3, const double* look_at3)
{
    if (!s || !pos3 || !look_at3) return -1;
    s->set_camera(pos3, look_at3);
    return 0;
}

int gih_build_octree(gih_scene* s)
{
    if (!s) return -1;
    if (s->mats.empty()) { s->err = "build_octree: no materials"; return -2; }
    return s->build_octree();
}

int gih_get_scene_desc(const gih_scene* s, gi_scene_desc* d)
{
    if (!s || !d) return -1;
    if (!s->tree_valid) return -4;
    memset(d, 0, sizeof *d);
    d->n_tri = s->n_tri();
    d->tri_pos = s->tri_pos.data(); d->tri_nrm = s->tri_nrm.data(); d->tri_uv = s->tri_uv.data(); d->tri_mat = s->tri_mat.data();
    d->n_mat = (int32_t)(s->mats.size() / 9); d->mats = s->mats.data();
    d->n_light = (int32_t)(s->lights.size() / 11); d->lights = s->lights.data();
    for (int k = 0; k < 3; k++) d->ambient[k] = s->st.ambient[k];
    d->n_node = (int32_t)(s->node_bbox.size() / 6);
    d->node_bbox = s->node_bbox.data(); d->node_child = s->node_child.data();
    d->node_ent_off = s->node_ent_off.data(); d->node_ent_idx = s->node_ent_idx.data();
    d->ent_kind = s->ent_kind.data();
    d->n_fog = (int32_t)(s->fog.size() / 12);
    d->fog = s->fog.data(); d->fog_grid_off = s->fog_grid_off.data(); d->fog_grid = s->fog_grid.data();
    if (s->textured()) {   // constant-colour scenes stay in the plain form (n_tex = 0)
        if (s->mat_tex.size() != (size_t)d->n_mat * 2) return -4;   // a material added as plain colours next to textured ones
        d->n_tex = (int32_t)s->tex_kind.size();
        d->tex_kind = s->tex_kind.data(); d->tex_param = s->tex_param.data(); d->mat_tex = s->mat_tex.data();
        d->tex_pixels = s->tex_pixels.data(); d->n_tex_pixel_bytes = (int64_t)s->tex_pixels.size();
    }
    return 0;
}

int gih_counts(const gih_scene* s, int32_t* n_tri, int32_t* n_mat, int32_t* n_light, int32_t* n_node, int32_t* n_ref)
{
    if (!s) return -1;
    if (n_tri) *n_tri = s->n_tri();
    if (n_mat) *n_mat = (int32_t)(s->mats.size() / 9);
    if (n_light) *n_light = (int32_t)(s->lights.size() / 11);
    if (n_node) *n_node = (int32_t)(s->node_bbox.size() / 6);
    if (n_ref) *n_ref = (int32_t)s->node_ent_idx.size();
    return 0;
}

int gih_build_photon_map(gih_scene* s, int32_t n, const double* photons)
{
    if (!s || n < 0 || (n && !photons)) return -1;
    if (!s->tree_valid) { s->err = "build_photon_map: build the scene octree first (the map lives in the scene's root box)"; return -4; }
    return s->build_photon_map(n, photons);
}

int gih_build_photon_map_in_box(gih_scene* s, const double* box6, int32_t n, const double* photons)
{
    if (!s || !box6 || n < 0 || (n && !photons)) return -1;
    s->root_box.lo = mk(box6[0], box6[1], box6[2]);
    s->root_box.hi = mk(box6[3], box6[4], box6[5]);
    return s->build_photon_map(n, photons);
}

int gih_get_photon_desc(const gih_scene* s, gi_photon_map_desc* d)
{
    if (!s || !d) return -1;
    memset(d, 0, sizeof *d);
    d->n_photon = (int32_t)(s->photons.size() / 9);
    d->photons = s->photons.data();
    d->n_node = (int32_t)(s->pm_bbox.size() / 6);
    d->node_bbox = s->pm_bbox.data(); d->node_child = s->pm_child.data(); d->node_off = s->pm_off.data(); d->node_idx = s->pm_idx.data();
    return 0;
}

// ---- single-object forms of what the tree builder and the loader do, for the C++ entity classes (include/gi/entities.h, atmosphere.h)
int gih_entity_bbox(int32_t kind, const double* pos9, double* out6)
{
    if (!pos9 || !out6 || (kind != 0 && kind != 1)) return -1;
    const D3 v[3] = {mk(pos9[0], pos9[1], pos9[2]), mk(pos9[3], pos9[4], pos9[5]), mk(pos9[6], pos9[7], pos9[8])};
    const Aabb b = entity_box(kind, v);
    out6[0] = b.lo.x; out6[1] = b.lo.y; out6[2] = b.lo.z; out6[3] = b.hi.x; out6[4] = b.hi.y; out6[5] = b.hi.z;
    return 0;
}

int gih_entity_overlaps_box(int32_t kind, const double* pos9, const double* box6)
{
    if (!pos9 || !box6 || (kind != 0 && kind != 1)) return -1;
    const D3 v[3] = {mk(pos9[0], pos9[1], pos9[2]), mk(pos9[3], pos9[4], pos9[5]), mk(pos9[6], pos9[7], pos9[8])};
    Aabb cell;
    cell.lo = mk(box6[0], box6[1], box6[2]); cell.hi = mk(box6[3], box6[4], box6[5]);
    return entity_in_cell(kind, v, cell) ? 1 : 0;
}

int gih_box_mesh(const double* pos3, const double* size3, const double* rot3, double* out108)
{
    if (!pos3 || !size3 || !rot3 || !out108) return -1;
    gih_scene s;
    const double m[9] = {1, 1, 1, 0, 0, 0, 0, 0, 0};
    s.mats.assign(m, m + 9);
    s.add_box(mk(pos3[0], pos3[1], pos3[2]), mk(size3[0], size3[1], size3[2]), mk(rot3[0], rot3[1], rot3[2]), 0);
    memcpy(out108, s.tri_pos.data(), 108 * sizeof(double));
    return 12;
}

int gih_fog_grid(const double* params12, uint64_t seed, double* grid, int32_t n_grid)
{
    if (!params12) return -1;
    gih_scene s;
    if (s.add_height_fog(params12, nullptr, 0, seed) != 0) return -2;
    const int n = (int)s.fog_grid.size();
    if (grid) { if (n_grid != n) return -2; memcpy(grid, s.fog_grid.data(), (size_t)n * sizeof(double)); }
    return n;
}

// gamma(color, 2.2), glm::clamp(color, 0, 1), Image::setPixel's (int)(255 c) (include/raytracer.h:150-157, include/util.h:94-97, include/image.h:14-16).
// pow of a negative channel is NaN, which the reference's clamp lets through and its cast turns into 0 (known answers: tests/golden/kat.npz kat_pixel).
int gih_to_rgb8(const void* lin, int32_t is_f64, int64_t n_values, uint8_t* out)
{
    if (n_values < 0 || (n_values && (!lin || !out))) return -1;
    for (int64_t i = 0; i < n_values; i++) {
        const double c = is_f64 ? static_cast<const double*>(lin)[i] : (double)static_cast<const float*>(lin)[i];
        const double g = std::pow(c, 1.0 / 2.2);
        out[i] = g != g ? (uint8_t)0 : (uint8_t)(int)(255 * std::min(std::max(g, 0.0), 1.0));
    }
    return 0;
}

}  // extern "C"
