// include/gi/material.h -- mirrors include/material.h:10-100 of the reference: texture (constant colour), checkerboard, imageTexture
// and Material.  get() / getAlpha() keep the reference's arithmetic for host-side callers; the device evaluates the same functions
// from the tables Octree::rebuild registers (gi_register -> gih_add_texture).
#pragma once
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../gi_raytracer_amd/csrc/gi_host.h"
#include "vec.h"
struct texture {
    texture(gi::dvec3 col) : color(col) {}
    virtual ~texture() {}
    virtual gi::dvec3 get(gi::dvec2&) { return color; }
    virtual double getAlpha(gi::dvec2&) { return 1; }
    // hands this texture to the flattened scene and returns its index there
    virtual int gi_register(gih_scene* s) const
    {
        const double q[8] = {color.x, color.y, color.z, 0, 0, 0, 0, 0};
        return gih_add_texture(s, 0, q, nullptr, 0);
    }
    gi::dvec3 color;
};
// Procedural checkerboard texture (include/material.h:32-50)
struct checkerboard : texture {
    gi::dvec3 a, b;
    checkerboard(int t, gi::dvec3 col0, gi::dvec3 col1) : texture(gi::dvec3(0, 0, 0)), a(col0), b(col1), tiles(t) {}
    gi::dvec3 get(gi::dvec2& uv) override
    {
        if (((int)(uv.x * tiles) % 2 == 0) ^ ((int)(uv.y * tiles) % 2 == 0)) return a;
        return b;
    }
    int gi_register(gih_scene* s) const override
    {
        const double q[8] = {a.x, a.y, a.z, b.x, b.y, b.z, (double)tiles, 0};
        return gih_add_texture(s, 1, q, nullptr, 0);
    }
    int tiles;
};
// Texture from an image file (include/material.h:52-81); 8-bit PNG, decoded as QImage presents it
struct imageTexture : texture {
    std::string fname;
    gi::dvec2 tile;
    int width = 0, height = 0;
    bool has_alpha = false;
    std::vector<unsigned char> rgba;   // rows top to bottom
    imageTexture(const char* name, gi::dvec2 t) : texture(gi::dvec3(0, 0, 0)), fname(name), tile(t)
    {
        int32_t w = 0, h = 0, a = 0;
        uint8_t* px = nullptr;
        char err[256] = {0};
        if (gih_load_png(name, &w, &h, &a, &px, err, (int32_t)sizeof err) != 0) throw std::runtime_error(std::string("imageTexture: ") + err);
        width = w; height = h; has_alpha = a != 0;
        rgba.assign(px, px + (size_t)w * h * 4);
        gih_free(px);
    }
    // the same from decoded pixels (RGBA8, rows top to bottom)
    imageTexture(int w, int h, bool alpha, const unsigned char* px, gi::dvec2 t) : texture(gi::dvec3(0, 0, 0)), fname(""), tile(t), width(w), height(h), has_alpha(alpha), rgba(px, px + (size_t)w * h * 4) {}
    const unsigned char* pixel(const gi::dvec2& uv) const
    {
        const int x = std::abs((int)(uv.x * width * tile.x) % width);
        const int y = height - std::abs((int)(uv.y * height * tile.y) % height) - 1;
        return &rgba[((size_t)y * width + x) * 4];
    }
    gi::dvec3 get(gi::dvec2& uv) override
    {
        const unsigned char* p = pixel(uv);
        const double g = 1.0 / 2.2;   // gamma(col, 1.0/GAMMA), include/util.h:31,94-97
        return gi::dvec3(std::pow(p[0] / 255.0, 1.0 / g), std::pow(p[1] / 255.0, 1.0 / g), std::pow(p[2] / 255.0, 1.0 / g));
    }
    double getAlpha(gi::dvec2& uv) override { return has_alpha ? pixel(uv)[3] / 255.0 : 1; }
    int gi_register(gih_scene* s) const override
    {
        const double q[8] = {tile.x, tile.y, (double)width, (double)height, has_alpha ? 1.0 : 0.0, 0, 0, 0};
        return gih_add_texture(s, 2, q, rgba.data(), (int64_t)rgba.size());
    }
};
struct Material {
    Material(texture* dif, texture* em, double r, double o, double i = 1) : diffuse(dif), emissive(em), roughness(r), opacity(o), IOR(i) {}
    double getAlpha(gi::dvec2& uv) { return opacity * diffuse->getAlpha(uv); }
    texture* diffuse;
    texture* emissive;
    double roughness, opacity, IOR;
};
