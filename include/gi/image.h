// include/gi/image.h -- Qt-free mirror of include/image.h:7-29: RGB888 store with the reference's truncating (int)(255 c);
// `rgb()` hands the bytes to a QImage (INTEGRATION.md shows the two-line adapter the Viewer needs).
#pragma once
#include <cstdint>
#include <vector>
#include "vec.h"
struct Image {
    Image() = delete;
    Image(int width, int height) : _w(width), _h(height), _rgb((size_t)width * height * 3, 0) {}
    int width() const { return _w; }
    int height() const { return _h; }
    void setPixel(int x, int y, gi::dvec3 c)
    {
        uint8_t* p = &_rgb[((size_t)y * _w + x) * 3];
        p[0] = (uint8_t)(int)(255 * c.x); p[1] = (uint8_t)(int)(255 * c.y); p[2] = (uint8_t)(int)(255 * c.z);
    }
    gi::dvec3 getPixel(int x, int y) const
    {
        const uint8_t* p = &_rgb[((size_t)y * _w + x) * 3];
        return gi::dvec3(p[0] / 255., p[1] / 255., p[2] / 255.);
    }
    void clear() { std::fill(_rgb.begin(), _rgb.end(), 0); }
    const uint8_t* rgb() const { return _rgb.data(); }
  private:
    int _w, _h;
    std::vector<uint8_t> _rgb;
};
