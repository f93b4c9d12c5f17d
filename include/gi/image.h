// include/gi/image.h -- the frame sink of the reference (include/image.h:7-29): RGB888 store written with the truncating (int)(255 c).
// With -DGI_USE_QT the store is a QImage named `_image` with Viewer as a friend, exactly what include/viewer.h:43 reads, so the reference's
// GUI compiles against this header unchanged; without Qt it is a plain byte array with the same methods (plus headless PPM / PFM writers).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "vec.h"
#ifdef GI_USE_QT
#include <QImage>
#include <QColor>
struct Image {
    Image() = delete;
    Image(int width, int height) : _image(width, height, QImage::Format_RGB888) { this->clear(); }
    int width() const { return _image.width(); }
    int height() const { return _image.height(); }
    void setPixel(int x, int y, gi::dvec3 c) { _image.setPixel(x, y, QColor((int)(255 * c.x), (int)(255 * c.y), (int)(255 * c.z)).rgb()); }
    void setPixel8(int x, int y, const uint8_t* rgb) { _image.setPixel(x, y, qRgb(rgb[0], rgb[1], rgb[2])); }
    gi::dvec3 getPixel(int x, int y) const { const QRgb p = _image.pixel(x, y); return gi::dvec3(qRed(p) / 255., qGreen(p) / 255., qBlue(p) / 255.); }
    void clear() { _image.fill(Qt::black); }
  private:
    QImage _image;
    friend class Viewer;
};
#else
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
    void setPixel8(int x, int y, const uint8_t* rgb) { uint8_t* p = &_rgb[((size_t)y * _w + x) * 3]; p[0] = rgb[0]; p[1] = rgb[1]; p[2] = rgb[2]; }
    gi::dvec3 getPixel(int x, int y) const
    {
        const uint8_t* p = &_rgb[((size_t)y * _w + x) * 3];
        return gi::dvec3(p[0] / 255., p[1] / 255., p[2] / 255.);
    }
    void clear() { std::fill(_rgb.begin(), _rgb.end(), 0); }
    const uint8_t* rgb() const { return _rgb.data(); }
    // headless output: binary PPM of the 8-bit frame the Viewer would paint
    bool save_ppm(const char* path) const
    {
        FILE* f = fopen(path, "wb");
        if (!f) return false;
        fprintf(f, "P6\n%d %d\n255\n", _w, _h);
        const bool ok = fwrite(_rgb.data(), 1, _rgb.size(), f) == _rgb.size();
        return fclose(f) == 0 && ok;
    }
  private:
    int _w, _h;
    std::vector<uint8_t> _rgb;
};
#endif
// linear (pre-gamma, unclamped) radiance as a little-endian PFM, rows bottom to top as the format wants them
inline bool gi_save_pfm(const char* path, const float* lin_rgb, int w, int h)
{
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    bool ok = true;
    for (int y = h - 1; y >= 0 && ok; y--) ok = fwrite(lin_rgb + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, f) == (size_t)w * 3;
    return fclose(f) == 0 && ok;
}
