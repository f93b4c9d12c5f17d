// include/gi/halton_enum.h -- Halton_enum of the reference (include/halton_enum.h:69-155): the per-pixel enumeration of the (2, 3) Halton
// sequence.  Same members (m_p2, m_p3, m_x, m_y, m_scale_x, m_scale_y, m_increment) and methods; the arithmetic is the kernels' halton_index.
#pragma once
#include "detail.h"
#include "../../gi_raytracer_amd/csrc/gi_layout.h"
class Halton_enum {
  public:
    Halton_enum(unsigned width, unsigned height)
    {
        _e = gi::make_halton_enum(width, height);
        m_p2 = _e.p2; m_p3 = _e.p3; m_x = _e.m_x; m_y = _e.m_y; m_increment = _e.inc; m_scale_x = _e.scale_x; m_scale_y = _e.scale_y;
    }
    unsigned get_max_samples_per_pixel() const { return ~0u / m_increment; }
    unsigned get_index(unsigned i, unsigned x, unsigned y) const { return gi::halton_index(_e, i, x, y); }   // wraps past the last addressable sample, as the reference's 32-bit result does
    float scale_x(float x) const { return x * m_scale_x; }
    float scale_y(float y) const { return y * m_scale_y; }
    unsigned m_p2, m_p3, m_x, m_y;
    float m_scale_x, m_scale_y;
    unsigned m_increment;
  private:
    gi::HaltonEnumD _e;
};
