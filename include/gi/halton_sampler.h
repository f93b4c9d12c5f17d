// include/gi/halton_sampler.h -- Halton_sampler of the reference (include/halton_sampler.h:573-603,626-900,1417-3286) for callers that pass one to
// RayTracer::radiance / tracePhotons or draw from it: init_faure() builds the Faure-permuted digit-group tables by rule (the same tables
// the kernels read, gi_layout.h: build_halton_tables), sample(dim, index) is the kernels' halton_sample on them -- bit-exact floats for all
// 256 dimensions (tests/golden/halton.npz).  init_random() is not provided (the renderer uses the Faure tables only).
#pragma once
#include <vector>
#include "detail.h"
#include "../../gi_raytracer_amd/csrc/gi_layout.h"
class Halton_sampler {
  public:
    void init_faure()
    {
        gi::build_halton_tables(_dims, _table);
        _S = gi::Scene{};
        _S.hdims = _dims.data();
        _S.htable = _table.data();
    }
    float sample(unsigned dimension, unsigned index) const { return dimension < 256 ? gi::halton_sample(_S, dimension, index) : 0.f; }
  private:
    std::vector<gi::HaltonDim> _dims;
    std::vector<uint16_t> _table;
    gi::Scene _S{};
};
