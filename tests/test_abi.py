"""The C-ABI shared library loads and exports every symbol the headers declare (no compute calls here: no GPU)."""
import ctypes as C
import os
import re

import pytest

import gi_raytracer_amd as gi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(gih?_[a-z0-9_]+)\s*\(", txt))


def test_library_exports_every_declared_symbol():
    L = gi.lib()
    names = declared("include/gi_hip.h") | declared("gi_raytracer_amd/csrc/gi_host.h")
    assert names == set(gi.ABI_SYMBOLS)
    for n in names:
        assert hasattr(L, n), n


def test_abi_signatures_carry_no_framework_types():
    txt = open(os.path.join(ROOT, "include/gi_hip.h")).read()
    assert "torch" not in txt and "std::" not in txt and "hip/" not in txt   # plain pointers and sizes only


def test_struct_sizes_match_the_header_layout():
    assert C.sizeof(gi.RenderParams) == 9 * 8 + 2 * 8 + 5 * 4 + 2 * 4 + 4 + 8 + 8   # incl. 4 bytes of padding before noise_thresh
    assert C.sizeof(gi.SceneDesc) % 8 == 0 and C.sizeof(gi.PhotonMapDesc) % 8 == 0


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gi.GiError):
        gi.RayTracer()


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gi_raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "gi_oracle" not in src and "oracle_lib" not in src and "libgi_emul" not in src, f
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dirpath, f)).read().lower(), f
