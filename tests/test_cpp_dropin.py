"""The reference's main.cpp flow compiled against the drop-in C++ headers (include/gi/*.h) and linked with the C-ABI library."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_dropin")


def build():
    lib = os.path.join(ROOT, "gi_raytracer_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"), "-L" + lib, "-lgi_raytracer_hip",
                    "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", EXE], check=True)


def test_host_side_of_the_dropin_api():
    build()
    out = subprocess.run([EXE, os.path.join(ROOT, "scenes/caustics/caustics.scn")], check=True, capture_output=True, text=True).stdout
    assert "lights 1 photons 750000 samples 8..32" in out           # what scenes/caustics/caustics.scn sets
    assert "light0 angle 0.079562916434" in out                     # Octree::rebuild's light cone, as the reference dump has it
    assert "quad valid 1" in out


@pytest.mark.gpu
def test_dropin_render_matches_python_path():
    import gi_raytracer_amd as gi
    import parity_checks as pc
    build()
    out = subprocess.run([EXE, os.path.join(ROOT, "scenes/caustics/caustics.scn"), "gpu"], check=True, capture_output=True, text=True).stdout
    m = re.search(r"linear mean ([0-9.eE+-]+)", out)
    assert m, out
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    rt.tracePhotons(2000)
    img = rt.run(64, 36, f64=False, min_samples=4, max_samples=4)
    assert abs(float(m.group(1)) - float(img.astype(np.float64).mean())) < 1e-9
