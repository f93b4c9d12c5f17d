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
    # Light's table of 250 directions (include/light.h:17-40, include/util.cpp:129-155: Hammersley point -> direction) and the points picked from it
    m = re.search(r"light0 points 250 p125 (\S+) (\S+) (\S+) getPoint (\S+) (\S+) rad (\S+)", out)
    assert m, out
    x = float(np.float32(125) / np.float32(250))
    y = float(np.float32(float(np.float32(int(format(125, "032b")[::-1], 2))) * 2.3283064365386963e-10))
    th = np.arccos(2 * y - 1)
    want = np.array([np.sin(th) * np.cos(2 * x * np.pi), np.sin(th) * np.sin(2 * x * np.pi), np.cos(th)])
    assert np.allclose([float(m.group(k)) for k in (1, 2, 3)], want, rtol=0, atol=1e-15)
    assert abs(float(m.group(4)) - float(m.group(6))) < 1e-11 and abs(float(m.group(5)) - float(m.group(6))) < 1e-11   # on the light's sphere


def test_dropin_texture_classes_match_the_reference(golden):
    """checkerboard / imageTexture of include/gi/material.h: the PNG decoded as the reference's QImage saw it, get() / getAlpha() as its
    texture::get did (fixture scene_textures: tex 2 = cutout_binary.png tiled 2 x 1)."""
    build()
    out = subprocess.run([EXE, os.path.join(ROOT, "scenes/caustics/caustics.scn"), "host", os.path.join(ROOT, "scenes/textures/cutout_binary.png")],
                         check=True, capture_output=True, text=True).stdout
    m = re.search(r"image (\d+)x(\d+) alpha (\d) get\(0.3,0.4\) ([0-9.eE+-]+) ([0-9.eE+-]+) ([0-9.eE+-]+) a ([0-9.eE+-]+)", out)
    assert m, out
    fx = golden("scene_textures")
    t = int(np.flatnonzero((fx["tex_kind"] == 2) & (fx["tex_param"][:, 4] == 1))[0])
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (int(fx["tex_param"][t, 2]), int(fx["tex_param"][t, 3]), 1)
    import oracle_lib as ol
    ref = ol.Oracle.from_fixture(fx).tex_eval(t, [[0.3, 0.4]])[0]
    got = np.array([float(m.group(k)) for k in (4, 5, 6, 7)])
    assert np.allclose(got, ref, rtol=0, atol=1e-11) and "quad valid 1" in out


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["gpu", "gpu2"])   # gpu2: RayTracer::devices = {0, 0, 0} -- run() goes through gi_group_* (one host thread and context per entry)
def test_dropin_render_matches_python_path(devices):
    import gi_raytracer_amd as gi
    import parity_checks as pc
    build()
    out = subprocess.run([EXE, os.path.join(ROOT, "scenes/caustics/caustics.scn"), devices], check=True, capture_output=True, text=True).stdout
    m = re.search(r"linear mean ([0-9.eE+-]+)", out)
    assert m, out
    scene = gi.Scene.load(os.path.join(ROOT, "scenes/caustics/caustics.scn"))
    scene.set_ambient((0.05, 0.06, 0.07))          # as the C++ program sets RayTracer::ambient
    rt = gi.RayTracer(0).setScene(scene.rebuild())
    rt.tracePhotons(2000)
    img = rt.run(64, 36, f64=False, min_samples=4, max_samples=4)
    assert abs(float(m.group(1)) - float(img.astype(np.float64).mean())) < 1e-9
    # a second run() after a scene edit re-uploads the scene but keeps the photon map (no second emission)
    m = re.search(r"photons before (\d+) after (\d+) valid 1", out)
    assert m and m.group(1) == m.group(2) and int(m.group(1)) > 0, out
    # progressive display: the shared image fills top to bottom, stripe by stripe; stop() ends run() within one stripe
    m = re.search(r"progressive: monotone (\d) partial_seen (\d) stopped at row (\d+) of 512 \(filled (\d+)\) in ([0-9.]+) ms running 0", out)
    assert m, out
    assert m.group(1) == "1" and m.group(2) == "1"
    assert 96 <= int(m.group(3)) < 512 and int(m.group(4)) <= int(m.group(3))          # it did stop before the frame was complete
    assert float(m.group(5)) < 2000.0                                                  # the cancel flag is polled once per pass of the pipeline (generous bound)
    # the stripes grow while a step is shorter than RayTracer::progressive_ms: a whole 1080p frame at the reference's default 8..32 samples,
    # shown progressively, costs a fraction of the same frame in 16-row steps (every call has a floor: the chain of its deepest path)
    g = re.search(r"frame 1920x1080 growing stripes: ([0-9.]+) ms", out)
    f = re.search(r"frame 1920x1080 16-row stripes: ([0-9.]+) ms", out)
    assert g and f, out
    assert float(g.group(1)) < 0.6 * float(f.group(1)), out
