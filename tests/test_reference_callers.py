"""The drop-in boundary checked from the reference's side (build container only: needs /root/reference, which never travels).

The reference's own, unmodified include/sceneLoader.cpp and include/meshLoader.cpp -- the callers a maintainer keeps -- are compiled from
where they lie against the headers of include/gi/ (symbolic links in a scratch directory make the compiler resolve their #include "..."
lines through -I include/gi instead of the reference's own directory; no reference text enters the repository), linked with the C-ABI
library, and run through the main.cpp flow on the reference's own scenes.  What Octree::rebuild then holds for the GPU must equal, bit
for bit, what the reference's loader + Octree::rebuild produced (tests/golden/scene_*.npz); the host-side queries of the drop-in classes
(Octree::_root, intersectSorted, intersect, PhotonMap::rebuild / getInRange) must answer as the reference's did."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include")), reason="no /root/reference here (GPU box): the reference's callers cannot be compiled")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("refcallers")
    for f in ("sceneLoader.cpp", "meshLoader.cpp", "sceneLoader.h", "meshLoader.h"):
        os.symlink(os.path.join(REF, "include", f), d / f)
    exe = d / "driver"
    lib = os.path.join(ROOT, "gi_raytracer_amd")
    subprocess.run(["g++", "-std=c++14", "-O1", "-w", "-DGI_USE_GLM", "-I", str(d), "-I", os.path.join(ROOT, "include", "gi"), "-I", os.path.join(REF, "3rd_party"),
                    os.path.join(ROOT, "tests", "cpp", "ref_callers_driver.cpp"), str(d / "sceneLoader.cpp"), str(d / "meshLoader.cpp"),
                    "-L" + lib, "-lgi_raytracer_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("name,scn", [("cornell", "scenes/cornell/test.scn"), ("caustics", "scenes/caustics/caustics.scn"), ("test_scene", "examples/test_scene/test.scn")])
def test_reference_loaders_on_dropin_headers(driver, golden, tmp_path, name, scn):
    fx = golden("scene_" + name)
    rays = fx["rays"][fx["leaforder_ray"]]
    rays.astype("<f8").tofile(tmp_path / "rays.f64")
    fx["shadow_q"].astype("<f8").tofile(tmp_path / "shadow.f64")
    if "photons" in fx:
        fx["photons"].astype("<f8").tofile(tmp_path / "photons.f64")
        fx["gather_q"].astype("<f8").tofile(tmp_path / "gather_q.f64")
    out = subprocess.run([driver, os.path.join(REF, scn), str(tmp_path)], check=True, capture_output=True, text=True, cwd=REF).stdout
    assert "ok:" in out, out
    rd = lambda f, dt: np.fromfile(tmp_path / f, dt)
    n = len(fx["tri_pos"])
    assert np.array_equal(rd("tri_pos.f64", "<f8").reshape(n, 3, 3), fx["tri_pos"])
    assert np.array_equal(rd("tri_nrm.f64", "<f8").reshape(n, 3, 3), fx["tri_nrm"])
    assert np.array_equal(rd("tri_uv.f64", "<f8").reshape(n, 3, 2), fx["tri_uv"])
    # materials: the drop-in Material takes IOR as given; a `mat` line without it leaves the reference's own loader variable uninitialised
    # (the fixture recorded what that run happened to read), so the IOR column is compared only where the scene file gives one
    got, ref = rd("tri_mat.f64", "<f8").reshape(n, 9), fx["tri_mat"]
    assert np.array_equal(got[:, [0, 1, 3, 4, 5, 6, 7, 8]], ref[:, [0, 1, 3, 4, 5, 6, 7, 8]])
    if len(fx["lights"]):
        assert np.array_equal(rd("lights.f64", "<f8").reshape(-1, 11), fx["lights"])       # incl. dir / angle from Octree::rebuild
    assert np.array_equal(rd("oct_bbox.f64", "<f8").reshape(-1, 6), fx["oct_bbox"])
    assert np.array_equal(rd("oct_child.i32", "<i4").reshape(-1, 8), fx["oct_child"])
    assert np.array_equal(rd("oct_ent_off.i32", "<i4"), fx["oct_ent_off"]) and np.array_equal(rd("oct_ent_idx.i32", "<i4"), fx["oct_ent_idx"])
    # _root mirrors the same tree
    assert np.array_equal(rd("root_bbox.f64", "<f8").reshape(-1, 6), fx["oct_bbox"])
    assert np.array_equal(rd("root_nent.i32", "<i4"), np.diff(fx["oct_ent_off"]))
    # Octree::intersectSorted: node sequence and entry distances, exactly
    assert np.array_equal(rd("leaf_off.i32", "<i4"), fx["leaforder_off"])
    assert np.array_equal(rd("leaf_node.i32", "<i4"), fx["leaforder_node"]) and np.array_equal(rd("leaf_t0.f64", "<f8"), fx["leaforder_t0"])
    assert np.array_equal(rd("shadow_ncand.i32", "<i4"), fx["shadow_ncand"])
    if "photons" in fx:
        assert np.array_equal(rd("gather_ncand.i32", "<i4"), fx["gather_ncand"])
        assert np.array_equal(rd("pm_bbox.f64", "<f8").reshape(-1, 6), fx["pm_bbox"])


QT = "/opt/conda/include/qt"


@pytest.mark.skipif(not os.path.isdir(QT + "/QtWidgets"), reason="no Qt 5 headers in this container")
def test_reference_gui_compiles_links_and_starts_on_dropin_headers(tmp_path):
    """The reference's unmodified main.cpp, gui.h and viewer.h (QMainWindow + Viewer with its 32 ms repaint timer and worker thread) and its two
    loaders, compiled against include/gi/ with -DGI_USE_QT (Image then wraps a QImage `_image` with Viewer as friend, include/image.h:7-29), linked
    with the C-ABI library and conda's Qt 5.9.7, started on the offscreen platform: the scene loads, the window is built, the Viewer's worker
    thread calls RayTracer::run -- which, in this container without a GPU, reports that there is no device (there is no CPU renderer to fall back to)."""
    for f in ("gui.h", "viewer.h", "sceneLoader.cpp", "meshLoader.cpp", "sceneLoader.h", "meshLoader.h"):
        os.symlink(os.path.join(REF, "include", f), tmp_path / f)
    os.symlink(os.path.join(REF, "main.cpp"), tmp_path / "main.cpp")
    lib = os.path.join(ROOT, "gi_raytracer_amd")
    exe = tmp_path / "global-illu"
    subprocess.run(["g++", "-std=c++14", "-O1", "-w", "-fPIC", "-DGI_USE_GLM", "-DGI_USE_QT", "-I", str(tmp_path), "-I", os.path.join(ROOT, "include", "gi"), "-I", os.path.join(REF, "3rd_party"),
                    "-I", QT, "-I", QT + "/QtWidgets", "-I", QT + "/QtGui", "-I", QT + "/QtCore", str(tmp_path / "main.cpp"), str(tmp_path / "sceneLoader.cpp"), str(tmp_path / "meshLoader.cpp"),
                    "/opt/conda/lib/libQt5Widgets.so.5", "/opt/conda/lib/libQt5Gui.so.5", "/opt/conda/lib/libQt5Core.so.5", "-L" + lib, "-lgi_raytracer_hip",
                    # the system libstdc++ first (the ROCm runtime needs it; conda ships an older one next to Qt)
                    "-Wl,-rpath,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/conda/lib", "-Wl,-rpath,/opt/conda/lib",
                    "-lpthread", "-o", str(exe)], check=True)
    import torch
    env = dict(os.environ, QT_QPA_PLATFORM="offscreen", QT_PLUGIN_PATH="/opt/conda/plugins")
    try:
        r = subprocess.run([str(exe), "scenes/caustics/caustics.scn"], cwd=REF, env=env, capture_output=True, text=True, timeout=15)
        out = r.stdout + r.stderr
    except subprocess.TimeoutExpired as e:      # the event loop runs until the window is closed: expected
        out = (e.stdout or b"").decode() + (e.stderr or b"").decode()
    # (the loader's own messages go to a block-buffered stdout and are lost when the event loop is killed; stderr is not)
    if not torch.cuda.is_available():
        assert "no usable HIP device" in out, out    # main -> loadScene -> Gui -> Viewer -> worker thread -> RayTracer::run reached the C ABI
