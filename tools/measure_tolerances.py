#!/usr/bin/env python3
"""Measured GPU-vs-oracle differences on the refractive / textured scenes (the numbers behind the bounds in tests/test_gpu_parity.py).
Prints one JSON line per case: frame RMSE, share of pixels differing by > 1e-9, largest difference, adaptive sample-count mismatches."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gi_raytracer_amd as gi   # noqa: E402
import parity_checks as pc      # noqa: E402


def case(name, scene, w, h, spp, photons, adaptive):
    rt = gi.RayTracer(0).setScene(scene)
    o = pc.oracle_for(scene)
    if photons > 0 and scene.desc().n_light > 0:
        ph, _ = rt.tracePhotons(photons)
        o.set_photons(ph)
    o.build_photon_map()
    if adaptive:
        img, nspp = rt.run(w, h, min_samples=spp, max_samples=4 * spp, noise_thresh=0.0015, want_spp=True)
        ref = o.render(w, h, spp, 4 * spp, 0.0015)
        mism = float((nspp != ref["spp"]).mean())
    else:
        img = rt.run(w, h, min_samples=spp, max_samples=spp)
        ref = o.render(w, h, spp)
        mism = 0.0
    d = np.abs(img - ref["lin"])
    print(json.dumps({"case": name, "adaptive": adaptive, "rmse": float(np.sqrt((d ** 2).mean())), "pix_gt_1e-9": float((d.max(axis=2) > 1e-9).mean()),
                      "max": float(d.max()), "spp_mismatch": mism, "mean": float(img.mean())}), flush=True)


def main():
    import big_scene
    for name, (w, h, spp, ph) in {"textures": (80, 60, 8, 1500), "textures_opaque": (96, 54, 8, 5000), "caustics_02": (96, 54, 8, 5000), "cornell_tex": (96, 54, 8, 5000),
                                  "teapot": (96, 54, 8, 5000)}.items():
        for ad in (False, True):
            case(name, pc.load_scene(name), w, h, spp, ph, ad)
    case("big_scene", big_scene.build(40, 80, textured=True), 96, 54, 8, 5000, False)


if __name__ == "__main__":
    main()
