"""Headless renderer: what the reference's main.cpp + Viewer do, without Qt.

    python -m gi_raytracer_amd scene.scn -o out.ppm [--pfm out.pfm] [--width 1000 --height 1000] [--samples MIN MAX [THRESH]] [--photons N]

The scene file's own `samples` / `photons` / `camera` lines apply unless overridden, exactly as loadScene sets RayTracer's fields
(include/sceneLoader.cpp:160-179); the frame size defaults to the reference window, 1000 x 1000 (main.cpp:43).
"""
import argparse
import sys
import time

import gi_raytracer_amd as gi


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m gi_raytracer_amd", description=__doc__.split("\n")[0])
    ap.add_argument("scene")
    ap.add_argument("-o", "--output", default="out.ppm", help="8-bit PPM of the display transform (gamma 2.2, clamp)")
    ap.add_argument("--pfm", default=None, help="also write the linear radiance as PFM")
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--height", type=int, default=1000)
    ap.add_argument("--samples", type=float, nargs="+", default=None, metavar="N", help="min max [noise threshold]")
    ap.add_argument("--photons", type=int, default=None)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    scene = gi.Scene.load(a.scene).rebuild()
    rt = gi.RayTracer(a.device).setScene(scene)          # raises without a GPU: there is no CPU path
    if a.samples:
        rt.min_samples = int(a.samples[0])
        rt.max_samples = int(a.samples[1]) if len(a.samples) > 1 else int(a.samples[0])
        if len(a.samples) > 2:
            rt.noise_thresh = float(a.samples[2])
    n_photons = rt.photons if a.photons is None else a.photons
    t0 = time.time()
    stored = 0
    if n_photons > 0 and scene.desc().n_light > 0:
        stored = len(rt.tracePhotons(n_photons)[0])
    t1 = time.time()
    lin, spp = rt.run(a.width, a.height, f64=False, want_spp=True)
    t2 = time.time()
    gi.save_ppm(a.output, lin)
    if a.pfm:
        gi.save_pfm(a.pfm, lin)
    n = int(spp.sum())
    print(f"{a.scene}: {a.width}x{a.height}, {n} samples (mean {n / (a.width * a.height):.1f} spp), {stored} photons stored; "
          f"photon pass {t1 - t0:.2f} s, frame {t2 - t1:.2f} s ({n / max(t2 - t1, 1e-9) / 1e6:.1f} Msamples/s incl. host copies) -> {a.output}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
