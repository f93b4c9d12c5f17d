#!/usr/bin/env python3
"""Tuning aid: lane utilisation of the wide walk, per loop level, in k_st_trace and k_st_shade.

Needs a library built with -DGI_EXP_DIV=1 (tools/build_exp.sh libgi_div -DGI_EXP_DIV=1) and GI_LIB_PATH=exp/libgi_div.so.
For each step kind (node descent, leaf, triangle test on the scalar path, triangle test on the per-lane path) the kernels count how
often a lane executed it and how often a wave did; lanes / (64 * waves) is the share of the issued work that was useful.
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gi_raytracer_amd as gi
from bench import SCN

def main():
    lib = ctypes.CDLL(gi.LIB_PATH)
    lib.gi_debug_div.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 32)()
    for name, w, h, spp, photons in (("caustics", 1920, 1080, 16, 200000), ("cornell", 512, 512, 16, 0), ("teapot", 1920, 1080, 8, 200000)):
        scene = gi.Scene.load(os.path.join(ROOT, SCN[name])).rebuild()
        rt = gi.RayTracer(0).setScene(scene)
        if photons:
            rt.tracePhotonsOnDevice(photons)
        lib.gi_debug_div(out, 1)
        rt.run(w, h, min_samples=spp, max_samples=spp)
        lib.gi_debug_div(out, 1)
        for k, kern in enumerate(("k_st_trace", "k_st_shade")):
            v = out[k * 16:(k + 1) * 16]
            row = []
            for j, what in enumerate(("node", "leaf", "tri_scalar", "tri_lane")):
                lanes, waves = v[2 * j], v[2 * j + 1]
                row.append("%s %.3g lane-steps, util %.2f" % (what, lanes, lanes / (64.0 * waves) if waves else 0.0))
            print(name, kern, "threads", v[8], "|", " | ".join(row), flush=True)

if __name__ == "__main__":
    main()
