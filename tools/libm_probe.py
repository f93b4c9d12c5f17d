#!/usr/bin/env python3
"""How often the device libm (OCML) and the host libm (glibc, what the reference ran on) disagree in the last bit, per function and
over the argument ranges the path uses.  Device values through gi_kat (include/gi_hip.h)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gi_raytracer_amd as gi   # noqa: E402

libm = C.CDLL("libm.so.6")
for f in ("sin", "cos", "acos", "asin"):
    getattr(libm, f).restype = C.c_double; getattr(libm, f).argtypes = [C.c_double]
for f in ("atan2", "pow"):
    getattr(libm, f).restype = C.c_double; getattr(libm, f).argtypes = [C.c_double, C.c_double]


def main():
    rt = gi.RayTracer(0)
    rs = np.random.RandomState(3)
    n = 200000
    cases = {
        "sin": (rs.rand(n, 1) * 2 * np.pi,), "cos": (rs.rand(n, 1) * 2 * np.pi,),
        "sin_f32arg": ((rs.rand(n, 1) * 2 * np.pi).astype(np.float32).astype(np.float64),), "cos_f32arg": ((rs.rand(n, 1) * 2 * np.pi).astype(np.float32).astype(np.float64),),
        "acos": (rs.rand(n, 1) * 2 - 1,), "asin": (rs.rand(n, 1) * 2 - 1,),
        "atan2": (rs.randn(n, 2),), "pow_5": (np.concatenate([rs.rand(n, 1), np.full((n, 1), 5.0)], 1),), "pow_2": (np.concatenate([rs.rand(n, 1) * 2 - 1, np.full((n, 1), 2.0)], 1),),
        "pow_frac": (np.concatenate([rs.rand(n, 1), 1.0 / (0.05 + rs.rand(n, 1))], 1),),
    }
    for name, (a,) in cases.items():
        fn = name.split("_")[0]
        got = rt.kat(fn, a)[:, 0]
        f = getattr(libm, fn)
        ref = np.array([f(*row) for row in a])
        ulp = np.abs(got.view(np.int64) - ref.view(np.int64))
        print(json.dumps({"fn": name, "n": n, "mismatch_share": float((ulp != 0).mean()), "max_ulp": int(ulp.max())}), flush=True)


if __name__ == "__main__":
    main()
