#!/usr/bin/env python3
"""Tuning aid: the two-light test scene at 1080p x 64 spp with the shadow walks inside the shade kernel (GI_DEFER_SHADOWS=0) and in k_st_shadow."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch, gi_raytracer_amd as gi, parity_checks as pc
    scene = pc.two_light_scene(True)
    rt = gi.RayTracer(0).setScene(scene)
    rt.tracePhotons(100000)
    w, h, spp = 1920, 1080, 64
    p = rt.params(w, h, min_samples=spp, max_samples=spp)
    buf = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
    for _ in range(2):
        rt.run_device(p, buf.data_ptr()); torch.cuda.synchronize()
    ms = rt.last_render_ms()[0]
    print("GI_DEFER_SHADOWS=%s: %.1f ms = %.1f Msamples/s" % (os.environ.get("GI_DEFER_SHADOWS", "1"), ms, w * h * spp / ms / 1e3), {k: round(v, 1) for k, v in rt.last_stage_ms().items()}, flush=True)
else:
    for d in ("0", "1"):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, GI_DEFER_SHADOWS=d), check=True)
