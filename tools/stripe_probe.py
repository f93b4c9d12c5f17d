"""Time one rank's share of the default frame for several world sizes on ONE GPU (predicts strong scaling)."""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gi_raytracer_amd as gi
scene = gi.Scene.load("scenes/caustics/caustics.scn").rebuild()
rt = gi.RayTracer(0).setScene(scene)
rt.tracePhotons(200000)
w, h, spp = 1920, 1080, 256
base = None          # time of the whole frame, when world 1 is among the runs
for world in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    sh = 16 if world > 1 else h
    p = rt.params(w, h, stripe_h=sh, rank=0, world=world, min_samples=spp, max_samples=spp)
    rows = rt.local_rows(p)
    buf = torch.empty((rows, w, 3), dtype=torch.float32, device="cuda:0")
    for it in range(2):
        rt.run_device(p, buf.data_ptr())
        torch.cuda.synchronize()
    ms = rt.last_render_ms()[0]
    st = rt.last_stage_ms()
    if world == 1:
        base = ms
    print(world, rows, round(ms, 1), "speed-up", round(base / ms, 2) if base else None, {k: round(v, 1) for k, v in st.items()}, flush=True)
