import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, gi_raytracer_amd as gi, parity_checks as pc
scene = pc.load_scene("caustics")
rt = gi.RayTracer(0).setScene(scene); rt.tracePhotons(200000)
w, h = 1920, 1080
for (mn, mx) in ((8, 32), (32, 128), (64, 256)):
    p = rt.params(w, h, min_samples=mn, max_samples=mx, noise_thresh=0.0015)
    buf = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0"); spp = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    for _ in range(2):
        rt.run_device(p, buf.data_ptr(), spp_ptr=spp.data_ptr()); torch.cuda.synchronize()
    ms = rt.last_render_ms()[0]; n = int(spp.sum().item())
    st = rt.last_stage_ms()
    print("adaptive %d..%d: %.1f ms, %d samples (mean spp %.1f) -> %.1f Msamples/s" % (mn, mx, ms, n, n / (w * h), n / ms / 1e3), {k: round(v, 1) for k, v in st.items()}, "launches", rt.last_render_ms()[1], flush=True)
