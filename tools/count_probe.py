#!/usr/bin/env python3
"""Tuning aid: what the streaming kernels execute per pass (gi_set_counters 2 + GI_DEBUG_WF=1 prints cumulative counters after every pass).
usage: GI_DEBUG_WF=1 python tools/count_probe.py [scene w h spp photons]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gi_raytracer_amd as gi
from bench import SCN
a = sys.argv[1:]
name, w, h, spp, photons = (a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4])) if len(a) >= 5 else ("caustics", 1920, 1080, 32, 200000)
scene = gi.Scene.load(os.path.join(ROOT, SCN[name])).rebuild()
rt = gi.RayTracer(0).setScene(scene)
if photons:
    rt.tracePhotonsOnDevice(photons)
rt.set_counters("stream")
rt.run(w, h, min_samples=spp, max_samples=spp, f64=False)
c = rt.stream_counters()
n = w * h * spp
ms, _ = rt.last_render_ms()
print(name, w, h, spp, "counted frame %.1f ms" % ms, rt.last_kernel_ms(), {k: round(v / n, 3) for k, v in c.items()})
rt.set_counters(0)
for _ in range(2):
    rt.run(w, h, min_samples=spp, max_samples=spp, f64=False)
print("plain frame %.1f ms" % rt.last_render_ms()[0], rt.last_kernel_ms())
