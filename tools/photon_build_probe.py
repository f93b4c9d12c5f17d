#!/usr/bin/env python3
"""The photon map of the reference's largest example (examples/caustics/test_16/render_7.5m.png: 7.5 M photons): build on the device
(gi_build_photon_map) against the host builder + upload, tables compared byte for byte, both timed."""
import sys, time
import numpy as np
import gi_raytracer_amd as gi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 7500000
scene = gi.Scene.load("scenes/caustics/caustics.scn").rebuild()
rt = gi.RayTracer(0).setScene(scene)
bb = scene.tables()["node_bbox"][0]
rs = np.random.RandomState(3)
# a caustic-like distribution: most photons in a few dense blobs on the floor, the rest spread out
c = bb[:3] + (bb[3:] - bb[:3]) * rs.rand(6, 3)
pos = np.concatenate([c[k] + rs.randn(n // 8, 3) * [0.15, 0.002, 0.15] for k in range(6)] + [bb[:3] + (bb[3:] - bb[:3]) * rs.rand(n - 6 * (n // 8), 3)])
d = rs.randn(n, 3); d /= np.linalg.norm(d, axis=1)[:, None]
ph = np.concatenate([pos, d, rs.rand(n, 3)], 1)
t0 = time.time(); scene.build_photon_map(ph); t1 = time.time(); rt.upload_photon_map(); t2 = time.time()
host = rt.photon_tables_on_device()
t3 = time.time(); rt.build_photon_map_on_device(ph); t4 = time.time()
dev = rt.photon_tables_on_device()
same = all(np.array_equal(host[k], dev[k]) for k in host)
print(f"{n} photons: {len(host['nodes'])} nodes, {len(host['ranges'])} ranges; host build {t1 - t0:.2f} s + flatten/upload {t2 - t1:.2f} s; device build (incl. the upload of the photons) {t4 - t3:.2f} s; tables identical: {same}")
