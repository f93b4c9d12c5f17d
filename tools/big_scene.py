"""A procedurally generated scene with many triangles (a bumpy glass blob and a bumpy diffuse blob over a tessellated floor): checks
that trees far larger than the LDS window render right, and what they cost.  Built through the programmatic API (no files)."""
import numpy as np
import gi_raytracer_amd as gi


def blob(centre, radius, n_lat, n_lon, bump, seed):
    rs = np.random.RandomState(seed)
    ph = rs.rand(6) * 6.28
    th = np.linspace(0, np.pi, n_lat + 1)[:, None]
    lo = np.linspace(0, 2 * np.pi, n_lon + 1)[None, :]
    r = radius * (1 + bump * (np.sin(5 * th + ph[0]) * np.sin(7 * lo + ph[1]) + 0.5 * np.sin(11 * lo * 0 + 13 * th + ph[2])))
    p = np.stack([r * np.sin(th) * np.cos(lo), r * np.cos(th) * np.ones_like(lo), r * np.sin(th) * np.sin(lo)], -1) + np.asarray(centre)
    n = p - np.asarray(centre)
    n = n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)
    uv = np.stack([np.broadcast_to(lo / (2 * np.pi), r.shape), np.broadcast_to(th / np.pi, r.shape)], -1)
    a, b, c, d = (slice(0, -1), slice(0, -1)), (slice(1, None), slice(0, -1)), (slice(1, None), slice(1, None)), (slice(0, -1), slice(1, None))
    def tri(x, i, j, k): return np.stack([x[i], x[j], x[k]], -2).reshape(-1, 3, x.shape[-1])
    pos = np.concatenate([tri(p, a, b, c), tri(p, a, c, d)]); nrm = np.concatenate([tri(n, a, b, c), tri(n, a, c, d)]); tuv = np.concatenate([tri(uv, a, b, c), tri(uv, a, c, d)])
    e1, e2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
    keep = np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-9          # drop the degenerate triangles at the poles
    f32 = lambda x: x.astype(np.float32).astype(np.float64)         # vertices as a mesh file would carry them (float)
    return f32(pos[keep]), f32(nrm[keep]), f32(tuv[keep])


def floor(size, n):
    g = np.linspace(-size, size, n + 1)
    x, z = np.meshgrid(g, g, indexing="ij")
    p = np.stack([x, np.zeros_like(x), z], -1)
    a, b, c, d = (slice(0, -1), slice(0, -1)), (slice(1, None), slice(0, -1)), (slice(1, None), slice(1, None)), (slice(0, -1), slice(1, None))
    def tri(i, j, k): return np.stack([p[i], p[j], p[k]], -2).reshape(-1, 3, 3)
    pos = np.concatenate([tri(a, c, b), tri(a, d, c)])
    nrm = np.tile(np.array([0.0, 1.0, 0.0]), (len(pos), 3, 1))
    uv = (pos[..., [0, 2]] + size) / (2 * size)
    return pos, nrm, uv


def build(n_lat=160, n_lon=320, textured=False, fog=False):
    s = gi.Scene()
    if textured:
        black = s.add_color_texture((0, 0, 0)); white = s.add_color_texture((1, 1, 1))
        cb = s.add_checkerboard((0.9, 0.9, 0.9), (0.2, 0.3, 0.7), 24)
        m_floor = s.add_material_tex(cb, black, 1, 1, 1); m_glass = s.add_material_tex(white, black, 0, 0, 1.5); m_diff = s.add_material_tex(s.add_checkerboard((0.8, 0.3, 0.2), (0.9, 0.8, 0.2), 16), black, 1, 1, 1)
    else:
        m_floor = s.add_material(1, 1, 1, (0.8, 0.8, 0.8)); m_glass = s.add_material(0, 0, 1.5, (1, 1, 1)); m_diff = s.add_material(1, 1, 1, (0.8, 0.3, 0.2))
    for (pos, nrm, uv), m in ((floor(8, 64), m_floor), (blob((-1.5, 1.6, 1.5), 1.4, n_lat, n_lon, 0.08, 1), m_glass), (blob((1.8, 1.3, -1.6), 1.2, n_lat, n_lon, 0.12, 2), m_diff)):
        s.add_triangles(pos, nrm, uv, np.full(len(pos), m, np.int32))
    if fog:   # a HeightFog slab over the floor (include/atmosphere.h:30-83): position, size, colour, density, scatter, noise scale
        s.add_height_fog((0, .6, 0), (10, 1.2, 10), (0.8, 0.85, 1.0), 2.0, .5, 4)
    s.add_light((2, 7, 3), (60, 60, 60), 0.1)
    s.set_ambient((0.03, 0.04, 0.06))
    return s.rebuild()


if __name__ == "__main__":
    import sys, time, torch
    sys.path.insert(0, "tests")
    import parity_checks as pc
    fog = "--fog" in sys.argv
    uhd = "--4k" in sys.argv
    t0 = time.time(); scene = build(fog=fog); t = scene.tables()
    print("triangles", len(t["tri_pos"]), "nodes", len(t["node_bbox"]), "refs", len(t["node_ent_idx"]), "host build %.2f s" % (time.time() - t0), flush=True)
    rt = gi.RayTracer(0).setScene(scene)
    ph, _ = rt.tracePhotons(100000)
    w, h, spp = (3840, 2160, 64) if uhd else (1920, 1080, 64)
    p = rt.params(w, h, min_samples=spp, max_samples=spp)
    buf = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
    for _ in range(2):
        rt.run_device(p, buf.data_ptr()); torch.cuda.synchronize()
    ms = rt.last_render_ms()[0]
    print(("4K" if uhd else "1080p") + (" + fog" if fog else "") + " x %d spp: %.1f ms = %.1f Msamples/s" % (spp, ms, w * h * spp / ms / 1e3), {k: round(v, 1) for k, v in rt.last_stage_ms().items()}, "wide", rt.set_wide_nodes(True), flush=True)
    o = pc.oracle_for(scene); o.set_photons(ph); o.build_photon_map()
    rows = np.array([200, 540, 800], np.int32) * (2 if uhd else 1)
    t0 = time.time(); lin, cnt = o.render_rows(w, h, rows, spp, rt.seed, 16); dt = time.time() - t0
    img = buf.cpu().numpy().astype(np.float64)
    d = np.abs(img[rows] - lin[rows]).max(axis=2)
    print("oracle rows: %.2f Msamples/s on 16 threads; rmse %.3e, pixels off by > 1e-6: %.4f, median %.2e" % (len(rows) * w * spp / dt / 1e6, np.sqrt(((img[rows] - lin[rows]) ** 2).mean()), (d > 1e-6).mean(), np.median(d)))
