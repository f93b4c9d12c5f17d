"""Backend-agnostic parity checks: the same assertions run on the CPU build of the device functions (not gpu) and on
the HIP library through the C ABI (gpu).  `rt` is a gi_raytracer_amd.RayTracer-like object; the oracle is the checker."""
import os

import numpy as np

import gi_raytracer_amd as gi
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCN = {"test_scene": "scenes/test_scene/test.scn", "cornell": "scenes/cornell/test.scn", "caustics": "scenes/caustics/caustics.scn", "caustics_02": "scenes/caustics_02/caustics.scn",
       "teapot": "scenes/cornell/teapot.scn", "textures": "scenes/textures/tex.scn", "textures_opaque": "scenes/textures/tex_opaque.scn", "cornell_tex": "scenes/textures/cornell_tex.scn", "spheres": "scenes/spheres/spheres.scn", "spheres_opaque": "scenes/spheres/spheres_opaque.scn", "fog": "scenes/fog/fog.scn"}

# float tolerance of the path (north_star: pixel RMSE < 1e-4 on linear radiance); measured values are ~1e-16
RMSE_TOL = 1e-4


def load_scene(name):
    return gi.Scene.load(os.path.join(ROOT, SCN[name])).rebuild()


def oracle_for(scene):
    t, st = scene.tables(), scene.settings
    o = ol.Oracle().set_scene(t["tri_pos"], t["tri_nrm"], t["tri_uv"], t["tri_mat"], t["mats"], t["lights"][:, :7], t["ambient"], kind=t["ent_kind"])
    o.set_camera(list(st.cam_pos), list(st.cam_up), list(st.cam_forward), st.sensor_diag, st.focal_dist)
    if len(t["tex_kind"]):
        o.set_textures(t["tex_kind"], t["tex_param"], t["mat_tex"], t["tex_pixels"])
    if len(t["fog"]):
        o.set_fog(t["fog"], t["fog_grid_off"], t["fog_grid"])     # the same noise grid the product's loader generated
    return o.build_octree()


def check_halton(rt, golden):
    g = golden("halton")
    idxs, ref = g["halton_sample_idx"], g["halton_sample"]
    dims = np.repeat(np.arange(256, dtype=np.uint32), len(idxs))
    got = rt.halton_sample(dims, np.tile(idxs, 256)).reshape(256, len(idxs))
    assert got.tobytes() == ref.tobytes()                       # bit-exact floats, all 256 dimensions
    sizes = g["halton_enum_params"][:, :2]
    for k in range(len(sizes)):
        rows = g["halton_enum_index"][g["halton_enum_index"][:, 0] == k]
        got = rt.halton_index(int(sizes[k][0]), int(sizes[k][1]), rows[:, 1:4])
        assert np.array_equal(got, rows[:, 4])


def check_kats(rt, golden):
    """a-11: the samplers, fastPow / fastPrecisePow, refr, reflect as the device computes them, against the reference's known answers
    (tests/golden/kat.npz, captured from include/util.h / util.cpp).  fastPow / fastPrecisePow are integer bit manipulation + exact
    products: bit-exact.  The samplers go through sin / cos / acos / sqrt: bit-exact on the CPU build (glibc, like the reference); on the
    GPU the libm is OCML, whose results may differ from glibc's in the last bit, so the bound there is 4 ulp of the result's magnitude
    scale (1.0: these are unit vectors), with the share of bit-exact answers reported."""
    k = golden("kat")
    exact = getattr(rt, "libm_is_glibc", False)

    def close(got, ref, what):
        if exact:
            assert got.tobytes() == np.ascontiguousarray(ref).tobytes(), what
        else:
            assert np.abs(got - ref).max() <= 4 * 2.0 ** -52, (what, np.abs(got - ref).max())
    pw = k["kat_pow"]
    assert rt.kat("fastPow", pw[:, :2])[:, 0].tobytes() == np.ascontiguousarray(pw[:, 2]).tobytes()
    assert rt.kat("fastPrecisePow", pw[:, :2])[:, 0].tobytes() == np.ascontiguousarray(pw[:, 3]).tobytes()
    h = k["kat_hemi"]
    close(rt.kat("hemisphereSample_cos", h[:, :6]), h[:, 6:9], "hemisphereSample_cos")
    ph = k["kat_phong"]
    close(rt.kat("sample_phong", np.concatenate([ph[:, 0:3], ph[:, 6:9]], 1)), ph[:, 9:12], "sample_phong")
    c = k["kat_cap"]
    close(rt.kat("sphereCapSample_cos", c[:, :7]), c[:, 7:10], "sphereCapSample_cos")
    u = k["kat_unitvec"]
    close(rt.kat("randomUnitVec", u[:, :2]), u[:, 2:5], "randomUnitVec")
    r = k["kat_refr"]
    assert rt.kat("refr", r[:, :7]).tobytes() == np.ascontiguousarray(r[:, 7:10]).tobytes()        # +, *, sqrt only: IEEE-exact everywhere
    assert rt.kat("reflect", r[:, :6]).tobytes() == np.ascontiguousarray(r[:, 10:13]).tobytes()


def check_rng(rt):
    """a-16: the counter RNG that stands in for drand() (include/util.h:52-80 is a sequential xorshift64* chain a parallel device cannot replay;
    DESIGN.md "RNG contract"), as the device evaluates it through gi_kat, against the oracle's on a lattice of (seed, stream, depth, purpose, a, b):
    every draw bit-exact (integer mixing + one exact scaling), and the stream key too -- a regression in the key mixing shows here, not as a frame."""
    L = ol.lib()
    rs = np.random.RandomState(16)
    seeds = [0, 1, gi.DEFAULT_SEED, 0xffffffffffffffff, 0x5048544f4e5eed00 ^ gi.DEFAULT_SEED] + [int(v) for v in rs.randint(0, 2**63, 6, dtype=np.int64)]
    streams = [0, 1, 2, 1146617855, 2**31 - 1, 2**31, 2**32 - 1] + [int(v) for v in rs.randint(0, 2**32, 6, dtype=np.int64)]
    depths = [0, 1, 2, 3, 10, 63, 64, 65]
    purposes = [0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 17, 18, 19, 20, 21, 22, 1 | (1 << 8), 2 | (3 << 8), 7 | (255 << 8)]
    ab = [(0, 0), (1, 0), (0, 1), (1557, 2191), (2**32 - 1, 2**32 - 1), (17, 2**31)] + [(int(a), int(b)) for a, b in rs.randint(0, 2**32, (6, 2), dtype=np.int64)]
    rows = [(sd, st, d, pu, a, b) for sd in seeds for st in streams for d in depths for pu in purposes for (a, b) in ab[:4]]
    rows += [(seeds[2], st, 1, pu, a, b) for st in streams for pu in purposes for (a, b) in ab[4:]]
    arg = np.array([[sd >> 32, sd & 0xffffffff, st, d, pu, a, b] for (sd, st, d, pu, a, b) in rows], np.float64)
    got = rt.kat("rng", arg)
    want = np.array([L.gio_counter_rand(sd, st, d, pu, a, b) for (sd, st, d, pu, a, b) in rows])
    assert got[:, 0].tobytes() == want.tobytes()
    assert ((got[:, 0] >= 0) & (got[:, 0] < 1)).all()
    # distinct keys of one seed give distinct draws (across seeds the contract lets (seed, stream) and (seed + golden ratio, stream - 1) coincide)
    for sd in seeds:
        d = got[[r[0] == sd for r in rows], 0]
        assert len(np.unique(d)) == len(d)
    # the stream key alone (what rng_make leaves in the lane's registers): draws of one stream at depth 0, purpose 0 agree => keys agree; and
    # distinct streams of one seed have distinct keys
    keys = got[:, 1] * 2.0 ** 32 + got[:, 2]
    first = {}
    for r, k in zip(rows, keys):
        assert first.setdefault((r[0], r[1]), k) == k
    for sd in seeds:
        ks = [k for (s_, _), k in first.items() if s_ == sd]
        assert len(set(ks)) == len(ks)
    return len(rows)


def check_leaf_order(rt, fx, set_wide=None):
    """a-4: the leaves the device walk meets for a ray, in order, against Octree::intersectSorted's t0-sorted list of the reference
    (leaforder_* of the scene fixtures).  The reference orders leaves with equal t0 by insertion (pre-order DFS, include/octree.cpp:285-313);
    the device's front-to-back order may permute such a group, never anything else -- and the test says how many rays had such a group."""
    rays = fx["rays"][fx["leaforder_ray"]]
    off, node, t0 = fx["leaforder_off"], fx["leaforder_node"], fx["leaforder_t0"]
    out = {}
    for wide in ((True, False) if set_wide is not None else (None,)):
        if wide is not None and set_wide(wide) != wide:
            continue                                     # a tree the wide records do not cover walks per node either way
        got = rt.leaf_order(rays)
        n_exact = n_tie = 0
        for j, g in enumerate(got):
            ref, rt0 = node[off[j]:off[j + 1]], t0[off[j]:off[j + 1]]
            assert len(g) == len(ref), (j, g, ref)
            if np.array_equal(g, ref):
                n_exact += 1
                continue
            # same leaves; every leaf stays inside its group of equal t0
            assert sorted(g) == sorted(ref), (j, g, ref)
            pos = {int(v): i for i, v in enumerate(ref)}
            assert all(rt0[pos[int(v)]] == rt0[i] for i, v in enumerate(g)), (j, g, ref, rt0)
            n_tie += 1
        out[wide] = (n_exact, n_tie)
        assert n_exact > 0.9 * len(got), out
    if set_wide is not None:
        set_wide(True)
    return out


def check_trace_table(rt, fx):
    """RayTracer::trace against the reference's own answers: hit flag / entity index exact, hit point and normal bit-exact."""
    hit, ent, res = rt.trace(fx["rays"])
    assert np.array_equal(hit, fx["trace_hit"])
    assert np.array_equal(ent, fx["trace_ent"])
    assert np.array_equal(res[:, :6], fx["trace_res"][:, :6])


def check_visible_table(rt, fx):
    assert np.array_equal(rt.visible(fx["shadow_q"]), fx["shadow_vis"])


def check_gather_table(rt, scene, fx):
    scene.build_photon_map(fx["photons"])
    rt.upload_photon_map()
    res, nc = rt.samplePhotons(fx["gather_q"])
    assert np.array_equal(nc, fx["gather_ncand"])                # candidate counts exact
    ref = fx["gather_res"]
    assert np.array_equal(res == 0, ref == 0)
    np.testing.assert_allclose(res, ref, rtol=1e-9, atol=1e-300)  # summation order differs (distance-sorted vs leaf order)


def check_emission(rt, scene, n):
    """tracePhotons: identical photon set (every draw is keyed) as the oracle."""
    o = oracle_for(scene)
    ph, tries = rt.tracePhotons(n)
    on, otries = o.emit_photons(n, 5, ol.RNG_COUNTER, rt.seed)
    oph = o.get_photons()
    assert len(ph) == on and tries == otries
    # same photons; refraction through glass amplifies the <=1 ulp differences between OCML and glibc sin/cos/acos
    np.testing.assert_allclose(ph, oph, rtol=1e-9, atol=1e-12)
    return o, ph


def check_render(rt, scene, w, h, spp, photons, adaptive=False, tol=RMSE_TOL, spp_mismatch=0.0):
    o = oracle_for(scene)
    if photons > 0 and scene.desc().n_light > 0:
        ph, _ = rt.tracePhotons(photons)
        o.set_photons(ph)
    o.build_photon_map()
    if adaptive:
        img, nspp = rt.run(w, h, min_samples=spp, max_samples=4 * spp, noise_thresh=0.0015, want_spp=True)
        ref = o.render(w, h, spp, 4 * spp, 0.0015)
        assert (nspp != ref["spp"]).mean() <= spp_mismatch       # per-pixel sample counts are integer work: exact (spp_mismatch = 0)
    else:
        img, nspp = rt.run(w, h, min_samples=spp, max_samples=spp, want_spp=True)
        ref = o.render(w, h, spp)
        assert (nspp == spp).all(), (np.unique(nspp, return_counts=True), float(img.mean()))
    rmse = float(np.sqrt(((img - ref["lin"]) ** 2).mean()))
    assert rmse < tol, rmse
    return rmse, img, ref["lin"]


def two_light_scene(with_glass=True):
    """A floor, a diffuse and a glass (or second diffuse) block, two lights of different colour (no reference scene has more than one): the reference keeps the
    share of the LAST visible light of a vertex (`i = ...` inside its light loop, include/raytracer.h), photons are emitted per (index, light)."""
    s = gi.Scene()
    white = s.add_material(1, 1, 1, (0.8, 0.8, 0.8)); red = s.add_material(1, 1, 1, (0.8, 0.3, 0.2)); glass = s.add_material(0, 0, 1.5, (1, 1, 1))
    quad = lambda a, b, c, d: [[a, b, c], [a, c, d]]
    tris, mats = [], []
    def box(lo, hi, m):
        (x0, y0, z0), (x1, y1, z1) = lo, hi
        f = [((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0)), ((x0, y0, z1), (x0, y1, z1), (x1, y1, z1), (x1, y0, z1)),
             ((x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (x0, y0, z1)), ((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0)),
             ((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1)), ((x0, y0, z0), (x0, y0, z1), (x1, y0, z1), (x1, y0, z0))]
        for q in f:
            tris.extend(quad(*q)); mats.extend([m, m])
    tris.extend(quad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4))); mats.extend([white, white])
    box((-1.6, 0.001, -0.4), (-0.6, 1.0, 0.6), red)
    box((0.5, 0.001, -0.7), (1.5, 1.2, 0.3), glass if with_glass else white)
    s.add_triangles(np.array(tris, float), mat_idx=mats)
    s.add_light((-2.5, 4, 1.5), (9, 7, 5), 0.1)
    s.add_light((3, 3.5, 2), (3, 5, 9), 0.15)
    s.set_camera((0.2, 2.2, 5.5), (0, 0.5, 0))
    return s.rebuild()


def coplanar_scene():
    """Entities that lie in one plane and overlap (a patch on the floor of a room, a panel in its ceiling -- what a Cornell box's light is): a ray into the
    overlap meets two entities at exactly the same distance and RayTracer::trace keeps the one it asks first (include/raytracer.h:446-472, strict <).
    The large floor triangles reach through many leaves, the patch through few, blocks on the floor make the octree split around them."""
    s = gi.Scene()
    white = s.add_material(1, 1, 1, (0.8, 0.8, 0.8)); red = s.add_material(1, 1, 1, (0.8, 0.3, 0.2)); blue = s.add_material(1, 1, 1, (0.2, 0.3, 0.8))
    quad = lambda a, b, c, d: [[a, b, c], [a, c, d]]
    tris, mats = [], []
    def put(q, m):
        tris.extend(quad(*q)); mats.extend([m, m])
    put(((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4)), white)                      # floor
    put(((-4, 4, -4), (4, 4, -4), (4, 4, 4), (-4, 4, 4)), white)                      # ceiling
    put(((-1.5, 0, -1.25), (-1.5, 0, 2.5), (2.25, 0, 2.5), (2.25, 0, -1.25)), red)    # patch in the floor's plane
    put(((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1)), blue)                       # panel in the ceiling's plane
    put(((-1.5, 0, -1.25), (-1.5, 0, 2.5), (2.25, 0, 2.5), (2.25, 0, -1.25)), blue)   # and the patch once more: the very same triangles
    rs = np.random.RandomState(11)
    for k in range(40):                                                               # small blocks standing on the floor
        x, z = rs.uniform(-3.5, 3.3, 2)
        w, h = rs.uniform(0.05, 0.2), rs.uniform(0.1, 0.6)
        (x0, y0, z0), (x1, y1, z1) = (x, 0.0, z), (x + w, h, z + w)
        for q in (((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0)), ((x0, y0, z1), (x0, y1, z1), (x1, y1, z1), (x1, y0, z1)),
                  ((x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (x0, y0, z1)), ((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0)),
                  ((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1))):
            put(q, white)
    s.add_triangles(np.array(tris, float), mat_idx=mats)
    s.add_light((0, 3.5, 0), (9, 9, 9), 0.1)
    s.set_camera((0.2, 2.2, 5.5), (0, 0.5, 0))
    return s.rebuild()


def check_equal_distance_hits(rt, scene, set_wide):
    """Rays into overlapping coplanar entities: the walk with its short cuts (boxes cut to the leaves, no look behind the best hit) returns the entity
    the plain per-node walk returns -- the one the reference meets first -- bit for bit, also where two or three entities tie."""
    rs = np.random.RandomState(3)
    n = 20000
    o = np.stack([rs.uniform(-3.5, 3.5, n), rs.uniform(0.5, 3.5, n), rs.uniform(-3.5, 3.5, n)], 1)
    tgt = np.stack([rs.uniform(-2.5, 3.0, n), np.where(rs.rand(n) < 0.7, 0.0, 4.0), rs.uniform(-2.0, 3.0, n)], 1)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1)[:, None]
    rays = np.concatenate([o, d], 1)
    assert set_wide(True)
    hit_w, ent_w, res_w = rt.trace(rays)
    set_wide(False)
    hit_n, ent_n, res_n = rt.trace(rays)
    set_wide(True)
    assert np.array_equal(hit_w, hit_n) and np.array_equal(ent_w, ent_n)
    assert np.array_equal(res_w[hit_w > 0].view(np.uint64), res_n[hit_n > 0].view(np.uint64))
    ents = set(ent_w[hit_w > 0].tolist())
    assert ents & {0, 1} and ents & {2, 3} and len(ents) > 20         # floor, ceiling, blocks
    return ents


def check_gather_float_ties(rt_factory):
    """Photons whose squared distances to the query agree to float precision around rank 32: the float-key heap cannot
    separate them, the exact pass must.  40 photons on a ray from the query point, spacing 1e-9."""
    s = gi.Scene()
    m = s.add_material(1.0, 1.0, 1.0, (1, 1, 1))
    s.add_triangles(np.array([[[0, 0, 0], [4, 0, 0], [0, 0, 4]], [[4, 0, 0], [4, 0, 4], [0, 0, 4]], [[0, 4, 0], [4, 4, 0], [0, 4, 4]]], float), mat_idx=[m] * 3)
    s.add_light((2, 3, 2), (1, 1, 1), .05)
    s.rebuild()
    rs = np.random.RandomState(5)
    n = 40
    q = np.array([1.0, 1.0, 1.0]) * 2.0 ** -3
    # 40 photons on a shell of radius 2^-20 (1 + k 2^-30) around q: squared distances differ by ~2e-9 relative, far below
    # float resolution; 10 clearly nearer ones put rank 32 inside that group.  The whole cluster (radius 1e-6) lies inside
    # the +-1e-5 neighbourhood of the leaf that contains q, so all 50 are candidates.
    u = rs.randn(n, 3); u /= np.linalg.norm(u, axis=1)[:, None]
    pos = q + u * (2.0 ** -20 * (1 + rs.permutation(n) * 2.0 ** -30))[:, None]
    d = rs.randn(n, 3); d /= np.linalg.norm(d, axis=1)[:, None]
    ph = np.concatenate([pos, d, rs.rand(n, 3)], axis=1)
    un = rs.randn(10, 3); un /= np.linalg.norm(un, axis=1)[:, None]
    near = np.concatenate([q + un * 2.0 ** -21 * (1 + rs.rand(10, 1)* 0.5), d[:10], rs.rand(10, 3)], axis=1)
    ph = np.concatenate([near, ph])
    rt = rt_factory().setScene(s)
    s.build_photon_map(ph)
    rt.upload_photon_map()
    o = oracle_for(s)
    o.set_photons(ph).build_photon_map()
    qq = np.array([[*q, 0.3, 0.5, 0.8]])
    got, nc = rt.samplePhotons(qq)
    ref, nco = o.gather(qq)
    assert nc[0] == nco[0] == 50
    np.testing.assert_allclose(got, ref, rtol=1e-9)


def check_stripes(rt, scene, w, h, spp, world, stripe_h):
    """Row-stripe sharding: rendering the stripes of every rank and interleaving them gives the single-GPU frame exactly."""
    full = rt.run(w, h, min_samples=spp, max_samples=spp)
    frame = np.zeros_like(full)
    for rank in range(world):
        part = rt.run(w, h, min_samples=spp, max_samples=spp, stripe_h=stripe_h, rank=rank, world=world)
        rows = stripe_rows(h, stripe_h, rank, world)
        assert len(rows) == len(part)
        frame[rows] = part
    assert np.array_equal(frame, full)


from gi_raytracer_amd.sharding import stripe_rows  # noqa: E402  (re-exported for the tests)


def adversarial_rays(scene, n=6000, seed=5):
    """Rays that stress the box arithmetic of the octree walk: random ones, axis-parallel ones (a zero direction component makes
    invDir infinite and 0 * inf = NaN when the origin lies on a plane), origins exactly on node planes, and -0.0 components."""
    t = scene.tables()
    bb = t["node_bbox"]
    rs = np.random.RandomState(seed)
    lo, hi = bb[0, :3], bb[0, 3:]
    o = lo + (hi - lo) * (rs.rand(n, 3) * 1.4 - 0.2)
    d = rs.randn(n, 3)
    k = n // 6
    for j in range(k):                              # axis-parallel and plane-parallel directions
        d[j, rs.randint(3)] = 0.0
    for j in range(k, 2 * k):
        a = rs.randint(3)
        d[j] = 0.0
        d[j, a] = rs.choice([-1.0, 1.0])
    for j in range(2 * k, 3 * k):                   # origins on planes of random nodes (min / max / a child's planes)
        nb = bb[rs.randint(len(bb))]
        a = rs.randint(3)
        o[j, a] = nb[a + 3 * rs.randint(2)]
    for j in range(3 * k, 4 * k):                   # both: origin on a plane, direction inside that plane
        nb = bb[rs.randint(len(bb))]
        a = rs.randint(3)
        o[j, a] = nb[a + 3 * rs.randint(2)]
        d[j, a] = rs.choice([0.0, -0.0])
    for j in range(4 * k, 5 * k):                   # negative zeros
        d[j, rs.randint(3)] = -0.0
    nrm = np.sqrt((d * d).sum(1))
    d = d / np.where(nrm == 0, 1, nrm)[:, None]
    return np.concatenate([o, d], 1)


def check_wide_walk(rt, scene, set_wide):
    """The wide-record walk and the per-node walk give identical answers (hit flag, entity, hit point bits, visibility)."""
    rays = adversarial_rays(scene)
    rs = np.random.RandomState(9)
    q = np.concatenate([rays[:, :3], rays[:, :3] + rays[:, 3:] * (rs.rand(len(rays), 1) * 12 + 0.01)], 1)
    if (scene.tables()["node_child"][0] < 0).all():
        assert not set_wide(True)          # the root is a leaf: nothing to walk, the per-node path handles it
        return
    assert set_wide(True), "the scene's octree was not accepted as exact octants"
    hit_w, ent_w, res_w = rt.trace(rays)
    vis_w = rt.visible(q)
    set_wide(False)
    hit_n, ent_n, res_n = rt.trace(rays)
    vis_n = rt.visible(q)
    set_wide(True)
    assert hit_w.sum() > 100
    assert np.array_equal(hit_w, hit_n) and np.array_equal(ent_w[hit_w > 0], ent_n[hit_n > 0])
    assert np.array_equal(res_w[hit_w > 0].view(np.uint64), res_n[hit_n > 0].view(np.uint64))
    assert np.array_equal(vis_w, vis_n)


def check_content_culling(rt, scene, photons=2000, render=True):
    """Content-box culling (a child of the walk is skipped when the ray misses the box of everything referenced below it) changes nothing:
    hit flag, entity, hit point bits, visibility on the adversarial rays, and whole frames bit for bit, with and without."""
    rays = adversarial_rays(scene)
    rs = np.random.RandomState(9)
    q = np.concatenate([rays[:, :3], rays[:, :3] + rays[:, 3:] * (rs.rand(len(rays), 1) * 12 + 0.01)], 1)
    if not rt.set_content_culling(True):
        return False                      # the tree does not take the wide walk: nothing to compare
    a = rt.trace(rays); va = rt.visible(q)
    assert not rt.set_content_culling(False)
    b = rt.trace(rays); vb = rt.visible(q)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64)) and np.array_equal(va, vb)
    if render:
        if scene.desc().n_light:
            rt.tracePhotons(photons)
        fb = rt.run(40, 30, min_samples=4, max_samples=4)
        assert rt.set_content_culling(True)
        fa = rt.run(40, 30, min_samples=4, max_samples=4)
        assert np.array_equal(fa.view(np.uint64), fb.view(np.uint64))
    rt.set_content_culling(True)
    return True


def check_photon_descent(rt, scene, photons=20000):
    """PhotonMap::getBounds walked one record per level (children's boxes from the parent's planes) and two records per level give
    identical gathers: inside, outside and on the faces of the map."""
    rt.tracePhotons(photons)
    t = scene.tables()
    bb = t["node_bbox"][0]
    rs = np.random.RandomState(1)
    pos = bb[:3] + (bb[3:] - bb[:3]) * (rs.rand(20000, 3) * 1.1 - 0.05)
    ph = scene.photon_tables()
    nb = ph["node_bbox"]
    for j in range(2000):                       # queries exactly on planes of photon-map nodes
        b = nb[rs.randint(len(nb))]
        pos[j, rs.randint(3)] = b[rs.randint(6)]
    q = np.concatenate([pos, rs.randn(len(pos), 3)], 1)
    rt.set_wide_nodes(True)
    assert rt.photon_planes, "the photon octree was not accepted for the one-record descent"
    a, na = rt.samplePhotons(q)
    rt.set_wide_nodes(False)
    assert not rt.photon_planes
    b, nbc = rt.samplePhotons(q)
    rt.set_wide_nodes(True)
    assert (na > 0).mean() > 0.1 and (na == 0).any()
    assert np.array_equal(na, nbc) and np.array_equal(a.view(np.uint64), b.view(np.uint64))
