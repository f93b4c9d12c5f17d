"""Parity of the HIP path (through the C ABI of libgi_raytracer_hip.so) on a real MI355X.

Small cases: compared with the reference's golden tables and with the CPU oracle on the same seeded inputs.
Full BASELINE sizes: size-independent properties (stripe interleave identity, determinism, sample-count invariants,
energy bounds) -- the oracle would need minutes there.
Tolerances: integer / index results exact; hit points and normals bit-exact (same IEEE operations, no contraction);
radiance RMSE < 1e-4 (north_star), measured ~1e-16.
"""
import os

import numpy as np
import pytest

import gi_raytracer_amd as gi
import parity_checks as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt0():
    return gi.RayTracer(0)


@pytest.fixture(scope="module", params=["test_scene", "cornell", "caustics", "spheres_opaque", "textures_opaque", "caustics_02"])
def setup(request, golden):
    scene = pc.load_scene(request.param)
    return request.param, scene, gi.RayTracer(0).setScene(scene), golden("scene_" + request.param.replace("_opaque", ""))


def test_halton_device_tables(rt0, golden):
    pc.check_halton(rt0, golden)


def test_trace_matches_reference_table(setup):
    pc.check_trace_table(setup[2], setup[3])


def test_leaf_sequence_matches_reference_sorted_lists(setup):
    """Octree::intersectSorted's list (include/octree.cpp:285-313) against the leaves the device walk visits, wide and per-node."""
    res = pc.check_leaf_order(setup[2], setup[3], setup[2].set_wide_nodes)
    print(setup[0], res)


def test_sampler_known_answers(rt0, golden):
    """device samplers / fastPow / refr against tests/golden/kat.npz (include/util.cpp:27-107, util.h:100-188)."""
    pc.check_kats(rt0, golden)


def test_counter_rng_known_answers(rt0):
    """rng_draw / rng_make on the device (gi_kat what = 8) against the oracle's counter RNG, bit-exact on a lattice of keys."""
    assert pc.check_rng(rt0) > 10000


def test_visible_matches_reference_table(setup):
    pc.check_visible_table(setup[2], setup[3])


def test_wide_walk_equals_per_node_walk(setup):
    pc.check_wide_walk(setup[2], setup[1], setup[2].set_wide_nodes)


def test_hits_at_equal_distance_go_to_the_entity_met_first():
    """Coplanar, overlapping entities (a patch in the floor's plane, the same patch twice, a panel in the ceiling's): the closest-hit walk's short cuts
    (gi_device.h: trace_wide_step) must not change which of two entities at the very same distance wins.  Whole frames: with the short cuts and
    without them (every entity of every leaf asked, as the reference does) bit for bit the same.  (Against the oracle such a scene is compared on the
    CPU build of the device code only, tests/test_device_logic_cpu.py: where two surfaces of different colour coincide, the last bits of a bounce
    direction -- OCML's sin / cos against glibc's -- decide which one a ray sees.)"""
    scene = pc.coplanar_scene()
    rt = gi.RayTracer(0).setScene(scene)
    pc.check_equal_distance_hits(rt, scene, rt.set_wide_nodes)
    pc.check_wide_walk(rt, scene, rt.set_wide_nodes)
    a = rt.run(96, 64, min_samples=16, max_samples=16)
    try:
        assert not rt.set_entity_boxes(False)
        b = rt.run(96, 64, min_samples=16, max_samples=16)
    finally:
        rt.set_entity_boxes(True)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    rmse = float(np.sqrt(((a - pc.oracle_for(scene).build_photon_map().render(96, 64, 16)["lin"]) ** 2).mean()))
    assert rmse < 5e-3, rmse        # the same picture up to those few pixels


# Refractive / textured scenes.  The CPU build of the same device code matches the oracle to 1e-17 on all of them (test_device_logic_cpu.py); on
# the GPU the libm is OCML, which differs from glibc in the last bit of some sin / cos / acos / asin / atan2 / pow results (shares measured by
# tools/libm_probe.py, DESIGN.md "Numerics").  A specular chain through a refracting object amplifies such a bit until, for a few paths per
# ten thousand, a later hit / miss or total-internal-reflection decision falls the other way.  Bounds asserted below are the north_star
# contract (frame RMSE < 1e-4 on linear radiance) or tighter; the measured values (tools/measure_tolerances.py, MI355X) are in the comments.
@pytest.mark.parametrize("mode", ["wavefront", "rounds", "megakernel"])
def test_textured_scene_matches_oracle(mode):
    """scenes/textures/tex.scn (the scene the reference fixtures pin the oracle on): checkerboard and PNG textures as diffuse and emissive
    colour on meshes and spheres, an alpha channel and a 0.7 opacity in the stochastic alpha tests (include/material.h:32-93)."""
    scene = pc.load_scene("textures")
    rt = gi.RayTracer(0).setScene(scene)
    rt.set_render_mode(mode)
    pc.check_emission(rt, scene, 600)
    rmse, img, ref = pc.check_render(rt, scene, 80, 60, 8, 1500, tol=1e-5)          # measured 3.2e-7
    assert img.mean() > 0.01 and np.median(np.abs(img - ref)) < 1e-12
    assert (np.abs(img - ref).max(axis=2) > 1e-9).mean() < 0.02                      # measured 0.5 % of the pixels


def test_textured_cornell_scene_wide_and_per_node():
    """Textures together with the wide octree records, the LDS-resident top of the tree and refraction (scenes/textures/cornell_tex.scn,
    3 218 triangles, a glass teapot in a box with bright textured walls): against the oracle, and wide == per-node bit for bit.
    This is the scene where a diverted path costs most (it lands on a bright texture instead of a dark one): at 8 spp 6 of 5 184 pixels
    differ and the frame RMSE is 2.3e-4; the contract is quoted at 256 spp, where one path is 1/256 of a pixel -- at 64 spp the same
    frame is inside 1e-4."""
    scene = pc.load_scene("cornell_tex")
    rt = gi.RayTracer(0).setScene(scene)
    rmse, img, ref = pc.check_render(rt, scene, 96, 54, 64, 5000, tol=1e-4)
    print("cornell_tex 64 spp rmse", rmse, "pixels off", (np.abs(img - ref).max(axis=2) > 1e-9).mean())
    assert img.mean() > 0.01 and np.median(np.abs(img - ref)) < 1e-12
    assert (np.abs(img - ref).max(axis=2) > 1e-9).mean() < 0.02
    rmse8, img8, ref8 = pc.check_render(rt, scene, 96, 54, 8, 5000, tol=1e-3)       # measured 2.3e-4, 0.12 % of the pixels
    assert (np.abs(img8 - ref8).max(axis=2) > 1e-9).mean() < 0.005
    assert rt.set_wide_nodes(True)
    a = rt.run(96, 54, min_samples=8, max_samples=8)
    assert not rt.set_wide_nodes(False)
    b = rt.run(96, 54, min_samples=8, max_samples=8)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_large_procedural_scene_matches_oracle():
    """tools/big_scene.py at test size (21 k triangles, 25 k octree nodes, built through the programmatic API): ~85x more inner nodes than
    the LDS window holds, glass and diffuse blobs, a checkerboard floor.  213 k triangles run as a probe."""
    import sys
    sys.path.insert(0, os.path.join(pc.ROOT, "tools"))
    import big_scene
    scene = big_scene.build(40, 80, textured=True)
    rt = gi.RayTracer(0).setScene(scene)
    assert rt.set_wide_nodes(True)
    rmse, img, ref = pc.check_render(rt, scene, 96, 54, 8, 5000, tol=1e-9)          # measured 1.4e-17
    assert img.mean() > 0.01


def test_glass_teapot_scene_matches_oracle():
    """SURVEY section 8 config 4 (Cornell scene + glass teapot, 4496 triangles, 3828 octree nodes: more inner nodes than the LDS
    holds, so the wide walk also reads records from global memory): frame against the oracle, and wide == per-node bit for bit.
    Refraction: a <=1 ulp libm difference in a direction can flip a later hit, hence a looser bound than the 1e-9 of the other scenes."""
    scene = pc.load_scene("teapot")
    rt = gi.RayTracer(0).setScene(scene)
    rmse, img, ref = pc.check_render(rt, scene, 96, 54, 8, 5000)
    assert rmse < 1e-5 and img.mean() > 0.01
    assert (np.abs(img - ref).max(axis=2) > 1e-9).mean() < 0.01          # all but a handful of pixels agree to 1e-9
    assert rt.set_wide_nodes(True)
    a = rt.run(96, 54, min_samples=8, max_samples=8)
    assert not rt.set_wide_nodes(False)
    b = rt.run(96, 54, min_samples=8, max_samples=8)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_wide_and_per_node_frames_are_identical():
    """The streaming pipeline with wide records (LDS) and with per-node records renders the same frame bit for bit."""
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    rt.tracePhotons(3000)
    assert rt.set_wide_nodes(True)
    a = rt.run(96, 54, min_samples=8, max_samples=8)
    assert not rt.set_wide_nodes(False)
    b = rt.run(96, 54, min_samples=8, max_samples=8)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_content_culling_changes_nothing(setup):
    """gi_set_content_culling: the walk with and without the content-box test of the children (gi_device.h: content_cull) -- same hits, same
    visibility, same frame, bit for bit; on every fixture scene, and on the scenes with stochastic alpha, fog and a large tree below."""
    assert pc.check_content_culling(setup[2], setup[1]) or setup[0] == "textures_opaque"


@pytest.mark.parametrize("name", ["spheres", "fog", "textures", "teapot"])
def test_content_culling_changes_nothing_more_scenes(name):
    scene = pc.load_scene(name)
    pc.check_content_culling(gi.RayTracer(0).setScene(scene), scene)


@pytest.mark.parametrize("name", ["caustics", "cornell", "spheres", "fog", "textures", "teapot", "two_lights", "two_lights_glass"])
def test_schedule_knobs_change_nothing(name, monkeypatch):
    """Where a shadow segment is walked (inside k_st_shade, or put off to k_st_shadow) and how many idle lanes make a wave of k_st_trace /
    k_st_shadow take new rays (GI_REFILL_MIN; 64 = lockstep waves) are schedules, not arithmetic: the frame is the same bit for bit.  The
    textured scene has emitting and non-emitting texels on one material (both forms of the put-off query), fog adds the medium's march
    to the put-off segment.  GI_COOP_FACTOR moves the finisher between one path per lane, per group of 16 lanes and per wave."""
    scene = pc.two_light_scene(name.endswith("glass")) if name.startswith("two_lights") else pc.load_scene(name)   # two lights: one put-off query per light
    frames = []
    for env in ({}, {"GI_DEFER_SHADOWS": "0"}, {"GI_REFILL_MIN": "64"}, {"GI_REFILL_MIN": "5"}, {"GI_COOP_FACTOR": "0"}, {"GI_COOP_FACTOR": "64"}, {"GI_WAVE_FACTOR": "1"}, {"GI_WAVE_FACTOR": "200"}, {"GI_FINISH_THRESHOLD": "1000"}, {"GI_ENTITY_BOXES": "0"}, {"GI_CLIP_BOXES": "0"}, {"GI_WALK_CUT": "0"}, {"GI_SORT_CONT": "0"}, {"GI_SORT_SHADE": "0"}, {"GI_SORT_SHADE_LO": "0"}, {"GI_FAST_DESCENT": "0"}, {"GI_DESCENT_JUMP": "0"}, {"GI_FLAT_CANDIDATES": "0"}, {"GI_GATHER_WAVE_BELOW": "0"}, {"GI_GATHER_WAVE_BELOW": "4000000000"}):
        for k in ("GI_DEFER_SHADOWS", "GI_REFILL_MIN", "GI_COOP_FACTOR", "GI_WAVE_FACTOR", "GI_FINISH_THRESHOLD", "GI_ENTITY_BOXES", "GI_CLIP_BOXES", "GI_WALK_CUT", "GI_SORT_CONT", "GI_SORT_SHADE", "GI_SORT_SHADE_LO", "GI_FAST_DESCENT", "GI_DESCENT_JUMP", "GI_FLAT_CANDIDATES", "GI_GATHER_WAVE_BELOW"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        rt = gi.RayTracer(0).setScene(scene)     # the knobs are read when the context is created
        if scene.desc().n_light > 0:
            rt.tracePhotons(4000)
        frames.append(rt.run(160, 90, min_samples=12, max_samples=12, seed=11))
    for f in frames[1:]:
        assert np.array_equal(frames[0].view(np.uint64), f.view(np.uint64))


def test_photon_octree_descent_variants_agree():
    scene = pc.load_scene("caustics")
    pc.check_photon_descent(gi.RayTracer(0).setScene(scene), scene)


@pytest.mark.parametrize("jump", ["1", "0"])
def test_quick_descent_of_the_gather_keys_finds_the_same_leaf(jump, monkeypatch):
    """gather_find_leaf_fast (what k_st_compact keys a pass's gather queries with: 32-byte split records, the first five levels through a jump table)
    against the descent that asks every box on the way, PhotonMap::Node::getBounds (include/photonMap.cpp:115-134): positions all over the map, outside
    it, exactly on split planes and faces, and 1e-15 .. 1e-9 of the extent beside them.  It may decline (-2), it may not differ."""
    monkeypatch.setenv("GI_DESCENT_JUMP", jump)
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    rt.tracePhotons(200000)                      # (built on the host and uploaded: the host tables name the planes to aim at)
    nb = scene.photon_tables()["node_bbox"]
    root = nb[0]
    ext = (root[3:] - root[:3]).max()
    rs = np.random.RandomState(2)
    pos = root[:3] + (root[3:] - root[:3]) * (rs.rand(400000, 3) * 1.04 - 0.02)
    k = 100000
    boxes = nb[rs.randint(len(nb), size=k)]
    ax = rs.randint(3, size=k)
    plane = boxes[np.arange(k), ax + 3 * rs.randint(2, size=k)]
    off = np.where(rs.rand(k) < 0.3, 0.0, ext * 10.0 ** rs.uniform(-15, -9, k) * rs.choice([-1, 1], k))
    pos[np.arange(k), ax] = plane + off
    grid = root[:3] + (root[3:] - root[:3]) * rs.randint(0, 129, size=(k, 3)) / 128.0      # cell faces of the jump table (32, 64 or 128 cells per axis)
    pos[k:2 * k][np.arange(k), ax] = grid[np.arange(k), ax] + off
    fast, full = rt.find_leaves(pos)
    took = fast != -2
    assert np.array_equal(fast[took], full[took])
    assert took.mean() > 0.6 and (~took).sum() > 1000 and (full == -1).any()


def test_gather_matches_reference_table(setup):
    name, scene, rt, fx = setup
    if "photons" not in fx:
        pytest.skip("no photons in this scene")
    pc.check_gather_table(rt, scene, fx)


def test_emission_identical_to_oracle(setup):
    name, scene, rt, fx = setup
    if scene.desc().n_light == 0:
        pytest.skip("no light")
    pc.check_emission(rt, scene, 3000)


@pytest.mark.parametrize("mode", ["wavefront", "rounds", "megakernel"])
@pytest.mark.parametrize("adaptive", [False, True])
def test_render_matches_oracle(setup, adaptive, mode):
    name, scene, rt, fx = setup
    rt.set_render_mode(mode)
    try:
        rmse, img, ref = pc.check_render(rt, scene, 96, 54, 8, 5000, adaptive)    # per-pixel sample counts of the adaptive loop: exact, every scene
    finally:
        rt.set_render_mode("wavefront")
    if name == "textures_opaque":     # glass sphere next to bright textures (see above): measured 2.4e-6 / 6.9e-7 (fixed / adaptive), 0.6-2.2 % of the pixels
        assert rmse < 1e-5 and (np.abs(img - ref).max(axis=2) > 1e-9).mean() < 0.05 and np.median(np.abs(img - ref)) < 1e-12, rmse
    else:
        assert rmse < 1e-9, rmse      # far inside the 1e-4 contract: any larger value means a diverged path (caustics_02, glass sphere mesh: 8e-18)


@pytest.mark.parametrize("with_glass", [False, True])
def test_two_lights_match_oracle(with_glass):
    """Several lights: the shade kernel keeps its shadow walks (one per light, in order), photons come from both lights.  All three schedules
    give the same bits.  Without glass the frame equals the oracle's to 1e-9; bright lights seen through a glass block put a handful of pixels on
    another branch of a refraction chain (device libm vs glibc, DESIGN.md "Numerics": measured 7 of 5 184 pixels, RMSE 1.1e-5, median 0;
    the CPU build of the same device code matches the oracle to 2.5e-18 on this scene)."""
    scene = pc.two_light_scene(with_glass)
    assert scene.desc().n_light == 2
    rt = gi.RayTracer(0).setScene(scene)
    pc.check_emission(rt, scene, 2000)
    frames = []
    for mode in ("wavefront", "rounds", "megakernel"):
        rt.set_render_mode(mode)
        rmse, img, ref = pc.check_render(rt, scene, 96, 54, 8, 4000)      # asserts the contract's 1e-4
        frames.append(img)
    rt.set_render_mode("wavefront")
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])
    d = np.abs(img - ref).max(axis=2)
    if with_glass:
        assert rmse < 5e-5 and (d > 1e-9).mean() < 0.01 and np.median(np.abs(img - ref)) < 1e-12, rmse
    else:
        assert rmse < 1e-9, rmse
    assert float(img.mean()) > 1e-3      # the lights do reach the scene


@pytest.mark.parametrize("with_photons", [False, True])
def test_wavefront_small_pool_many_rounds(setup, with_photons):
    """Few path slots -> many rounds of few samples each: the same frame as with one big round.  With and without a photon map: without one the
    shade stage's compaction job has an unused (gather) queue in the middle, and the free list behind it is what the refills draw from."""
    name, scene, rt, fx = setup
    if with_photons and scene.desc().n_light > 0:
        rt.tracePhotons(3000)
    else:
        rt.clear_photons()
    a = rt.run(40, 30, min_samples=6, max_samples=6)
    rt.set_pool_slots(40 * 32 * 2)      # 2 paths per (padded) pixel
    try:
        b = rt.run(40, 30, min_samples=6, max_samples=6)           # streaming: the pool is refilled ~3x
        rt.set_render_mode("rounds")
        c = rt.run(40, 30, min_samples=6, max_samples=6)           # synchronous rounds of 2 samples
    finally:
        rt.set_pool_slots(1 << 30)
        rt.set_render_mode("wavefront")
    assert np.array_equal(a, b) and np.array_equal(a, c)


@pytest.mark.parametrize("name", ["cornell", "caustics", "teapot"])
def test_pool_refills_without_photon_map(name):
    """Path regeneration from the free list with NO photon map loaded (BASELINE config 2's state, but with a pool smaller than the frame, as on a
    4K x 256 spp frame or on ranks that share a device): slots freed by the shade stage must reach the next pass's free list.  Frames with pools of
    1/3, 1/7 and 1/20 of the samples equal the whole-frame-in-flight frame and the synchronous-rounds frame, bit for bit."""
    scene = pc.load_scene(name)
    rt = gi.RayTracer(0).setScene(scene)          # setScene drops any photon map (gi_upload_scene + gi_clear_photons)
    w, h, spp = 64, 48, 12
    a = rt.run(w, h, min_samples=spp, max_samples=spp)
    assert a.mean() > 1e-3
    try:
        for div in (3, 7, 20):
            rt.set_pool_slots(max(64, w * h * spp // div))
            assert np.array_equal(a.view(np.uint64), rt.run(w, h, min_samples=spp, max_samples=spp).view(np.uint64)), div
        rt.set_render_mode("rounds")
        assert np.array_equal(a.view(np.uint64), rt.run(w, h, min_samples=spp, max_samples=spp).view(np.uint64))
    finally:
        rt.set_pool_slots(1 << 30)
        rt.set_render_mode("wavefront")
    # and the same with a map loaded afterwards, then dropped again: the photon-free job is back
    rt.tracePhotons(2000)
    b = rt.run(w, h, min_samples=spp, max_samples=spp)
    rt.clear_photons()
    rt.set_pool_slots(max(64, w * h * spp // 5))
    try:
        assert np.array_equal(a.view(np.uint64), rt.run(w, h, min_samples=spp, max_samples=spp).view(np.uint64))
    finally:
        rt.set_pool_slots(1 << 30)
    assert b.shape == a.shape


@pytest.mark.parametrize("n", [1, 63, 64, 8191, 8192, 8193, 100003, 2500000, 5 * 8192 * 256 + 777])
def test_own_radix_sort_is_a_stable_sort(rt0, n):
    """gi_sort.inc through gi_debug_sort_pairs: the pipeline's radix sort (gather queries by photon-map leaf, continuing rays by coherence key) on
    random pairs -- the result is the stable sort of the pairs by the selected key bits (so: a permutation, every value exactly once), for the bit
    ranges the pipeline uses (15-17 bits from 0, 27 bits, a raised lowest bit) and for all 32 bits; sizes around the 8 192-pair tile and beyond one
    workgroup's share."""
    rs = np.random.RandomState(n % 1000)
    vals = np.arange(n, dtype=np.uint32)
    for (lo, hi), kmax in (((0, 15), 27000), ((0, 17), 72738), ((0, 27), 2 ** 27), ((6, 27), 2 ** 27), ((0, 32), 2 ** 32), ((0, 3), 8)):
        keys = rs.randint(0, kmax, n, dtype=np.uint64).astype(np.uint32)
        if n > 1000:
            keys[rs.randint(0, n, n // 3)] = keys[0]            # a heavy digit: one bin takes a third of the input
        ko, vo = rt0.sort_pairs(keys, vals, lo, hi)
        sel = (keys.astype(np.uint64) >> lo) & ((1 << (hi - lo)) - 1)
        order = np.argsort(sel, kind="stable")
        assert np.array_equal(vo, vals[order]) and np.array_equal(ko, keys[order]), (n, lo, hi)


def test_gather_resolves_float_key_ties_exactly():
    pc.check_gather_float_ties(lambda: gi.RayTracer(0))


def test_radiance_entry_matches_oracle(setup):
    name, scene, rt, fx = setup
    o = pc.oracle_for(scene)
    ph = np.zeros((0, 9))
    if scene.desc().n_light:
        ph, _ = rt.tracePhotons(2000)
    o.set_photons(ph).build_photon_map()
    rays = fx["rays"][:4000]
    stream = np.arange(len(rays), dtype=np.uint32) * 7919
    got, ref = rt.radiance(rays, stream), o.radiance(rays, stream, rt.seed)
    # A path whose decision sits on a discontinuity (total internal reflection threshold, silhouette of a sphere) may take the
    # other branch when sin/cos/acos differ in the last bit between OCML and glibc: allow a handful of such paths, require
    # all the others to agree to 1e-9 and the whole set to stay far inside the 1e-4 contract
    err = np.abs(got - ref).max(axis=1)
    assert (err > 1e-9).mean() < 2e-3, (err > 1e-9).sum()
    assert np.sqrt(((got - ref) ** 2).mean()) < 1e-5


def test_stripe_sharding_is_exact(setup):
    name, scene, rt, fx = setup
    pc.check_stripes(rt, scene, 64, 50, 2, world=4, stripe_h=8)


def test_empty_and_edge_inputs(setup):
    name, scene, rt, fx = setup
    hit, ent, res = rt.trace(np.zeros((0, 6)))
    assert len(hit) == 0
    assert len(rt.visible(np.zeros((0, 6)))) == 0
    # a ray that starts outside the root box and points away misses; axis-parallel rays (infinite inverse components) work
    away = np.array([[100.0, 100.0, 100.0, 0.0, 1.0, 0.0]])
    assert rt.trace(away)[0][0] == 0
    # 0 samples per pixel: the running mean keeps its initial 0.5 (include/raytracer.h:102)
    img = rt.run(8, 8, min_samples=0, max_samples=0)
    assert (img == 0.5).all()
    # 1x1 frame
    assert rt.run(1, 1, min_samples=1, max_samples=1).shape == (1, 1, 3)


@pytest.mark.parametrize("mode", ["wavefront", "rounds", "megakernel"])
def test_render_with_stochastic_alpha_and_glass_matches_oracle(mode):
    """scenes/spheres/spheres.scn: analytic spheres -- mirror, glass (refraction + Fresnel lobe choice), glossy Phong lobe, a
    half-transparent one (alpha test draws keyed by leaf and entity) -- and non-zero ambient."""
    scene = pc.load_scene("spheres")
    rt = gi.RayTracer(0).setScene(scene)
    rt.set_render_mode(mode)
    rmse, img, ref = pc.check_render(rt, scene, 80, 60, 8, 3000)
    assert rmse < 1e-9 and img.mean() > 0.01


@pytest.mark.parametrize("mode", ["wavefront", "rounds", "megakernel"])
def test_height_fog_render_and_emission_match_oracle(mode):
    """scenes/fog/fog.scn: HeightFog ray-marched on camera segments, shadow rays and photon paths."""
    scene = pc.load_scene("fog")
    rt = gi.RayTracer(0).setScene(scene)
    rt.set_render_mode(mode)
    pc.check_emission(rt, scene, 2000)
    rmse, img, ref = pc.check_render(rt, scene, 80, 60, 8, 2000)
    assert rmse < 1e-9 and img.mean() > 0.005


def test_no_photon_map_gather_is_zero():
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    res, nc = rt.samplePhotons(np.array([[0.0, 0.0, 0.0, 0.0, 1.0, 0.0]]))
    assert (res == 0).all() and nc[0] == 0


def test_sample_chunks_give_the_same_frame(monkeypatch):
    """A frame whose per-sample radiance buffer exceeds the budget (16 GiB by default) is rendered in chunks of samples, each folded
    into the running mean in order: same bits as in one piece.  Here the budget is shrunk to two samples per chunk."""
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    rt.tracePhotons(3000)
    a = rt.run(40, 30, min_samples=7, max_samples=7)
    monkeypatch.setenv("GI_LBUF_MAX_BYTES", str(40 * 30 * 24 * 2))
    rt2 = gi.RayTracer(0).setScene(scene)          # the budget is read when the context is created
    rt2.tracePhotons(3000)
    b = rt2.run(40, 30, min_samples=7, max_samples=7)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_errors_are_reported_not_swallowed():
    rt = gi.RayTracer(0)
    with pytest.raises(gi.GiError):
        rt.trace(np.zeros((1, 6)))            # no scene uploaded
    with pytest.raises(gi.GiError):
        gi.RayTracer(99)                      # no such device


@pytest.mark.parametrize("name,w,h,spp,photons", [("test_scene", 256, 256, 1, 0), ("cornell", 512, 512, 4, 0), ("caustics", 1920, 1080, 2, 20000),
                                                    # 4K: other Halton enumeration constants (2^12 x 3^7), 8.3 M pixels, 17 stripes of 16 rows per rank
                                                    ("caustics", 3840, 2160, 1, 20000)])
def test_full_size_properties(name, w, h, spp, photons):
    """BASELINE frame sizes (configs 1-3) at reduced spp: properties that do not need the oracle."""
    scene = pc.load_scene(name)
    rt = gi.RayTracer(0).setScene(scene)
    if photons and scene.desc().n_light:
        rt.tracePhotons(photons)
    a, na = rt.run(w, h, min_samples=spp, max_samples=spp, want_spp=True)
    assert (na == spp).all() and np.isfinite(a).all()       # (the caustic term col*dot(photon.dir, dir) may be negative, as in the reference)
    b = rt.run(w, h, min_samples=spp, max_samples=spp)
    assert np.array_equal(a, b)                                   # deterministic: counter RNG + Halton, no races
    # stripes of 16 rows over 8 ranks interleave back to the same frame (the multi-GPU decomposition)
    frame = np.zeros_like(a)
    for rank in range(8):
        frame[pc.stripe_rows(h, 16, rank, 8)] = rt.run(w, h, min_samples=spp, max_samples=spp, stripe_h=16, rank=rank, world=8)
    assert np.array_equal(frame, a)
    # float32 output is the rounding of the float64 output
    c = rt.run(w, h, min_samples=spp, max_samples=spp, f64=False)
    assert np.array_equal(c, a.astype(np.float32))
    if name == "test_scene":
        assert (a == 0).all()          # no light, ambient 0, emissive 0: the reference image is black
    # a sub-window of the frame agrees with the oracle (seconds on the CPU)
    if name != "test_scene":
        o = pc.oracle_for(scene)
        ph = scene.photon_tables()["photons"]
        o.set_photons(ph).build_photon_map()
        ref = o.render(w, h, spp, y0=h // 2, y1=h // 2 + 4)["lin"][h // 2:h // 2 + 4]
        assert np.sqrt(((a[h // 2:h // 2 + 4] - ref) ** 2).mean()) < 1e-9


@pytest.mark.parametrize("name,w,h,spp,photons,rows", [
    ("cornell", 512, 512, 64, 0, (101, 256, 380, 490)),            # BASELINE config 2
    ("caustics", 1920, 1080, 256, 200000, (400, 700)),             # BASELINE config 3 (the benchmark frame): 2 rows = 983 k samples on the CPU
    ("teapot", 1920, 1080, 256, 200000, (620, 900)),               # BASELINE config 4 (one GPU's view of the frame; the 8-rank stripes are test_full_size_properties')
])
def test_baseline_configs_at_their_own_spp_match_oracle(name, w, h, spp, photons, rows):
    """BASELINE.json configs 2, 3, 4 rendered ONCE at their own frame size, sample count and photon count; full-width rows spread over the frame
    against the oracle (<= 1 M samples of CPU work each).  Contract: RMSE < 1e-4 on linear radiance (north_star); asserted two orders below it,
    the measured value is printed (config 4 refracts through the glass teapot: a handful of paths take another branch, DESIGN.md "Numerics")."""
    scene = pc.load_scene(name)
    rt = gi.RayTracer(0).setScene(scene)
    o = pc.oracle_for(scene)
    ph = np.zeros((0, 9))
    if photons:
        ph, _ = rt.tracePhotons(photons)
        assert len(ph) > 0.5 * photons
    o.set_photons(ph).build_photon_map()
    img, nspp = rt.run(w, h, min_samples=spp, max_samples=spp, want_spp=True)
    assert (nspp == spp).all() and np.isfinite(img).all()
    rows = np.array(rows, np.int32)
    ref, _ = o.render_rows(w, h, rows, spp, rt.seed, 16)
    d = img[rows] - ref[rows]
    rmse = float(np.sqrt((d ** 2).mean()))
    print(f"{name} {w}x{h} {spp} spp: rmse vs oracle on rows {rows.tolist()} = {rmse:.3e}, max |diff| {np.abs(d).max():.3e}, mean radiance {img.mean():.6f}")
    assert rmse < (1e-5 if name == "teapot" else 1e-8), rmse      # measured (MI355X): cornell 1.2e-15, caustics 1.7e-18, teapot 6.7e-7 (max |diff| 6.1e-5)
    assert np.median(np.abs(d)) < 1e-12


def test_textured_glass_scene_at_256_spp_is_inside_the_contract():
    """scenes/textures/cornell_tex.scn (glass teapot between bright textured walls: the scene where a diverted refraction chain costs most) at the
    sample count the contract is quoted at: RMSE < 1e-4 asserted (at 8 spp the same frame measures 2.3e-4, at 64 spp 4.2e-5)."""
    scene = pc.load_scene("cornell_tex")
    rt = gi.RayTracer(0).setScene(scene)
    rmse, img, ref = pc.check_render(rt, scene, 96, 54, 256, 5000, tol=1e-4)
    print("cornell_tex 96x54 256 spp rmse", rmse, "pixels off", (np.abs(img - ref).max(axis=2) > 1e-9).mean())
    assert np.median(np.abs(img - ref)) < 1e-12


@pytest.mark.parametrize("name,photons", [("cornell", 0), ("caustics", 5000), ("teapot", 3000)])
def test_streaming_work_counters_against_the_reference_counts(name, photons):
    """gi_set_counters(ctx, 2): what the streaming kernels execute.  With content-box culling and entity boxes OFF the walk of k_st_trace meets the leaves of
    Octree::intersectSorted's list in the same order and stops after the same one, so it tests exactly the entities RayTracer::trace tests
    (include/raytracer.h:446-472): its entity tests equal the oracle's T_trace; the any-hit shadow walk stops at the first blocker like the
    reference's candidate loop (include/raytracer.h:290-305), in front-to-back instead of DFS order: T_shadow within a few per cent.  Box tests stay at or below the reference's, which walks the whole
    tree along the ray before it tests anything (include/octree.cpp:188-211,256-313) where the device walks on demand; walks begun equal the
    reference's trace() / visible() calls; gather queries, candidates and shaded hits are equal too.  With culling ON (the product's default) the frame is the same bit for bit and the counters shrink: that ratio is what
    bench.py prints next to the reference's work."""
    scene = pc.load_scene(name)
    rt = gi.RayTracer(0).setScene(scene)
    o = pc.oracle_for(scene)
    ph = np.zeros((0, 9))
    if photons:
        ph, _ = rt.tracePhotons(photons)
    o.set_photons(ph).build_photon_map()
    w, h, spp = 64, 40, 6
    ref = o.render(w, h, spp, want_counters=True)
    oc = ref["counters"]           # v_trace, v_shadow, tri, shaded, pcand, traces, shadows, gathers, tri_shadow
    plain = rt.run(w, h, min_samples=spp, max_samples=spp)
    rt.set_counters("stream")
    try:
        rt.set_content_culling(False)
        assert not rt.set_entity_boxes(False)
        a = rt.run(w, h, min_samples=spp, max_samples=spp)
        c0 = rt.stream_counters()
        assert rt.set_content_culling(True) and rt.set_entity_boxes(True)
        b = rt.run(w, h, min_samples=spp, max_samples=spp)
        c1 = rt.stream_counters()
    finally:
        rt.set_counters(0)
        rt.set_content_culling(True)
        rt.set_entity_boxes(True)
    assert np.array_equal(a.view(np.uint64), plain.view(np.uint64)) and np.array_equal(b.view(np.uint64), plain.view(np.uint64))   # counting changes nothing
    # trace(): one root test per call (a ray that misses the root box -- a camera ray outside the scene -- begins a walk and tests nothing else)
    assert c0["trace_walks"] == oc[5] and c0["trace_rays"] >= oc[5] and c0["trace_walks"] + c0["trace_child_boxes"] <= oc[0], (c0, oc)
    assert c0["trace_tris"] == oc[2] - oc[8]
    # visible(): the reference tests the candidates of the touched leaves in its DFS (child 0..7) order until one blocks, the device meets the leaves
    # front to back -- the same answer (any blocker), about the same number of tests
    assert abs(c0["shadow_tris"] - oc[8]) <= 0.05 * oc[8], (c0["shadow_tris"], oc[8])
    assert c0["shadow_walks"] == oc[6] and c0["shadow_rays"] == oc[6] and c0["shadow_walks"] + c0["shadow_child_boxes"] <= oc[1]
    assert c0["shaded"] == oc[3]
    if photons:      # (without a map the reference still calls samplePhotons, which returns at once; the pipeline has no gather stage then)
        # the reference asks samplePhotons before its roulette decides whether the answer is used (include/raytracer.h:258-267); the pipeline asks only
        # for vertices whose path goes on: never more queries than the reference, and most of them
        assert 0.8 * oc[7] <= c0["gather_queries"] <= oc[7] and 0.8 * oc[4] <= c0["gather_candidates"] <= oc[4], (c0, oc)
    else:
        assert c0["gather_queries"] == 0 and oc[4] == 0
    assert c0["trace_content_boxes"] == 0 and c1["trace_content_boxes"] > 0 and c0["trace_entity_boxes"] == 0 and c1["trace_entity_boxes"] > 0
    # culling on: same rays, same shaded hits and gathers, fewer boxes and entity tests
    for k in ("trace_rays", "shadow_rays", "shaded", "gather_queries", "gather_candidates"):
        assert c1[k] == c0[k], k
    # a closest-hit walk that met two entities at the very same distance is made again the plain way (gi_device.h: trace_wide_over): a few per thousand
    assert c0["trace_walks"] <= c1["trace_walks"] <= 1.01 * c0["trace_walks"], (c0["trace_walks"], c1["trace_walks"])
    assert c1["trace_child_boxes"] < c0["trace_child_boxes"] and c1["trace_tris"] <= c0["trace_tris"] and c1["shadow_tris"] <= c0["shadow_tris"]
    print(name, "executed / reference: boxes %.3f, entity tests %.3f" % ((c1["trace_walks"] + c1["trace_child_boxes"] + c1["shadow_walks"] + c1["shadow_child_boxes"]) / (oc[0] + oc[1]),
                                                                      (c1["trace_tris"] + c1["shadow_tris"]) / oc[2]))


def test_eight_bit_frame_matches_oracle():
    """the display frame -- gamma 2.2, clamp, truncating (int)(255 c) (include/raytracer.h:150-157, include/image.h:14-16) --
    of the product's radiance against the oracle's bytes (whose sink is pinned to the reference's own Image, kat_pixel, and to its run()
    frames).  The caustics frame has pixels with a negative channel (col * dot(photon.dir, dir) < 0): the reference's pow() makes them NaN and
    its cast 0."""
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    o = pc.oracle_for(scene)
    ph, _ = rt.tracePhotons(5000)
    o.set_photons(ph).build_photon_map()
    for kw in (dict(min_samples=4, max_samples=4), dict(min_samples=4, max_samples=16, noise_thresh=0.0015)):
        lin = rt.run(96, 54, **kw)
        ref = o.render(96, 54, kw["min_samples"], kw["max_samples"], kw.get("noise_thresh", 0.0015), want_u8=True)
        assert (lin < 0).any(), "the frame should hold pixels with a negative channel"
        got = gi.to_rgb8(lin)
        assert got[(lin < 0)].max() == 0
        assert np.array_equal(got, ref["u8"])


def test_config5_standin_fog_photons_large_tree():
    """BASELINE config 5 stand-in (sponza.obj is absent upstream): a large procedural tree (tools/big_scene.py), a HeightFog slab
    (include/atmosphere.h:30-83) and a photon map together.  Small frame against the oracle; then the 3840x2160 frame through the
    size-independent properties, a 4-row window against the oracle, and sample index 478 -- the last one a 4K pixel can address before the
    reference's 32-bit Halton index wraps (include/halton_enum.h:41,109-113)."""
    import sys
    sys.path.insert(0, os.path.join(pc.ROOT, "tools"))
    import big_scene
    scene = big_scene.build(40, 80, fog=True)
    assert scene.desc().n_fog == 1 and scene.desc().n_node > 20000
    rt = gi.RayTracer(0).setScene(scene)
    for mode in ("wavefront", "megakernel"):
        rt.set_render_mode(mode)
        rmse, img, ref = pc.check_render(rt, scene, 96, 54, 8, 20000, tol=1e-9)
        rmse_a, _, _ = pc.check_render(rt, scene, 64, 36, 4, 20000, adaptive=True, tol=1e-9)
        print("config5 stand-in", mode, rmse, rmse_a)
    rt.set_render_mode("wavefront")
    assert len(scene.photon_tables()["photons"]) > 5000
    w, h, spp = 3840, 2160, 2
    a, na = rt.run(w, h, min_samples=spp, max_samples=spp, want_spp=True)
    assert (na == spp).all() and np.isfinite(a).all()
    assert np.array_equal(a, rt.run(w, h, min_samples=spp, max_samples=spp))                 # deterministic
    frame = np.zeros_like(a)
    for rank in range(8):
        frame[pc.stripe_rows(h, 16, rank, 8)] = rt.run(w, h, min_samples=spp, max_samples=spp, stripe_h=16, rank=rank, world=8)
    assert np.array_equal(frame, a)                                                           # the 8-GPU decomposition
    o = pc.oracle_for(scene)
    o.set_photons(scene.photon_tables()["photons"]).build_photon_map()
    rows = slice(h // 2 + 200, h // 2 + 204)
    ref = o.render(w, h, spp, y0=rows.start, y1=rows.stop)["lin"][rows]
    assert np.sqrt(((a[rows] - ref) ** 2).mean()) < 1e-9
    # samples 477 and 478 of a 4K pixel (the last addressable ones) and 479.. (wrapped, as in the reference): the index arithmetic is pinned
    # bit-exactly by the halton fixture (test_halton_device_tables); here the radiance of those samples against the oracle
    rays, idx = [], []
    for s_ in (477, 478, 479, 1023):
        for (x, y) in ((0, 0), (1917, 1083), (3839, 2159)):
            i, r = o.primary_ray(w, h, s_, x, y)
            rays.append(r); idx.append(i)
    got, want = rt.radiance(np.array(rays), np.array(idx, np.uint32)), o.radiance(np.array(rays), np.array(idx, np.uint32), rt.seed)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)


def _tables_equal(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ("nodes", "ranges", "pos", "dircol"))


@pytest.mark.parametrize("name", ["cornell", "caustics", "caustics_02"])
def test_device_photon_map_build_equals_host_builder(name, golden):
    """PhotonMap::rebuild / Node::partition (include/photonMap.cpp:33-47,137-192) on the device (gi_build_photon_map): the tables
    the gather kernel reads -- node records, candidate ranges, photons in leaf order -- equal, byte for byte, the ones gi_upload_photons derives
    from the host builder's tree (itself node for node the reference's, tests/test_host_builders.py), on the reference's own photon sets."""
    fx = golden("scene_" + name)
    scene = pc.load_scene(name)
    rt = gi.RayTracer(0).setScene(scene)
    scene.build_photon_map(fx["photons"])
    rt.upload_photon_map()
    host = rt.photon_tables_on_device()
    res_h, nc_h = rt.samplePhotons(fx["gather_q"])
    rt.build_photon_map_on_device(fx["photons"])
    dev = rt.photon_tables_on_device()
    assert len(host["nodes"]) > 100 and _tables_equal(host, dev)
    res_d, nc_d = rt.samplePhotons(fx["gather_q"])
    assert np.array_equal(nc_d, fx["gather_ncand"]) and np.array_equal(res_d.view(np.uint64), res_h.view(np.uint64))


def test_device_photon_map_build_edge_cases_and_size():
    """Few photons (the root stays a leaf), photons outside the box (no child contains them: dropped at the first split), a cluster that splits
    deep, a million photons; and emission + build without leaving the device (gi_trace_photons) against the host hop."""
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    bb = scene.tables()["node_bbox"][0]
    rs = np.random.RandomState(11)

    def photons(n, spread=1.0, centre=None):
        c = (bb[:3] + bb[3:]) / 2 if centre is None else centre
        pos = c + (rs.rand(n, 3) - 0.5) * (bb[3:] - bb[:3]) * spread
        d = rs.randn(n, 3); d /= np.linalg.norm(d, axis=1)[:, None]
        return np.concatenate([pos, d, rs.rand(n, 3)], 1)

    cases = {"few": photons(10), "seventeen": photons(17), "outside": photons(3000, spread=1.6), "cluster": np.concatenate([photons(2000), photons(4000, spread=1e-4)]),
             "million": photons(1000000)}
    for label, ph in cases.items():
        scene.build_photon_map(ph)
        rt.upload_photon_map()
        host = rt.photon_tables_on_device()
        rt.build_photon_map_on_device(ph)
        dev = rt.photon_tables_on_device()
        assert _tables_equal(host, dev), label
    assert len(host["nodes"]) > 100000                       # the million-photon tree
    # an empty set: no map, the gather returns 0
    rt.build_photon_map_on_device(np.zeros((0, 9)))
    assert rt.samplePhotons(np.array([[0.0, 0.1, 0.0, 0, 1, 0]]))[0].max() == 0
    # emission on the device + build on the device == emission, host build, upload
    ph, tries = rt.tracePhotons(20000)
    host = rt.photon_tables_on_device()
    n, tries_d = rt.tracePhotonsOnDevice(20000)
    assert n == len(ph) and tries_d == tries and _tables_equal(host, rt.photon_tables_on_device())
    img_a = rt.run(64, 36, min_samples=4, max_samples=4)
    rt.tracePhotons(20000)
    assert np.array_equal(img_a, rt.run(64, 36, min_samples=4, max_samples=4))


def test_group_of_contexts_renders_the_single_context_frame():
    """gi_group_* (include/gi_hip.h): one process, several contexts, one host thread each, stripes dealt round-robin and gathered -- the C path a
    C++ caller gets on a multi-GPU node.  On this one-GPU box the group holds three contexts on device 0: the frame must equal the single
    context's bit for bit, for the whole frame (host gather and device gather) and for a window of stripes (progressive display)."""
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    ph, _ = rt.tracePhotons(3000)
    w, h, spp = 96, 70, 4
    full = rt.run(w, h, min_samples=spp, max_samples=spp)
    grp = gi.RayTracerGroup([0, 0, 0]).setScene(scene, ph)
    p = rt.params(w, h, min_samples=spp, max_samples=spp)
    assert np.array_equal(grp.run(p, stripe_h=16), full)                    # 5 stripes (the last one 6 rows) over 3 contexts
    assert np.array_equal(grp.run(p, stripe_h=8, f64=False), full.astype(np.float32))
    win = grp.run(p, stripe_h=16, first_stripe=1, n_stripes=2)              # rows 16..47 only
    assert np.array_equal(win[16:48], full[16:48]) and (win[:16] == 0).all() and (win[48:] == 0).all()
    # the device gather: a frame buffer on device 0 from the HIP runtime the library itself uses (no second runtime in this process)
    import ctypes
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    d_buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_buf), ctypes.c_size_t(full.nbytes)) == 0
    try:
        assert hip.hipMemset(d_buf, 0, ctypes.c_size_t(full.nbytes)) == 0
        grp.run_device(p, d_buf.value, stripe_h=16, f64=True)
        got = np.zeros_like(full)
        assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_buf, ctypes.c_size_t(full.nbytes), 2) == 0   # hipMemcpyDeviceToHost
    finally:
        hip.hipFree(d_buf)
    assert np.array_equal(got, full)
    # adaptive sampling through the group as well
    pa = rt.params(w, h, min_samples=4, max_samples=16, noise_thresh=0.0015)
    assert np.array_equal(grp.run(pa, stripe_h=16), rt.run(w, h, min_samples=4, max_samples=16, noise_thresh=0.0015))


def test_headless_cli_writes_the_display_frame(tmp_path):
    """python -m gi_raytracer_amd scene.scn -o out.ppm: the reference's main.cpp + Viewer flow without Qt; the PPM holds the display
    transform of the frame the API returns for the same settings."""
    import subprocess
    import sys
    out, pfm = tmp_path / "f.ppm", tmp_path / "f.pfm"
    r = subprocess.run([sys.executable, "-m", "gi_raytracer_amd", os.path.join(pc.ROOT, "scenes/caustics/caustics.scn"), "-o", str(out), "--pfm", str(pfm),
                        "--width", "96", "--height", "54", "--samples", "4", "8", "--photons", "3000"], cwd=pc.ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    hdr = b"P6\n96 54\n255\n"
    assert raw.startswith(hdr)
    img = np.frombuffer(raw[len(hdr):], np.uint8).reshape(54, 96, 3)
    scene = pc.load_scene("caustics")
    rt = gi.RayTracer(0).setScene(scene)
    rt.tracePhotons(3000)
    lin = rt.run(96, 54, f64=False, min_samples=4, max_samples=8)
    assert np.array_equal(img, gi.to_rgb8(lin)) and "Msamples/s" in r.stdout


_RCCL_CHILD = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
import numpy as np, torch, torch.distributed as dist
import gi_raytracer_amd as gi, parity_checks as pc
from gi_raytracer_amd.sharding import STRIPE_H, FrameGather
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)            # nccl IS RCCL on ROCm
assert dist.get_backend() == "nccl"
scene = pc.load_scene("caustics"); rt = gi.RayTracer(0).setScene(scene); rt.set_stream(torch.cuda.current_stream().cuda_stream)
rt.tracePhotonsOnDevice(3000)
w, h, spp = 96, 70, 4
p = rt.params(w, h, stripe_h=STRIPE_H, rank=0, world=1, min_samples=spp, max_samples=spp)
fg = FrameGather(torch, dist, w, h, STRIPE_H, 0, 1, dev, torch.float32, always_collective=True)
rt.run_device(p, fg.local.data_ptr(), f64=False)
frame = fg.gather()                                                                 # dist.gather of device tensors over RCCL
dist.barrier(); torch.cuda.synchronize()
tt = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks of its timings
ref = rt.run(w, h, f64=False, min_samples=spp, max_samples=spp)
print(json.dumps({"equal": bool(np.array_equal(frame.cpu().numpy(), ref)), "mean": float(ref.mean()), "tt": tt.tolist(), "collectives": fg.n_collectives}))
dist.destroy_process_group()
"""


def test_rccl_gather_path_runs_on_one_rank(tmp_path):
    """The RCCL branch of the multi-GPU path (bench.py: init_process_group("nccl", device_id=...), sharding.FrameGather.gather on device tensors,
    the all-reduce of the timings) executed once on this one-GPU box: world_size 1, in a fresh child process that initialises its group before
    any other GPU work.  The gathered frame equals the plain single-GPU frame."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(script), pc.ROOT, str(port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["equal"] and out["mean"] > 1e-3 and out["tt"] == [1.5, 2.5] and out["collectives"] == 1
