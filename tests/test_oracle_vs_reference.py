"""Pins the CPU oracle (oracle/gi_oracle.cpp) to the reference.

Every expected value in tests/golden/*.npz was produced by the UNMODIFIED reference sources, compiled and driven
by oracle/ref/ref_driver.cpp (regenerate with `python oracle/ref/gen_fixtures.py` in the build container).
All comparisons are bit-exact: the oracle restates the reference's arithmetic in the reference's operand order.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

SCENES = ["test_scene", "cornell", "caustics", "spheres", "textures", "caustics_02"]


def P(a):
    return np.ascontiguousarray(a, np.float64).ctypes.data_as(ol.c_dp)


# ------------------------------------------------------------------ Halton
def test_halton_enum_params_and_index(golden):
    g, L = golden("halton"), ol.lib()
    for w, h, p2, p3, mx, my, inc in g["halton_enum_params"]:
        out = np.zeros(5, np.uint32)
        L.gio_halton_enum_params(int(w), int(h), out.ctypes.data_as(ol.c_up))
        assert list(out) == [p2, p3, mx, my, inc]
    sizes = g["halton_enum_params"][:, :2]
    for k, s, x, y, idx in g["halton_enum_index"]:
        w, h = sizes[k]
        assert L.gio_halton_index(int(w), int(h), int(s), int(x), int(y)) == idx
    for k, v, sx, sy in g["halton_enum_scale"]:
        w, h = sizes[int(k)]
        assert np.float32(L.gio_halton_scale(int(w), int(h), 0, C.c_float(v))) == sx
        assert np.float32(L.gio_halton_scale(int(w), int(h), 1, C.c_float(v))) == sy


def test_halton_sampler_all_256_dims_bit_exact(golden):
    g, L = golden("halton"), ol.lib()
    idxs, ref = g["halton_sample_idx"], g["halton_sample"]
    got = np.array([[L.gio_halton_sample(d, int(i)) for i in idxs] for d in range(256)], np.float32)
    assert got.tobytes() == ref.tobytes()


# ------------------------------------------------------------------ util KATs
def test_pow_hacks(golden):
    k, L = golden("kat")["kat_pow"], ol.lib()
    assert np.array_equal([L.gio_fast_pow(a, b) for a, b in k[:, :2]], k[:, 2])
    assert np.array_equal([L.gio_fast_precise_pow(a, b) for a, b in k[:, :2]], k[:, 3])


def test_samplers_bit_exact(golden):
    k, L = golden("kat"), ol.lib()
    o = np.zeros(3)
    for r in k["kat_hemi"]:
        L.gio_hemi_cos_n(P(r[0:3]), C.c_float(r[3]), C.c_float(r[4]), r[5], P(o)); assert o.tobytes() == r[6:9].tobytes()
        L.gio_hemi_cos(C.c_float(r[3]), C.c_float(r[4]), r[5], o.ctypes.data_as(ol.c_dp)); assert o.tobytes() == r[9:12].tobytes()
    for r in k["kat_phong"]:
        L.gio_sample_phong(P(r[0:3]), P(r[3:6]), r[6], r[7], r[8], o.ctypes.data_as(ol.c_dp)); assert o.tobytes() == r[9:12].tobytes()
    for r in k["kat_cap"]:
        L.gio_sphere_cap(P(r[0:3]), C.c_float(r[3]), C.c_float(r[4]), r[5], r[6], o.ctypes.data_as(ol.c_dp)); assert o.tobytes() == r[7:10].tobytes()
    for r in k["kat_unitvec"]:
        L.gio_unit_vec(r[0], r[1], o.ctypes.data_as(ol.c_dp)); assert o.tobytes() == r[2:5].tobytes()
    for r in k["kat_refr"]:
        L.gio_refr(P(r[0:3]), P(r[3:6]), r[6], o.ctypes.data_as(ol.c_dp)); assert o.tobytes() == r[7:10].tobytes()
        L.gio_reflect(P(r[0:3]), P(r[3:6]), o.ctypes.data_as(ol.c_dp)); assert o.tobytes() == r[10:13].tobytes()


def test_pixel_sink_known_answers(golden):
    """a-19: gamma 2.2 + glm::clamp + Image::setPixel as the reference's own Image stored them (kat_pixel: negative channels, values above
    1, values one ulp either side of the k/255 thresholds) -- for the oracle's sink and for the product's host-side sink gih_to_rgb8."""
    import gi_raytracer_amd as gi
    k, L = golden("kat")["kat_pixel"], ol.lib()
    lin = np.ascontiguousarray(k[:, :3]).reshape(-1)
    out = np.zeros(len(lin), np.uint8)
    L.gio_pixel8(len(lin), lin.ctypes.data_as(ol.c_dp), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert np.array_equal(out.reshape(-1, 3), k[:, 3:].astype(np.uint8))
    assert (k[:, :3] < 0).any() and (k[:, :3] > 1).any()
    assert np.array_equal(gi.to_rgb8(k[:, :3]), k[:, 3:].astype(np.uint8))


def test_tri_box_overlap(golden):
    k, L = golden("kat")["kat_tribox"], ol.lib()
    got = [L.gio_tri_box_overlap(P(r[0:3]), P(r[3:6]), P(r[6:15])) for r in k]
    assert np.array_equal(got, k[:, 15].astype(int))


def test_xorshift_chain(golden):
    k, L = golden("kat")["kat_drand"], ol.lib()
    st = C.c_uint64(1500000000)
    assert np.array_equal([L.gio_chain_drand(C.byref(st)) for _ in range(len(k))], k)


# ------------------------------------------------------------------ scene functions (SURVEY 8 a-3 .. a-13)
@pytest.fixture(scope="module", params=SCENES)
def scene(request, golden):
    fx = golden("scene_" + request.param)
    return fx, ol.Oracle.from_fixture(fx).build_octree()


def test_octree_build_identical(scene):
    fx, o = scene
    assert np.array_equal(o.ent_bbox(), fx["tri_bbox"])
    bbox, child, off, idx = o.octree()
    assert np.array_equal(bbox, fx["oct_bbox"]) and np.array_equal(child, fx["oct_child"])
    assert np.array_equal(off, fx["oct_ent_off"]) and np.array_equal(idx, fx["oct_ent_idx"])
    if len(fx["lights"]):
        assert np.array_equal(o.lights_dir_angle(), fx["lights"][:, 7:11])


def test_trace_identical(scene):
    fx, o = scene
    hit, ent, res, nl = o.trace(fx["rays"])
    assert np.array_equal(hit, fx["trace_hit"]) and np.array_equal(ent, fx["trace_ent"])
    assert np.array_equal(res, fx["trace_res"]) and np.array_equal(nl, fx["trace_nleaves"])


def test_sorted_leaf_order_identical(scene):
    fx, o = scene
    for j, r in enumerate(fx["leaforder_ray"]):
        node, t0 = o.leaf_order(fx["rays"][r])
        a, b = fx["leaforder_off"][j], fx["leaforder_off"][j + 1]
        assert np.array_equal(node, fx["leaforder_node"][a:b]) and np.array_equal(t0, fx["leaforder_t0"][a:b])


def test_visible_identical(scene):
    fx, o = scene
    vis, nc = o.visible(fx["shadow_q"])
    assert np.array_equal(vis, fx["shadow_vis"]) and np.array_equal(nc, fx["shadow_ncand"])


def test_photon_octree_and_gather_identical(scene):
    fx, o = scene
    if "photons" not in fx:
        pytest.skip("scene has no light / photons")
    o.set_photons(fx["photons"]).build_photon_map()
    b, fc, off, idx = o.pmap()
    assert np.array_equal(b, fx["pm_bbox"]) and np.array_equal(fc, fx["pm_firstchild"])
    assert np.array_equal(off, fx["pm_off"]) and np.array_equal(idx, fx["pm_idx"])
    res, nc = o.gather(fx["gather_q"])
    assert np.array_equal(nc, fx["gather_ncand"]) and np.array_equal(res, fx["gather_res"])
    assert (nc == 0).any() and (nc > 32).any() and ((nc > 0) & (nc < 32)).any()  # edge cases are covered


def test_textures_get_and_alpha_bit_exact(golden):
    """texture::get / getAlpha of every texture of scenes/textures/tex.scn (colour, checkerboard, PNG with and without alpha channel) on a
    uv lattice that leaves [0, 1] on both sides -- the reference's own answers, image pixels as its QImage returned them."""
    fx = golden("scene_textures")
    o = ol.Oracle.from_fixture(fx)
    assert set(fx["tex_kind"].tolist()) == {0, 1, 2}
    for t in range(len(fx["tex_kind"])):
        k = fx["tex_kat"][t]
        assert np.array_equal(o.tex_eval(t, k[:, :2]), k[:, 2:6]), t
    alpha = fx["tex_kat"][np.flatnonzero(fx["tex_param"][:, 4] == 1)[0]][:, 5]
    assert (alpha == 0).any() and (alpha == 1).any()   # partial alpha (128/255) is in tex.scn, pinned by the chain_textures_* frames


# ------------------------------------------------------------------ whole frames on the pinned RNG chain (a-1, a-2, a-10, f1)
@pytest.mark.parametrize("name", ["chain_test_scene_lin", "chain_caustics_lin", "chain_cornell_lin", "chain_caustics_run", "chain_cornell_run",
                                  "chain_spheres_lin", "chain_spheres_run", "chain_fog_lin", "chain_fog_run",
                                  "chain_textures_lin", "chain_textures_run", "chain_caustics_02_lin", "chain_caustics_02_run"])
def test_whole_frame_matches_reference_bit_for_bit(golden, name):
    """The reference's frame (its own RayTracer::run for *_run; radiance() per sample for *_lin) on a pinned time() and
    one OpenMP thread, including tracePhotons and the photon-map build, reproduced by the oracle's chain RNG mode."""
    fx = golden(name)
    W, H, spp, nph, T, isrun = [int(v) for v in fx["chain_meta"]]
    o = ol.Oracle.from_fixture(fx).build_octree().chain_seed(T)
    if "fog_grid" in fx:
        o.chain_discard(len(fx["fog_grid"]))      # the HeightFog constructor drew its noise grid from the chain at load time
    if nph > 0 and len(fx["lights"]):
        n, _ = o.emit_photons(nph, 5, ol.RNG_CHAIN)
        o.build_photon_map()
        # photons outside the half-open root box are emitted but sit in no leaf (textures scene: 410 of 1500)
        assert n >= len(fx["chain_photons_leaforder"]) and len(o.pmap()[3]) == len(fx["chain_photons_leaforder"])
    o.build_photon_map()
    r = o.render(W, H, spp, rng_mode=ol.RNG_CHAIN, chain_predraws=2 if isrun else 0, want_u8=True)
    if isrun:
        assert np.array_equal(r["u8"], fx["chain_run_u8"])
    else:
        assert np.array_equal(r["lin"], fx["chain_lin"])
    assert (r["spp"] == spp).all()
