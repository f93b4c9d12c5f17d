"""Kernel logic on the CPU: the per-lane device functions of gi_raytracer_amd/csrc/gi_device.h compiled for the host
(tests/host_emul) and checked against the reference's golden tables and against the oracle.  This is NOT the product
path (the product has no CPU path); it exists so that logic errors are caught where there is no GPU."""
import numpy as np
import pytest

import emul_lib as el
import parity_checks as pc


@pytest.fixture(scope="module", params=["test_scene", "cornell", "caustics", "spheres_opaque", "textures_opaque", "caustics_02"])
def setup(request, golden):
    scene = pc.load_scene(request.param)
    return request.param, scene, el.EmulRayTracer().setScene(scene), golden("scene_" + request.param.replace("_opaque", ""))


def test_halton_device_tables(golden):
    pc.check_halton(el.EmulRayTracer(), golden)


def test_counter_rng_matches_oracle():
    import oracle_lib as ol
    L, E = ol.lib(), el.lib()
    rs = np.random.RandomState(1)
    for _ in range(200):
        a = [int(v) for v in rs.randint(0, 2**31, 6)]
        assert L.gio_counter_rand(a[0] * 7919, a[1], a[2] % 70, a[3] % 4096, a[4], a[5]) == E.emul_rng(a[0] * 7919, a[1], a[2] % 70, a[3] % 4096, a[4], a[5])


def test_counter_rng_known_answers_through_kat():
    """SURVEY 8 a-16 on the CPU build: the same lattice the GPU test runs through gi_kat (what = 8)."""
    assert pc.check_rng(el.EmulRayTracer()) > 10000


def test_trace_matches_reference_table(setup):
    pc.check_trace_table(setup[2], setup[3])


def test_leaf_sequence_matches_reference_sorted_lists(setup):
    print(pc.check_leaf_order(setup[2], setup[3], setup[2].set_wide_nodes))


def test_sampler_known_answers(golden):
    rt = el.EmulRayTracer()
    rt.libm_is_glibc = True          # this build calls the same glibc the reference was run with: bit-exact
    pc.check_kats(rt, golden)


def test_visible_matches_reference_table(setup):
    pc.check_visible_table(setup[2], setup[3])


def test_turn_wise_shadow_walk_answers_like_visible(setup):
    """k_st_shadow walks a segment one turn at a time (wwalk_turn) and tests a leaf when it stands on one (visible_leaf_blocks): the same
    answers as RayTracer::visible's loop, on the fixture's shadow segments and on the adversarial rays (axis-parallel, origins on planes)."""
    name, scene, rt, fx = setup
    rays = pc.adversarial_rays(scene, n=3000, seed=9)
    q = np.concatenate([rays[:, :3], rays[:, :3] + rays[:, 3:6] * 4.0], axis=1)
    if "shadow_q" in fx:
        q = np.concatenate([q, fx["shadow_q"][:3000]])
    turns = rt.visible_turns(q)
    if turns is None:
        pytest.skip("no wide records for this tree")
    assert np.array_equal(turns, rt.visible(q))
    assert 0 < turns.sum() < len(turns)
    # segments that end at a light, from surface points and from anywhere: k_st_shadow's own boxes (cut to the leaves where no entity comes near a light)
    lights = scene.tables()["lights"].reshape(-1, 11)
    if len(lights):
        hit, _, res = rt.trace(rays)
        org = np.concatenate([res[hit > 0, :3] + 1e-4 * res[hit > 0, 3:6], rays[:, :3]])
        rs = np.random.RandomState(4)
        u = rs.randn(len(org), 3)
        u /= np.linalg.norm(u, axis=1)[:, None]
        ql = np.concatenate([org, lights[0, :3] + lights[0, 6] * u], axis=1)
        lb = rt.visible_turns(ql, light_bound=True)
        assert np.array_equal(lb, rt.visible(ql))
        assert 0 < lb.sum() < len(lb)


def test_wide_walk_on_a_deeper_tree():
    scene = pc.load_scene("teapot")
    rt = el.EmulRayTracer().setScene(scene)
    pc.check_wide_walk(rt, scene, rt.set_wide_nodes)


def test_wide_walk_equals_per_node_walk(setup):
    pc.check_wide_walk(setup[2], setup[1], setup[2].set_wide_nodes)


def test_hits_at_equal_distance_go_to_the_entity_met_first():
    scene = pc.coplanar_scene()
    rt = el.EmulRayTracer().setScene(scene)
    pc.check_equal_distance_hits(rt, scene, rt.set_wide_nodes)
    pc.check_wide_walk(rt, scene, rt.set_wide_nodes)
    pc.check_render(rt, scene, 48, 32, 8, 0)        # and a frame against the oracle, which asks the entities in the reference's order


def test_content_culling_changes_nothing(setup):
    pc.check_content_culling(setup[2], setup[1], render=setup[0] in ("caustics", "spheres_opaque"))


def test_photon_octree_descent_variants_agree():
    scene = pc.load_scene("caustics")
    pc.check_photon_descent(el.EmulRayTracer().setScene(scene), scene)


def test_gather_matches_reference_table(setup):
    name, scene, rt, fx = setup
    if "photons" not in fx:
        pytest.skip("no photons in this scene")
    pc.check_gather_table(rt, scene, fx)


def test_emission_identical_to_oracle(setup):
    name, scene, rt, fx = setup
    if scene.desc().n_light == 0:
        pytest.skip("no light")
    pc.check_emission(rt, scene, 1500)


@pytest.mark.parametrize("adaptive", [False, True])
def test_render_matches_oracle(setup, adaptive):
    name, scene, rt, fx = setup
    rmse, _, _ = pc.check_render(rt, scene, 40, 24, 4, 2000, adaptive)
    assert rmse < 1e-12


def test_stripe_sharding_is_exact(setup):
    name, scene, rt, fx = setup
    pc.check_stripes(rt, scene, 24, 21, 2, world=3, stripe_h=4)


def test_gather_resolves_float_key_ties_exactly():
    pc.check_gather_float_ties(el.EmulRayTracer)


def test_render_with_stochastic_alpha_and_glass_matches_oracle():
    """scenes/spheres/spheres.scn: mirror, glass (refraction + Fresnel lobe choice), glossy Phong lobe, a half-transparent sphere
    (alpha test draws keyed by leaf and entity), non-zero ambient."""
    scene = pc.load_scene("spheres")
    rt = el.EmulRayTracer().setScene(scene)
    rmse, img, ref = pc.check_render(rt, scene, 40, 30, 6, 1500)
    assert rmse < 1e-12 and img.mean() > 0.01


def test_textured_scene_render_and_emission_match_oracle():
    """scenes/textures/tex.scn: checkerboard and PNG textures on meshes and spheres as diffuse and emissive colour, an alpha channel
    and a 0.7 opacity in the stochastic alpha test of trace / visible / rayType (include/material.h:32-93)."""
    scene = pc.load_scene("textures")
    rt = el.EmulRayTracer().setScene(scene)
    pc.check_emission(rt, scene, 600)
    rmse, img, ref = pc.check_render(rt, scene, 40, 30, 6, 1500)
    assert rmse < 1e-12 and img.mean() > 0.01
    for mode in ("adaptive",):
        pc.check_render(rt, scene, 32, 24, 4, 1500, adaptive=True)


def test_height_fog_render_and_emission_match_oracle():
    """scenes/fog/fog.scn: HeightFog ray-marched on camera segments, shadow rays and photon paths (include/raytracer.h:209-228,
    308-316, 658-675); every 0.04 step draws from the keyed RNG."""
    scene = pc.load_scene("fog")
    assert scene.desc().n_fog == 1
    rt = el.EmulRayTracer().setScene(scene)
    pc.check_emission(rt, scene, 800)
    rmse, img, ref = pc.check_render(rt, scene, 32, 24, 4, 800)
    assert rmse < 1e-12
    # the medium changes the picture: the same scene without fog is different
    t = scene.tables()
    assert len(t["fog_grid"]) == 6 * 2 * 6 * 64


def test_two_lights_match_oracle():
    """Two lights (no reference scene has more than one): the last visible light's share wins, photons come from both -- the CPU build of the
    device code against the oracle, with the glass block (glibc on both sides: no libm chaos)."""
    scene = pc.two_light_scene(True)
    rt = el.EmulRayTracer().setScene(scene)
    pc.check_emission(rt, scene, 1000)
    rmse, img, ref = pc.check_render(rt, scene, 48, 27, 4, 2000)
    assert rmse < 1e-12 and img.mean() > 1e-3
