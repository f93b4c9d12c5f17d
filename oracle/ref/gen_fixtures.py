#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference (build container only).

Runs oracle/_ref/ref_driver (built by oracle/ref/Makefile from /root/reference's own sources) and packs its
.npy outputs into one compressed .npz per fixture.  The .npz files hold data only (inputs and the reference's
outputs); no reference source enters the repository.  Re-run:  python oracle/ref/gen_fixtures.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
GOLDEN = os.path.join(ROOT, "tests", "golden")

SCENES = {
    # name: (scene file, W, H, photons)           SURVEY.md section 8 configs 1-3 at fixture size
    "test_scene": (f"{REF}/examples/test_scene/test.scn", 96, 96, 0),
    "cornell": (f"{REF}/scenes/cornell/test.scn", 64, 48, 5000),
    "caustics": (f"{REF}/scenes/caustics/caustics.scn", 64, 48, 20000),
    # our own scene file with the `sphere` keyword (analytic spheres: mirror, glass, glossy), run through the reference
    "spheres": (f"{ROOT}/scenes/spheres/spheres_opaque.scn", 48, 36, 4000),
    # checkerboard and image textures (with and without alpha), on meshes and on spheres; also pins texture::get / getAlpha on a uv lattice
    "textures": (f"{ROOT}/scenes/textures/tex_opaque.scn", 48, 36, 1500),
    # the reference's second caustics scene: a glass sphere MESH (IOR 1.5, smooth normals), a metal mesh, ambient light
    "caustics_02": (f"{REF}/scenes/caustics_02/caustics.scn", 64, 48, 6000),
}
CHAINS = {
    # name: (scene, W, H, spp, photons, mode)
    "chain_caustics_run": (f"{REF}/scenes/caustics/caustics.scn", 48, 27, 4, 3000, "run"),
    "chain_caustics_lin": (f"{REF}/scenes/caustics/caustics.scn", 48, 27, 4, 3000, "lin"),
    "chain_cornell_lin": (f"{REF}/scenes/cornell/test.scn", 32, 32, 4, 2000, "lin"),
    "chain_cornell_run": (f"{REF}/scenes/cornell/test.scn", 32, 32, 4, 2000, "run"),
    "chain_test_scene_lin": (f"{REF}/examples/test_scene/test.scn", 32, 32, 2, 0, "lin"),
    # all four spheres, one of them half transparent (stochastic alpha test: exact only on the pinned chain)
    "chain_spheres_lin": (f"{ROOT}/scenes/spheres/spheres.scn", 40, 30, 4, 1500, "lin"),
    "chain_spheres_run": (f"{ROOT}/scenes/spheres/spheres.scn", 40, 30, 4, 1500, "run"),
    # HeightFog: ray-marched medium on camera segments, shadow rays and photon paths (scenes/fog/fog.scn, our scene file)
    "chain_fog_lin": (f"{ROOT}/scenes/fog/fog.scn", 40, 30, 4, 1500, "lin"),
    "chain_fog_run": (f"{ROOT}/scenes/fog/fog.scn", 40, 30, 4, 1500, "run"),
    "chain_textures_lin": (f"{ROOT}/scenes/textures/tex.scn", 40, 30, 4, 1500, "lin"),
    "chain_textures_run": (f"{ROOT}/scenes/textures/tex.scn", 40, 30, 4, 1500, "run"),
    "chain_caustics_02_lin": (f"{REF}/scenes/caustics_02/caustics.scn", 40, 30, 4, 2000, "lin"),
    "chain_caustics_02_run": (f"{REF}/scenes/caustics_02/caustics.scn", 40, 30, 4, 2000, "run"),
}


def pack(tmp, name):
    arrs = {}
    for f in sorted(os.listdir(tmp)):
        if f.endswith(".npy"):
            arrs[f[:-4]] = np.load(os.path.join(tmp, f))
    out = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(out, **arrs)
    print(f"{name}: {len(arrs)} arrays, {os.path.getsize(out) / 1024:.0f} KiB")


def run(args, tmp):
    with open(os.path.join(tmp, "log.txt"), "w") as log:
        subprocess.run([DRIVER] + [str(a) for a in args], check=True, stdout=log, stderr=subprocess.STDOUT)


def main():
    if not os.path.isdir(REF):
        sys.exit("no /root/reference here: fixtures can only be regenerated in the build container")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "ref")], check=True)
    os.makedirs(GOLDEN, exist_ok=True)
    only = set(sys.argv[1:])
    for name, cmd in (("halton", "halton"), ("kat", "kat")):
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as tmp:
            run([cmd, tmp], tmp)
            pack(tmp, name)
    for name, (scn, w, h, ph) in SCENES.items():
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as tmp:
            run(["scene", scn, tmp, w, h, ph], tmp)
            pack(tmp, "scene_" + name)
    for name, (scn, w, h, spp, ph, mode) in CHAINS.items():
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as tmp:
            run(["chain", scn, tmp, w, h, spp, ph, mode], tmp)
            pack(tmp, name)


if __name__ == "__main__":
    main()
