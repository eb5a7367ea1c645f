#!/usr/bin/env python3
"""Writes tests/golden/converged_<scene>_64x64_1024spp.npy: the strict CPU oracle's radiance sums of a
64 x 64 rendering at 1024 samples per pixel (per-path libstdc++ streams, seed 0), for the image-level
"converged rendering on matched random streams" tests:

    python oracle/ref_harness/converged_fixture.py

The CPU test recomputes the files bit for bit (the oracle must not drift); the GPU test renders the
same samples and compares in the display domain.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

SIZE, SPP = 64, 1024
SCENES = {"cornell": (0, False, None), "glasscube": (2, True, None),
          "glasscubesphere": (2, True, (0.3, [-0.45, 0.7, -0.55, 1], [1, 0, 0]))}


def render(name, threads=8):
    import __graft_entry__ as g
    abi = g.load_package().abi
    which, glass, sph = SCENES[name]
    tris = O.load_test_model(which)
    spheres = None
    if sph:
        spheres = np.zeros(1, dtype=abi.SPHERE_DTYPE)
        spheres["radius"], spheres["center"], spheres["color"], spheres["material"] = sph[0], sph[1], sph[2], abi.MAT_GLASS
    cam, light = O.default_camera(SIZE), O.default_light(glass)
    p = O.default_params(SIZE, SIZE, rng_mode=abi.RNG_MINSTD_PATH, seed=0)
    acc, _ = O.render_accumulate(tris, spheres, cam, light, p, 0, SPP, threads=threads)
    return acc


if __name__ == "__main__":
    for name in SCENES:
        acc = render(name)
        path = os.path.join(ROOT, "tests", "golden", f"converged_{name}_64x64_1024spp.npy")
        np.save(path, acc)
        print(path, acc.shape, float(acc.mean()))
