"""The oracle's integrated path against numbers the reference itself produced.

tests/golden/survey_kat.json holds exact integer call counts of the reference raytracer
(closestIntersection / existsIntersection / bounce* entries, glm::determinant and
glm::inverse entries) for whole frames rendered from its default-seeded global RNG. Every
one of them depends on each gate, tie, Russian-roulette, Fresnel and draw-order decision
of ~10^6 rays, so reproducing them exactly pins the composition of the path, not just its
primitives.
"""
import json
import os

import numpy as np
import pytest

KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "survey_kat.json")))
SCENES = {"cornell": (0, False), "glasscube": (2, True)}


def test_centre_pixel_hit(O):
    k = KAT["centre_pixel"]
    tris = O.load_test_model(0)
    o = np.array([k["origin"]], dtype=np.float32)
    d = np.array([k["direction"]], dtype=np.float32)
    idx, t, _ = O.closest_hit(tris, None, o, d)
    assert idx[0] >= 0
    assert float(t[0]) == float.fromhex(k["distance_hex"])
    assert np.allclose(tris[idx[0]]["normal"], k["normal"], atol=5e-7)


@pytest.mark.parametrize("case", KAT["call_counts"], ids=lambda c: f"{c['scene']}{c['size']}")
def test_whole_frame_call_counts(O, pkg, case):
    which, glass = SCENES[case["scene"]]
    tris = O.load_test_model(which)
    W = case["size"]
    cam = O.default_camera(W)
    light = O.default_light(glass)
    p = O.default_params(W, W, rng_mode=pkg.abi.RNG_MINSTD_GLOBAL)
    _, st = O.render_accumulate(tris, None, cam, light, p, 0, case["spp"])
    for key in ("closest_rays", "shadow_rays", "diffuse_bounces", "mirror_bounces", "glass_bounces"):
        if key in case:
            assert getattr(st, key) == case[key], key
    if "rays" in case:
        assert st.closest_rays + st.shadow_rays == case["rays"]
        assert 2 * st.tri_tests == case["determinant_calls"]
        assert st.gate_passes == case["inverse_calls"]
