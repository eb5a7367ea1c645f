"""The reference's own rendering against this implementation, at image level.

tests/golden/glasswolf_screen_500x500.npz holds the pixels of CW/raytracer/screenshot.bmp: the
ARGB screen buffer the reference saved (main.cpp:105) after rendering its glassWolf scene
(main.cpp:78-89; 500 x 500, definitions.h:43-44) for about 10^4 samples per pixel; made by
oracle/ref_harness/screenshot_fixture.py. It is the one output of the whole path that the
reference's files hold, and it pins what call counts cannot: the radiance values (light power,
inverse-square fall-off, bounce weights, Russian roulette, the tonemap constant).

What can be expected. The reference was built with g++ -Ofast and rendered from one racing global
random engine, so nothing is comparable bit for bit. In the ROOM (diffuse walls, floor, ceiling,
the wolf's shadow) paths are robust against the last bit and the two images agree as two Monte-Carlo
renderings of one estimator do: the means to a fraction of a per cent. In the glass WOLF a path
restarts exactly on the surface it left (pathtracer.cpp:100, no offset), whether the next scan
re-hits that surface at t ~ 0 hangs on the last bit, and every such re-hit multiplies by 1.15:
strict IEEE arithmetic (the parity contract) renders the glass brighter than -Ofast arithmetic does.
test_ofast_oracle_matches_reference_glass pins that explanation: the C oracle rebuilt with the
reference's own flags (-Ofast, CW/raytracer/Makefile:12-14) reproduces the screenshot's glass to
within Monte-Carlo noise (0.995 .. 0.998 at 768 000 paths), and on identical random streams the
-Ofast build is 0.85 .. 0.86 times as bright as the strict build there (only 72 % of the paths
through the glass keep their bits between the two builds).
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
S = 500


def load():
    ref = np.load(os.path.join(HERE, "golden", "glasswolf_screen_500x500.npz"))["rgb"].astype(np.float64)
    assert ref.shape == (S, S, 3)
    room = np.ones((S, S), bool)
    room[250:475, 175:425] = False  # the wolf's bounding box on the screen
    wolf = np.zeros((S, S), bool)
    wolf[275:450, 200:400] = True
    return ref, room, wolf


def scene(pkg):
    return np.fromfile(os.path.join(HERE, "golden", "obj", "glasswolf_scene.bin"), dtype=pkg.abi.TRIANGLE_DTYPE)


def test_oracle_radiance_matches_reference_screenshot(O, pkg):
    # CPU: 500 pixels of the room x 48 paths through the strict oracle against the screenshot's
    # values there (un-tonemapped: (value + 0.5) / 255 is the pixel's radiance * 2 * 3.1415)
    ref, room, _ = load()
    tris = scene(pkg)
    cam, light = O.default_camera(S), O.default_light(True)
    p = O.default_params(S, S, rng_mode=1, seed=2)
    ok = room & (ref[..., 0] >= 20) & (ref[..., 0] < 150)  # neither quantised to nothing nor clipped
    ys, xs = np.nonzero(ok)
    pick = np.random.default_rng(2).choice(len(xs), 500, replace=False)
    spp = 48
    px = np.repeat(xs[pick], spp).astype(np.int32)
    py = np.repeat(ys[pick], spp).astype(np.int32)
    sm = np.tile(np.arange(spp), len(pick)).astype(np.uint32)
    rgb, _ = O.trace_paths(tris, None, cam, light, p, px, py, sm)
    ours = 255.0 * rgb.reshape(len(pick), spp, 3).mean(1)[:, 0] / (1.0 / (2 * 3.1415))
    theirs = ref[ys[pick], xs[pick], 0] + 0.5
    ratio = ours.mean() / theirs.mean()
    print(f"oracle / reference screenshot, 500 room pixels x {spp} paths: {ratio:.4f}")
    assert 0.92 < ratio < 1.08  # 24 000 heavy-tailed samples: a few per cent of noise


def _wolf_paths(O, pkg, n_pixels, spp, seed):
    ref, _, wolf = load()
    ok = wolf & (ref[..., 0] < 200)  # unclipped pixels of the glass
    ys, xs = np.nonzero(ok)
    pick = np.random.default_rng(seed).choice(len(xs), n_pixels, replace=False)
    px = np.repeat(xs[pick], spp).astype(np.int32)
    py = np.repeat(ys[pick], spp).astype(np.int32)
    sm = np.tile(np.arange(spp), n_pixels).astype(np.uint32)
    return ref[ys[pick], xs[pick], 0] + 0.5, px, py, sm


def test_ofast_oracle_matches_reference_glass(O, pkg):
    # Mirror / glass radiance against the reference's own image. The same 384 000 paths through the
    # glass wolf (2000 unclipped pixels x 192 samples, identical random streams) with the strict
    # oracle and with the oracle built like the reference (-Ofast):
    #   * the -Ofast build agrees with the reference's screenshot within Monte-Carlo noise;
    #   * the strict build -- the parity contract -- is brighter by the factor the two builds show
    #     against each other, which is the whole difference between the GPU image and the screenshot.
    import ctypes as C
    tris = scene(pkg)
    cam, light = O.default_camera(S), O.default_light(True)
    p = O.default_params(S, S, rng_mode=1, seed=1)
    n_pixels, spp = 2000, 192
    theirs, px, py, sm = _wolf_paths(O, pkg, n_pixels, spp, seed=1)

    def trace(fast):
        rgb = np.zeros((len(px), 3), dtype=np.float32)
        rc = O.oracle(fast).or_trace_paths(O.ptr(tris), len(tris), None, 0, C.byref(cam), C.byref(light), C.byref(p),
                                           O.ptr(px), O.ptr(py), O.ptr(sm), len(px), O.ptr(rgb), None)
        assert rc == 0
        return rgb

    strict, ofast = trace(False), trace(True)
    display = lambda rgb: 255.0 * rgb.reshape(n_pixels, spp, 3).mean(1)[:, 0] * (2 * 3.1415)
    fast_vs_ref = display(ofast).mean() / theirs.mean()
    strict_vs_ref = display(strict).mean() / theirs.mean()
    fast_vs_strict = float(ofast[:, 0].sum() / strict[:, 0].sum())
    same = (strict.view(np.uint32) == ofast.view(np.uint32)).all(axis=1).mean()
    print(f"glass wolf, {len(px)} paths: -Ofast oracle / screenshot = {fast_vs_ref:.4f}, strict oracle / screenshot = "
          f"{strict_vs_ref:.4f}, -Ofast / strict = {fast_vs_strict:.4f}, paths with identical bits {same:.3f}")
    assert 0.96 < fast_vs_ref < 1.04
    assert 0.80 < fast_vs_strict < 0.91
    assert 1.10 < strict_vs_ref < 1.24
    assert 0.6 < same < 0.85


@pytest.mark.gpu
def test_gpu_render_matches_reference_screenshot(O, pkg):
    import torch
    assert torch.cuda.is_available()
    pkg.abi.load_library()
    api, abi = pkg.api, pkg.abi
    ref, room, wolf = load()
    tris = scene(pkg)
    cam, light = O.default_camera(S), O.default_light(True)
    spp = 4096  # 10^9 paths
    sc = api.Scene(tris)
    assert sc.bvh_info()[0]
    acc = sc.render_accumulate(cam, light, O.default_params(S, S, rng_mode=abi.RNG_COUNTER, seed=7), 0, spp)
    img = sc.resolve_argb(acc, spp)
    sc.close()
    rgb = np.stack([(img >> 16) & 255, (img >> 8) & 255, img & 255], -1).astype(np.float64)
    unclipped = (ref[..., 0] < 200) & (rgb[..., 0] < 200)
    m = room & unclipped
    for ch in (0, 1):
        ratio = rgb[..., ch][m].mean() / ref[..., ch][m].mean()
        print(f"room, channel {ch}: ours / reference = {ratio:.4f} over {m.sum()} pixels")
        assert 0.99 < ratio < 1.01
    # 10 x 10 blocks of the room: Monte-Carlo noise of two renderings, no structure
    d = ((rgb - ref) * m[..., None]).reshape(S // 10, 10, S // 10, 10, 3).mean((1, 3))[..., 0]
    assert np.sqrt((d ** 2).mean()) < 1.5
    assert np.corrcoef(rgb[..., 0].ravel(), ref[..., 0].ravel())[0, 1] > 0.97
    # the glass: brighter under strict IEEE arithmetic than under the reference's -Ofast build
    w = rgb[..., 0][wolf].mean() / ref[..., 0][wolf].mean()
    sel = wolf & (ref[..., 0] < 200)  # the pixels the CPU test draws from
    wu = rgb[..., 0][sel].mean() / ref[..., 0][sel].mean()
    print(f"wolf: ours / reference = {w:.4f} (display values, clipped pixels included), {wu:.4f} over the pixels the reference left unclipped")
    # the factor between strict and -Ofast arithmetic that test_ofast_oracle_matches_reference_glass
    # measures on the CPU (strict oracle / screenshot = 1.15 on those pixels; the display's clamp at
    # 255 hides a part of it here)
    assert 1.04 < w < 1.12
    assert 1.11 < wu < 1.20
