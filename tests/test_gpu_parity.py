"""GPU parity: the HIP path through the C ABI against the CPU oracle on identical inputs.

Bars: bit-exact for hit index, distance, position, occlusion and the ARGB image (integer /
bit-pattern work); per-path radiance compared bit for bit too, with the fraction of paths
allowed to differ stated per test (the only device/host difference is powf in the Fresnel
comparison of glass vertices; cosf/sinf reproduce glibc's results).
"""
import os

import numpy as np
import pytest

import raygen

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def api(pkg):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg.abi.load_library()
    return pkg.api


SCENES = {"cornell": (0, False), "glasscube": (2, True), "none": (3, True)}


def check_closest(api, O, tris, spheres, o, d):
    sc = api.Scene(tris, spheres)
    idx, t, pos = sc.closest_hit(o, d)
    sc.close()
    oidx, ot, opos = O.closest_hit(tris, spheres, o, d)
    assert np.array_equal(idx, oidx), f"{np.count_nonzero(idx != oidx)} of {len(idx)} hit indices differ"
    hit = oidx >= 0
    assert np.array_equal(bits(t[hit]), bits(ot[hit]))
    assert np.array_equal(bits(pos[hit]), bits(opos[hit]))
    return idx


@pytest.mark.parametrize("scene", ["cornell", "glasscube", "none"])
def test_closest_hit_bit_exact(api, O, scene):
    tris = O.load_test_model(SCENES[scene][0])
    n = 1 << 16
    for name, (o, d) in {
        "primary": raygen.primary_rays(n),
        "interior": raygen.interior_rays(n),
        "surface": raygen.surface_rays(tris, n),
        "degenerate": raygen.degenerate_rays(tris, 4096),
    }.items():
        idx = check_closest(api, O, tris, None, o, d)
        if name in ("primary", "interior"):
            assert (idx >= 0).mean() > 0.3, name


@pytest.mark.parametrize("scene", ["cornell", "glasscube"])
def test_fast_scan_equals_literal_scan_at_scale(api, O, scene):
    # full-size property check on the device: the shipped scan (estimate-based gate, fast exact
    # reciprocal) against the literal reference arithmetic on 2^30 generated rays, and the fast
    # reciprocal against IEEE division for every float
    tris = O.load_test_model(SCENES[scene][0])
    sc = api.Scene(tris)
    rcp_bad, rcp_err_ulp, closest_bad, any_bad = sc.selfcheck(1 << 30, seed=123)
    assert sc.last_selfcheck_sincos == 0
    sc.close()
    assert rcp_bad == 0
    assert rcp_err_ulp <= 1.0, rcp_err_ulp   # the gate's band of 2^-20 assumes an estimate good to ~1 ulp
    assert closest_bad == 0 and any_bad == 0, (closest_bad, any_bad)


def test_selfcheck_beyond_one_launch(api, O):
    # more rays than one launch holds threads: split into launches of 2^30, a seed each
    sc = api.Scene(O.load_test_model(2))
    _, _, closest_bad, any_bad = sc.selfcheck((1 << 32) + 12345, seed=5)
    sc.close()
    assert (closest_bad, any_bad) == (0, 0)


def test_closest_hit_with_sphere(api, O):
    tris = O.load_test_model(2)
    sph = api.make_sphere(0.3, [-0.45, 0.7, -0.55, 1], [1, 0, 0])  # main.cpp:75
    n = 1 << 16
    for o, d in (raygen.primary_rays(n), raygen.interior_rays(n)):
        idx = check_closest(api, O, tris, sph, o, d)
        assert (idx == len(tris)).sum() > 100  # the sphere is actually hit


def test_closest_hit_tie_rule_on_shared_diagonal(api, O):
    # rays through the shared diagonal of the two floor triangles: both pass, the later index wins
    tris = O.load_test_model(0)
    s = np.linspace(-0.9, 0.9, 2001, dtype=np.float32)
    o = np.stack([s, np.full_like(s, -0.5), -s, np.ones_like(s)], axis=1)  # above the floor diagonal x = -z
    d = np.tile(np.array([0, 1, 0, 0], dtype=np.float32), (len(s), 1))
    check_closest(api, O, tris, None, o, d)


@pytest.mark.parametrize("scene", ["cornell", "glasscube"])
def test_any_hit_bit_exact(api, O, scene):
    tris = O.load_test_model(SCENES[scene][0])
    o, d, r = raygen.shadow_rays(tris, 1 << 16)
    sc = api.Scene(tris)
    occ = sc.any_hit(o, d, r)
    sc.close()
    oocc = O.any_hit(tris, o, d, r)
    assert np.array_equal(occ, oocc)
    assert 0.02 < occ.mean() < 0.98


def test_empty_and_ragged_inputs(api, O):
    tris = O.load_test_model(3)
    sc = api.Scene(tris)
    z4 = np.zeros((0, 4), dtype=np.float32)
    idx, t, pos = sc.closest_hit(z4, z4)
    assert idx.shape == (0,)
    assert sc.any_hit(z4, z4, np.zeros(0, dtype=np.float32)).shape == (0,)
    o, d = raygen.interior_rays(257)  # not a multiple of the block size
    check_closest(api, O, tris, None, o, d)
    sc.close()
    empty = api.Scene(np.zeros(0, dtype=O._pkg().abi.TRIANGLE_DTYPE))
    idx, _, _ = empty.closest_hit(o, d)
    assert np.all(idx == -1)
    assert not empty.any_hit(o, d, np.ones(len(o), dtype=np.float32)).any()
    empty.close()


def path_list(W, H, n, spp, seed=7):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, W, n).astype(np.int32), rng.integers(0, H, n).astype(np.int32),
            rng.integers(0, spp, n).astype(np.uint32))


@pytest.mark.parametrize("scene,rng_mode,max_differ", [
    ("cornell", 0, 0.0), ("cornell", 1, 0.0), ("glasscube", 0, 0.0), ("glasscube", 1, 0.0)])
def test_path_radiance_matched_streams(api, O, pkg, scene, rng_mode, max_differ):
    which, glass = SCENES[scene]
    tris = O.load_test_model(which)
    W = H = 256
    cam, light = O.default_camera(W), O.default_light(glass)
    p = O.default_params(W, H, rng_mode=rng_mode, seed=11)
    px, py, smp = path_list(W, H, 200_000, 4096)
    sc = api.Scene(tris)
    rgb, cnt = sc.trace_paths(cam, light, p, px, py, smp)
    sc.close()
    orgb, ocnt = O.trace_paths(tris, None, cam, light, p, px, py, smp)
    same = np.all(bits(rgb) == bits(orgb), axis=1)
    lit = np.any(orgb != 0, axis=1)
    assert lit.sum() > 5000
    differ = 1.0 - same.mean()
    assert differ <= max_differ, f"{(~same).sum()} of {len(same)} paths differ"
    assert (np.any(cnt != ocnt, axis=1)).mean() <= max_differ


def test_path_radiance_with_sphere(api, O):
    tris = O.load_test_model(2)
    sph = api.make_sphere(0.3, [-0.45, 0.7, -0.55, 1], [1, 0, 0])
    W = H = 256
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, seed=5)
    px, py, smp = path_list(W, H, 100_000, 1024)
    sc = api.Scene(tris, sph)
    rgb, cnt = sc.trace_paths(cam, light, p, px, py, smp)
    sc.close()
    orgb, ocnt = O.trace_paths(tris, sph, cam, light, p, px, py, smp)
    same = np.all(bits(rgb) == bits(orgb), axis=1)
    assert same.all(), f"{(~same).sum()} paths differ"


@pytest.mark.parametrize("scene,W,H", [("cornell", 64, 64), ("glasscube", 72, 50)])
def test_render_accumulate_matches_oracle(api, O, pkg, scene, W, H):
    which, glass = SCENES[scene]
    tris = O.load_test_model(which)
    cam, light = O.default_camera(W), O.default_light(glass)
    p = O.default_params(W, H, seed=3, flags=pkg.abi.FLAG_STATS)
    spp = 48
    sc = api.Scene(tris)
    acc = sc.render_accumulate(cam, light, p, 0, spp)
    st = sc.stats()
    acc2 = sc.render_accumulate(cam, light, p, 0, spp)
    sc.close()
    assert np.array_equal(bits(acc), bits(acc2)), "render is not deterministic"
    _, ost = O.render_accumulate(tris, None, cam, light, p, 0, spp, threads=8)
    # a pixel's samples are added 16 at a time (one work item), the item sums in sample order: the
    # oracle summed the same way gives the same bits
    oacc = None
    for b in range(0, spp, 16):
        img, _ = O.render_accumulate(tris, None, cam, light, p, b, 16, threads=8)
        oacc = img.copy() if oacc is None else (oacc + img).astype(np.float32)
    assert np.array_equal(bits(acc), bits(oacc)), f"{np.count_nonzero(bits(acc) != bits(oacc))} of {acc.size} channel sums differ"
    assert st.paths == ost.paths == W * H * spp
    for k in ("closest_rays", "shadow_rays", "tri_tests", "gate_passes", "diffuse_bounces",
              "mirror_bounces", "glass_bounces"):
        assert getattr(st, k) == getattr(ost, k), (k, getattr(st, k), getattr(ost, k))


def test_config0_whole_frame_is_the_oracle_frame(api, O, pkg):
    # BASELINE configs[0] in full: the Cornell test model at 256 x 256, 16 spp, diffuse only, no fog,
    # the reference's random engine seeded per path. 16 samples are one work item: one lane adds a
    # pixel's radiances in sample order exactly like main.cpp:137 does, so the frames agree bit for
    # bit -- accumulation buffer, ARGB screen and every counter
    abi = pkg.abi
    W = H = 256
    tris = O.load_test_model(0)
    cam, light = O.default_camera(W), O.default_light(False)
    for flags in (0, abi.FLAG_STATS):   # the shipped kernel and the counting one
        p = O.default_params(W, H, seed=0, rng_mode=abi.RNG_MINSTD_PATH, flags=flags)
        sc = api.Scene(tris)
        acc = sc.render_accumulate(cam, light, p, 0, 16)
        st = sc.stats()
        argb = sc.resolve_argb(acc, 16)
        sc.close()
        oacc, ost = O.render_accumulate(tris, None, cam, light, p, 0, 16)
        assert oacc.sum() > 0
        assert np.array_equal(bits(acc), bits(oacc)), np.count_nonzero(bits(acc) != bits(oacc))
        assert np.array_equal(argb, O.resolve_argb(oacc, 16))
        if flags:
            for k in ("paths", "closest_rays", "shadow_rays", "tri_tests", "gate_passes", "diffuse_bounces"):
                assert getattr(st, k) == getattr(ost, k), k
            assert st.mirror_bounces == st.glass_bounces == 0


@pytest.mark.parametrize("rng", ["counter", "minstd"])
def test_bench_workload_image_is_the_oracle_image(api, O, pkg, rng):
    # BASELINE configs[1] at reduced size: the GlassCube scene (glass block, mirror) with fog, as
    # bench.py renders it. The oracle renders 16 samples at a time and the images are added in
    # order -- the GPU's association (one work item = 16 samples, item sums added in sample order) --
    # and then every pixel agrees bit for bit, shipped kernel and counting kernel
    import bench
    abi = pkg.abi
    W = H = 160
    spp = 80
    tris = O.load_test_model(2)
    cam, light = O.default_camera(W), O.default_light(True)
    mode = abi.RNG_COUNTER if rng == "counter" else abi.RNG_MINSTD_PATH
    want = None
    p = O.default_params(W, H, seed=0, rng_mode=mode, fog_sigma_t=bench.FOG_SIGMA_T, fog_albedo=bench.FOG_ALBEDO)
    for c in range(spp // 16):
        img, _ = O.render_accumulate(tris, None, cam, light, p, 16 * c, 16, threads=8)
        want = img.copy() if want is None else (want + img).astype(np.float32)
    assert want.sum() > 0
    for flags in (0, abi.FLAG_STATS):
        p.flags = flags
        sc = api.Scene(tris)
        acc = sc.render_accumulate(cam, light, p, 0, spp)
        sc.close()
        assert np.array_equal(bits(acc), bits(want)), (flags, np.count_nonzero(bits(acc) != bits(want)))


def test_bench_frame_at_full_resolution_is_the_oracle_frame(api, O, pkg):
    # BASELINE configs[1] at its 1024 x 1024 with a quarter of its samples (2.7 * 10^8 paths; the CPU side
    # takes ~20 s on 16 threads). All 1024 samples were compared once with tools/full_frame_parity.py:
    # 0 of 3 145 728 channel sums differ (profiles/r2_full_frame_parity.json).
    import bench
    abi = pkg.abi
    S, spp = bench.SIZE, 256
    threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "0") or 64))
    tris = O.load_test_model(2)
    cam, light = O.default_camera(S), O.default_light(True)
    p = O.default_params(S, S, seed=0, rng_mode=abi.RNG_COUNTER, fog_sigma_t=bench.FOG_SIGMA_T, fog_albedo=bench.FOG_ALBEDO)
    sc = api.Scene(tris)
    acc = sc.render_accumulate(cam, light, p, 0, spp)
    sc.close()
    want = None
    for b in range(0, spp, 16):
        img, _ = O.render_accumulate(tris, None, cam, light, p, b, 16, threads=threads)
        want = img.copy() if want is None else (want + img).astype(np.float32)
    assert want.sum() > 0 and np.array_equal(bits(acc), bits(want)), np.count_nonzero(bits(acc) != bits(want))


def test_render_items_and_fold_are_the_two_halves(api, O, pkg):
    # mipt_render_items_dev + mipt_fold_items_dev == mipt_render_accumulate_dev, bit for bit; the fold may
    # run on another stream and into an image that holds earlier samples; the items of one render at a time
    import torch
    abi = pkg.abi
    W, H = 96, 80
    tris = O.load_test_model(2)
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, seed=8, rng_mode=abi.RNG_COUNTER, fog_sigma_t=0.1)
    sc = api.Scene(tris)
    whole = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    sc.render_accumulate_dev(cam, light, p, 0, 80, whole)
    halves = torch.zeros_like(whole)
    sc.render_items_dev(cam, light, p, 0, 48)
    with pytest.raises(api.MiptError):            # unfolded items: no second render on this scene
        sc.render_items_dev(cam, light, p, 48, 32)
    with pytest.raises(api.MiptError):
        sc.render_accumulate_dev(cam, light, p, 48, 32, halves)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                 # the fold waits for the render through the scene's event
        sc.fold_items_dev(halves)
        sc.fold_items_dev(halves)                 # nothing pending: leaves the image alone
    side.synchronize()
    sc.render_items_dev(cam, light, p, 48, 32)
    sc.fold_items_dev(halves)
    torch.cuda.synchronize()
    sc.close()
    assert whole.abs().sum().item() > 0
    assert torch.equal(whole.view(torch.int32), halves.view(torch.int32))


@pytest.mark.parametrize("W,H,spp", [(1, 1, 40), (1, 9, 17), (3, 2500, 3), (4097, 1, 2), (9, 8, 16), (8, 8, 1)])
def test_odd_image_shapes_are_the_oracle_images(api, O, pkg, W, H, spp):
    # one pixel, single rows and columns, sizes around the 8 x 8 tiles the work queue hands out
    tris = O.load_test_model(2)
    cam, light = O.default_camera(max(W, H)), O.default_light(True)
    p = O.default_params(W, H, seed=13, rng_mode=pkg.abi.RNG_COUNTER, fog_sigma_t=0.1)
    sc = api.Scene(tris)
    acc = sc.render_accumulate(cam, light, p, 5, spp)
    screen = sc.resolve_argb(acc, spp)
    sc.close()
    want = None
    for b in range(0, spp, 16):
        img, _ = O.render_accumulate(tris, None, cam, light, p, 5 + b, min(16, spp - b))
        want = img.copy() if want is None else (want + img).astype(np.float32)
    assert acc.shape == (W, H, 3) and np.array_equal(bits(acc), bits(want))
    assert np.array_equal(screen, O.resolve_argb(want, spp))


def moved_camera_and_light(O, width, yaw, pitch, steps_forward, light_shift, light_colour):
    """What handleKeyboard (keyboard.cpp:5-76) leaves after some key presses: rotateY / rotateX
    (camera.cpp:15-35) set the two rotation matrices, W moves the camera along Rx * Ry * (0, 0, 0.25, 0),
    the arrow keys move the light."""
    cam, light = O.default_camera(width), O.default_light(True)
    cy, sy = np.float32(np.cos(np.float32(yaw))), np.float32(np.sin(np.float32(yaw)))
    cx, sx = np.float32(np.cos(np.float32(pitch))), np.float32(np.sin(np.float32(pitch)))
    ry = np.array([[cy, 0, -sy, 0], [0, 1, 0, 0], [sy, 0, cy, 0], [0, 0, 0, 1]], dtype=np.float32)   # rows = GLM columns
    rx = np.array([[1, 0, 0, 0], [0, cx, sx, 0], [0, -sx, cx, 0], [0, 0, 0, 1]], dtype=np.float32)
    cam.rot_y[:] = ry.ravel().tolist()
    cam.rot_x[:] = rx.ravel().tolist()
    step = (rx.T @ (ry.T @ np.array([0, 0, 0.25, 0], dtype=np.float32))).astype(np.float32)
    pos = np.array(list(cam.pos), dtype=np.float32)
    for _ in range(steps_forward):
        pos = (pos + step).astype(np.float32)
    cam.pos[:] = pos.tolist()
    light.pos[:] = (np.array(list(light.pos), dtype=np.float32) + np.array(light_shift + [0], dtype=np.float32)).tolist()
    light.color[:] = light_colour
    return cam, light


@pytest.mark.parametrize("scene", ["cornell", "glasscube"])
def test_moved_camera_and_coloured_light(api, O, pkg, scene):
    # every test above looks through reset_camera's camera at reset_light's white light; the
    # interactive loop turns and moves the camera and the light before it calls Draw. Camera yawed,
    # pitched and walked forward, light shifted and coloured (three channels, three divisions in
    # directLight): image, ARGB screen, counters and listed paths against the oracle, bit for bit
    abi = pkg.abi
    which, glass = SCENES[scene]
    tris = O.load_test_model(which)
    W, H, spp = 120, 96, 48
    cam, light = moved_camera_and_light(O, W, yaw=0.12, pitch=-0.08, steps_forward=2, light_shift=[-0.06, 0.04, 0.1],
                                        light_colour=[12.0, 15.5, 9.25])
    p = O.default_params(W, H, seed=6, rng_mode=abi.RNG_MINSTD_PATH, fog_sigma_t=0.05, flags=abi.FLAG_STATS)
    sc = api.Scene(tris)
    acc = sc.render_accumulate(cam, light, p, 0, spp)
    st = sc.stats()
    screen = sc.resolve_argb(acc, spp)
    px, py, smp = path_list(W, H, 40_000, spp)
    rgb, cnt = sc.trace_paths(cam, light, p, px, py, smp)
    sc.close()
    _, ost = O.render_accumulate(tris, None, cam, light, p, 0, spp, threads=8)
    want = None
    for b in range(0, spp, 16):
        img, _ = O.render_accumulate(tris, None, cam, light, p, b, 16, threads=8)
        want = img.copy() if want is None else (want + img).astype(np.float32)
    # the three channels of the light differ, so do the image's
    assert want.sum() > 0 and not np.allclose(want[..., 0], want[..., 1])
    assert np.array_equal(bits(acc), bits(want)) and np.array_equal(screen, O.resolve_argb(want, spp))
    for k in ("paths", "closest_rays", "shadow_rays", "tri_tests", "gate_passes", "diffuse_bounces", "mirror_bounces",
              "glass_bounces", "fog_events"):
        assert getattr(st, k) == getattr(ost, k), k
    orgb, ocnt = O.trace_paths(tris, None, cam, light, p, px, py, smp)
    assert np.array_equal(bits(rgb), bits(orgb)) and np.array_equal(cnt, ocnt)


@pytest.mark.parametrize("kw", [
    dict(max_depth=3, rr_prob=0.35, diffuse_weight=0.8, glass_weight=1.05, ior_glass=1.33, ior_air=1.0003),
    dict(max_depth=1, rr_prob=0.0, diffuse_weight=1.3, glass_weight=0.9, ior_glass=2.4, ior_air=1.0, fog_sigma_t=0.4, fog_albedo=0.5),
    dict(max_depth=16, rr_prob=0.9, seed=0xFFFFFFFF, ior_glass=1.0, ior_air=1.5)])
def test_parameters_off_their_defaults(api, O, pkg, kw):
    # the #defines of definitions.h:43-55 as run-time parameters, away from the reference's values:
    # depth limit, roulette, bounce weights, refraction indices (also air denser than glass), fog
    abi = pkg.abi
    tris = O.load_test_model(2)
    sph = api.make_sphere(0.3, [-0.45, 0.7, -0.55, 1], [1, 0, 0])
    W, H, spp = 80, 64, 32
    cam, light = O.default_camera(W), O.default_light(True)
    for rng_mode in (abi.RNG_COUNTER, abi.RNG_MINSTD_PATH):
        p = O.default_params(W, H, rng_mode=rng_mode, flags=abi.FLAG_STATS, **kw)
        sc = api.Scene(tris, sph)
        acc = sc.render_accumulate(cam, light, p, 0, spp)
        st = sc.stats()
        sc.close()
        _, ost = O.render_accumulate(tris, sph, cam, light, p, 0, spp, threads=8)
        want = None
        for b in range(0, spp, 16):
            img, _ = O.render_accumulate(tris, sph, cam, light, p, b, 16, threads=8)
            want = img.copy() if want is None else (want + img).astype(np.float32)
        assert np.array_equal(bits(acc), bits(want)), (kw, rng_mode)
        for k in ("paths", "closest_rays", "shadow_rays", "sphere_tests", "diffuse_bounces", "mirror_bounces", "glass_bounces",
                  "fog_events"):
            assert getattr(st, k) == getattr(ost, k), (k, kw)


def test_render_sample_ranges_compose(api, O):
    # samples [0,64) in one call == [0,32) then [32,64) accumulated, bit for bit: the stream of a path
    # depends only on (pixel, sample index), and the sums of the work items (16 samples each) join the
    # running image one by one in sample order, whichever call brings them. A split inside a work
    # item changes the association of that item's samples: equal to rounding
    tris = O.load_test_model(2)
    W = H = 64
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, rng_mode=1, seed=9, fog_sigma_t=0.2)
    sc = api.Scene(tris)
    whole = sc.render_accumulate(cam, light, p, 0, 64)
    assert whole.sum() > 0
    for cuts in ((32,), (16, 32, 48), (48,)):
        parts, begin = None, 0
        for end in cuts + (64,):
            parts = sc.render_accumulate(cam, light, p, begin, end - begin, accum=parts)
            begin = end
        assert np.array_equal(bits(whole), bits(parts)), cuts
    ragged = sc.render_accumulate(cam, light, p, 0, 20)
    ragged = sc.render_accumulate(cam, light, p, 20, 44, accum=ragged)
    sc.close()
    assert np.allclose(whole, ragged, rtol=2e-6, atol=1e-7) and not np.array_equal(bits(whole), bits(ragged))


def test_resolve_argb_bit_exact(api, O):
    tris = O.load_test_model(0)
    W, H = 80, 48
    cam, light = O.default_camera(W), O.default_light(False)
    p = O.default_params(W, H)
    sc = api.Scene(tris)
    acc = sc.render_accumulate(cam, light, p, 0, 16)
    acc[3, 5] = [np.nan, -1.0, 1e9]  # clamp / NaN handling
    img = sc.resolve_argb(acc, 16)
    sc.close()
    oimg = O.resolve_argb(acc, 16)
    assert img.shape == (H, W)
    assert np.array_equal(img, oimg)
    assert np.all(img >> 24 == 128)


def test_error_codes(api, O, pkg):
    abi = pkg.abi
    tris = O.load_test_model(0)
    sc = api.Scene(tris)
    W = H = 32
    cam, light = O.default_camera(W), O.default_light(False)
    with pytest.raises(api.MiptError) as e:
        sc.render_accumulate(cam, light, O.default_params(W, H, rng_mode=abi.RNG_MINSTD_GLOBAL), 0, 1)
    assert e.value.code == abi.MIPT_E_UNSUPPORTED
    with pytest.raises(api.MiptError) as e:
        sc.render_accumulate(cam, light, O.default_params(W, H, max_depth=99), 0, 1)
    assert e.value.code == abi.MIPT_E_INVALID
    bad = tris.copy()
    bad["material"][0] = 17
    with pytest.raises(api.MiptError):
        api.Scene(bad)
    sc.close()


def test_fog_matches_oracle(api, O, pkg):
    # fog has no reference code; the check is GPU vs this repository's CPU restatement. The fog's
    # logarithm and exponential are fixed sequences of binary32 operations shared by host and device,
    # so a dense medium changes nothing about parity: image (the oracle summed 16 samples at a time,
    # like the GPU's work items), event counter and every listed path, bit for bit
    tris = O.load_test_model(2)
    W = H = 64
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, seed=21, fog_sigma_t=0.35, fog_albedo=0.8, flags=pkg.abi.FLAG_STATS)
    sc = api.Scene(tris)
    acc = sc.render_accumulate(cam, light, p, 0, 64)
    st = sc.stats()
    sc.close()
    _, ost = O.render_accumulate(tris, None, cam, light, p, 0, 64, threads=8)
    assert st.fog_events == ost.fog_events > 0
    oacc = None
    for b in range(0, 64, 16):
        img, _ = O.render_accumulate(tris, None, cam, light, p, b, 16, threads=8)
        oacc = img.copy() if oacc is None else (oacc + img).astype(np.float32)
    assert np.array_equal(bits(acc), bits(oacc))
    px, py, smp = path_list(W, H, 50_000, 64)
    p2 = O.default_params(W, H, seed=21, fog_sigma_t=0.35, fog_albedo=0.8)
    sc = api.Scene(tris)
    rgb, _ = sc.trace_paths(cam, light, p2, px, py, smp)
    sc.close()
    orgb, _ = O.trace_paths(tris, None, cam, light, p2, px, py, smp)
    assert np.array_equal(bits(rgb), bits(orgb))


def quad_scene(O, dtype, n_quads, seed, axis_planes=False):
    """Random planar quads split into two triangles with every vertex labelling, next to each other
    in index order: parallelograms, general convex quads, thin ones, at scales 1e-3 .. 1; plus
    near-pairs that must NOT merge (bent quads, triangles sharing only a vertex). With axis_planes
    every quad lies in a plane normal to a coordinate axis (one coordinate exactly constant), half
    of them with edges along the other two axes, half turned within the plane."""
    import bigscenes
    rng = np.random.default_rng(seed)
    v0s, v1s, v2s = [], [], []
    for k in range(n_quads):
        c = rng.uniform(-0.8, 0.8, 3)
        size = 10 ** rng.uniform(-3, 0)
        a = rng.normal(size=3); a /= np.linalg.norm(a)
        b = rng.normal(size=3); b -= a * (a @ b); b /= np.linalg.norm(b)
        if axis_planes:
            ax = k % 3
            i, j = (ax + 1) % 3, (ax + 2) % 3
            phi = 0.0 if (k // 3) % 2 == 0 else rng.uniform(0, 2 * np.pi)
            a = np.zeros(3); b = np.zeros(3)
            a[i], a[j] = np.cos(phi), np.sin(phi)
            b[i], b[j] = -np.sin(phi), np.cos(phi)
            if phi == 0.0:
                a[j] = 0.0; b[i] = 0.0
            if k % 5 == 0:
                a = -a   # both orientations of the frame
        aspect = 10 ** rng.uniform(-1.5, 0)
        P = [c, c + a * size, c + a * size + b * size * aspect, c + b * size * aspect]   # parallelogram
        kind = k % 4
        if kind == 1:    # general convex quad: move one corner in the plane
            P[2] = P[2] + (a * rng.uniform(-0.3, 0.6) + b * aspect * rng.uniform(-0.3, 0.6)) * size
        elif kind == 2:  # bent: one corner out of the plane (must not merge, or merge harmlessly)
            P[2] = P[2] + np.cross(a, b) * size * 10 ** rng.uniform(-7, -1)
        P = [np.float32(p) for p in P]
        t1 = [P[0], P[1], P[2]]
        t2 = [P[0], P[2], P[3]] if kind != 3 else [P[0] + np.float32(1e-3), P[2], P[3]]  # kind 3: only one shared vertex
        r1, r2 = rng.permutation(3), rng.permutation(3)
        for t, r in ((t1, r1), (t2, r2)):
            v0s.append(t[r[0]]); v1s.append(t[r[1]]); v2s.append(t[r[2]])
    n = len(v0s)
    color = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    return bigscenes._finish(dtype, np.array(v0s), np.array(v1s), np.array(v2s), color, np.zeros(n, dtype=np.uint32))


def test_cornell_walls_are_filtered_as_pairs(api, O):
    # GlassCube: 5 walls, 5 faces of the block, the mirror = 11 pairs of 22 triangles; the walls
    # and the block's top lie in axis planes. The five walls are faces of one cube; the block's base
    # is no parallelogram, so its four sides make two two-face boxes around opposite vertical edges.
    sc = api.Scene(O.load_test_model(2))
    assert sc.pair_units() == 11
    assert 6 <= sc.axis_pair_units() <= 7
    assert sc.box_units() == 3
    sc.close()
    sc = api.Scene(O.load_test_model(0))
    assert sc.pair_units() >= 13   # 5 walls + 4 side faces of each block (their tops are no parallelograms)
    assert sc.axis_pair_units() >= 5
    assert sc.box_units() == 5
    sc.close()


def box_scene(O, dtype, n_boxes, seed, sizes=(-1.3, 0.0)):
    """Parallelepipeds whose faces are pairs of consecutive triangles: turned and sheared frames,
    sizes over a decade, faces missing at random, either diagonal, any vertex labelling, either order
    of a face's two triangles; every fourth box has one corner moved (within and beyond the tolerance
    of the filter's box units, and far beyond it). 12 triangles per whole box, so boxes also straddle
    the chunks of 32 triangles the filter works in."""
    import bigscenes
    rng = np.random.default_rng(seed)
    v0s, v1s, v2s = [], [], []
    for k in range(n_boxes):
        size = 10 ** rng.uniform(*sizes)
        r = rng.uniform(-0.8, 0.8, 3)
        if k % 3 == 0:      # axis-aligned
            F = np.diag(rng.uniform(0.3, 1.0, 3) * rng.choice([-1.0, 1.0], 3))
        else:               # turned, and sheared for every other one
            F, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            F = F * rng.uniform(0.3, 1.0, 3)
            if k % 2 == 0:
                F[:, 2] += F[:, 0] * rng.uniform(-0.5, 0.5)
        F = F * size
        corner = {}
        for i in (0, 1):
            for j in (0, 1):
                for l in (0, 1):
                    corner[(i, j, l)] = np.float32(r + F[:, 0] * i + F[:, 1] * j + F[:, 2] * l)
        if k % 4 == 3:
            corner[(1, 1, 1)] = np.float32(corner[(1, 1, 1)] + F[:, 0] * 10 ** rng.choice([-7.5, -5.0, -2.0]))
        for w in range(3):
            for side in (0, 1):
                if rng.uniform() < 0.25:
                    continue
                ip, iq = (w + 1) % 3, (w + 2) % 3
                def at(p, q):
                    c = [0, 0, 0]
                    c[w], c[ip], c[iq] = side, p, q
                    return corner[tuple(c)]
                P = [at(0, 0), at(1, 0), at(1, 1), at(0, 1)]
                if rng.uniform() < 0.5:
                    P = P[1:] + P[:1]        # the other diagonal
                ta, tb = [P[0], P[1], P[2]], [P[0], P[2], P[3]]
                if rng.uniform() < 0.5:
                    ta, tb = tb, ta
                for t in (ta, tb):
                    o = rng.permutation(3)
                    v0s.append(t[o[0]]); v1s.append(t[o[1]]); v2s.append(t[o[2]])
    n = len(v0s)
    color = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    return bigscenes._finish(dtype, np.array(v0s), np.array(v1s), np.array(v2s), color, np.zeros(n, dtype=np.uint32))


@pytest.mark.parametrize("seed,sizes", [(11, (-1.3, 0.0)), (12, (-1.3, 0.0)), (13, (-3.0, -1.0))])
def test_box_units_equal_literal_scan(api, O, pkg, seed, sizes):
    # faces of parallelepipeds are filtered in the parallelepiped's frame (box units); the scan's
    # results must stay those of the literal scan and of the oracle
    tris = box_scene(O, pkg.abi.TRIANGLE_DTYPE, 24, seed, sizes)
    assert len(tris) < 256
    sc = api.Scene(tris)
    assert not sc.bvh_info()[0]
    if sizes[0] > -2:
        assert sc.box_units() >= 12
    _, _, closest_bad, any_bad = sc.selfcheck(1 << 26, seed=seed)
    assert (closest_bad, any_bad) == (0, 0)
    o, d = raygen.surface_rays(tris, 1 << 14, seed=seed)
    idx, t, pos = sc.closest_hit(o, d)
    occ = sc.any_hit(o, d, np.full(len(o), 0.7, dtype=np.float32))
    sc.close()
    oidx, ot, opos = O.closest_hit(tris, None, o, d)
    assert np.array_equal(idx, oidx)
    assert np.array_equal(bits(t[idx >= 0]), bits(ot[oidx >= 0]))
    assert np.array_equal(occ, O.any_hit(tris, o, d, np.full(len(o), 0.7, dtype=np.float32)))
    # rays in the faces' planes, along their edges, with zero, huge and tiny directions
    o, d = raygen.degenerate_rays(tris, 4096, seed=seed)
    check_closest(api, O, tris, None, o, d)


def test_eight_two_face_boxes_in_one_chunk(api, O, pkg):
    # eight L-shaped quad pairs fill a chunk of 32 triangles: seven become box units, the eighth stays two
    # pair units (the chunk's count field holds 7); the shipped scan = the literal scan on generated rays,
    # and closest hits = the oracle's
    from test_host_library import l_shapes
    for n in (8, 16):
        tris = l_shapes(pkg.abi, n)
        sc = api.Scene(tris)
        assert sc.box_units() == 7 * (n // 8) and sc.pair_units() == 2 * n
        _, _, bad_closest, bad_any = sc.selfcheck(1 << 24, seed=n)
        assert bad_closest == 0 and bad_any == 0
        rng = np.random.default_rng(n)
        m = 4096
        o = np.ones((m, 4), dtype=np.float32)
        o[:, :3] = rng.uniform(-1, 1, (m, 3))
        k = rng.integers(0, len(tris), m)       # aimed at a point of some triangle (it may hit another first)
        u = rng.uniform(0, 1, m)
        v = rng.uniform(0, 1, m) * (1 - u)
        target = tris["v0"][k, :3] + u[:, None] * (tris["v1"][k, :3] - tris["v0"][k, :3]) + v[:, None] * (tris["v2"][k, :3] - tris["v0"][k, :3])
        d = np.ones((m, 4), dtype=np.float32)
        d[:, :3] = target - o[:, :3] + rng.normal(scale=0.02, size=(m, 3))
        idx, t, pos = sc.closest_hit(o, d)
        oidx, ot, opos = O.closest_hit(tris, None, o, d)
        hit = oidx >= 0
        assert hit.sum() > 1000 and np.array_equal(idx, oidx)
        assert np.array_equal(bits(t[hit]), bits(ot[hit])) and np.array_equal(bits(pos[hit]), bits(opos[hit]))
        sc.close()


@pytest.mark.parametrize("seed", [4, 5])
def test_axis_plane_pairs_equal_literal_scan(api, O, pkg, seed):
    # quads in planes normal to a coordinate axis take the filter's axis form (no cross products)
    tris = quad_scene(O, pkg.abi.TRIANGLE_DTYPE, 60, seed, axis_planes=True)
    sc = api.Scene(tris)
    assert sc.axis_pair_units() >= 20
    _, _, closest_bad, any_bad = sc.selfcheck(1 << 26, seed=seed)
    assert (closest_bad, any_bad) == (0, 0)
    o, d = raygen.surface_rays(tris, 1 << 14, seed=seed)
    idx, t, pos = sc.closest_hit(o, d)
    sc.close()
    oidx, ot, opos = O.closest_hit(tris, None, o, d)
    assert np.array_equal(idx, oidx)
    assert np.array_equal(bits(t[idx >= 0]), bits(ot[oidx >= 0]))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pair_filter_equals_literal_scan_on_random_quads(api, O, pkg, seed):
    tris = quad_scene(O, pkg.abi.TRIANGLE_DTYPE, 60, seed)   # 120 triangles: linear scan, 4 chunks
    sc = api.Scene(tris)
    assert not sc.bvh_info()[0]
    assert sc.pair_units() >= 20
    _, _, closest_bad, any_bad = sc.selfcheck(1 << 26, seed=seed)
    assert (closest_bad, any_bad) == (0, 0)
    o, d = raygen.surface_rays(tris, 1 << 14, seed=seed)
    idx, t, pos = sc.closest_hit(o, d)
    sc.close()
    oidx, ot, opos = O.closest_hit(tris, None, o, d)
    assert np.array_equal(idx, oidx)


@pytest.mark.parametrize("scene,fog", [("cornell", 0.0), ("glasscube", 0.0), ("glasscube", 0.3)])
def test_production_kernel_equals_counting_kernel(api, O, pkg, scene, fog):
    # the counting variant scans every triangle in index order and casts every shadow ray, like
    # the reference; the shipped variant filters candidates (pair units, two phases) and ends a
    # path at a terminal vertex the light cannot reach. Same image, bit for bit.
    which, glass = SCENES[scene]
    tris = O.load_test_model(which)
    W, H, spp = 96, 80, 32
    cam, light = O.default_camera(W), O.default_light(glass)
    sc = api.Scene(tris)
    for rng_mode in (pkg.abi.RNG_MINSTD_PATH, pkg.abi.RNG_COUNTER):
        plain = sc.render_accumulate(cam, light, O.default_params(W, H, seed=5, rng_mode=rng_mode, fog_sigma_t=fog), 0, spp)
        counted = sc.render_accumulate(cam, light, O.default_params(W, H, seed=5, rng_mode=rng_mode, fog_sigma_t=fog,
                                                                    flags=pkg.abi.FLAG_STATS), 0, spp)
        assert np.array_equal(bits(plain), bits(counted)), rng_mode
        assert plain.sum() > 0
    sc.close()


def test_specular_material_on_a_linear_scan_scene(api, O, pkg):
    # bounceSpecular (pathtracer.cpp:65-70): no shipped scene assigns "specular"; here the tall block
    # of the Cornell model carries it. Counted and compared with the oracle, paths bit for bit.
    tris = O.load_test_model(0).copy()
    tris["material"][20:30] = pkg.abi.MAT_SPECULAR
    W, H = 64, 48
    cam, light = O.default_camera(W), O.default_light(False)
    sc = api.Scene(tris)
    assert not sc.bvh_info()[0]
    p = O.default_params(W, H, seed=13, flags=pkg.abi.FLAG_STATS)
    img = sc.render_accumulate(cam, light, p, 0, 8)
    st = sc.stats()
    ref, ost = O.render_accumulate(tris, None, cam, light, p, 0, 8)
    assert st.specular_bounces == ost.specular_bounces and st.specular_bounces > 500
    for k in ("closest_rays", "shadow_rays", "diffuse_bounces", "mirror_bounces", "glass_bounces"):
        assert getattr(st, k) == getattr(ost, k), k
    assert np.array_equal(bits(img), bits(ref))
    plain = sc.render_accumulate(cam, light, O.default_params(W, H, seed=13), 0, 8)
    assert np.array_equal(bits(plain), bits(img))
    px, py, smp = path_list(W, H, 4000, 8)
    rgb, counts = sc.trace_paths(cam, light, p, px, py, smp)
    orgb, ocounts = O.trace_paths(tris, None, cam, light, p, px, py, smp)
    assert np.array_equal(counts, ocounts) and np.array_equal(bits(rgb), bits(orgb))
    sc.close()


def test_progressive_frame_equals_host_round_trips(api, O):
    # pixels + samplesSeenSoFar resident on the device: same sums and the same ARGB screen as
    # render_accumulate / resolve_argb through host buffers; reset starts over
    tris = O.load_test_model(2)
    W, H = 72, 56
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, seed=11, fog_sigma_t=0.2)
    sc = api.Scene(tris)
    fr = api.Frame(sc, W, H)
    for n in (1, 4, 7):
        fr.add_samples(cam, light, p, n)
    assert fr.samples == 12
    screen = fr.present()
    pixels = fr.read_pixels()
    acc = None
    begin = 0
    for n in (1, 4, 7):
        acc = sc.render_accumulate(cam, light, p, begin, n, accum=acc)
        begin += n
    assert np.array_equal(bits(pixels), bits(acc))
    assert np.array_equal(screen, sc.resolve_argb(acc, 12))
    fr.reset()
    assert fr.samples == 0 and not fr.read_pixels().any()
    assert fr.poll()            # nothing queued
    fr.add_samples(cam, light, p, 12)
    polls = 0
    while not fr.poll():        # non-blocking: the host keeps its event loop going
        polls += 1
        assert polls < 10 ** 7
    whole = fr.read_pixels()
    assert np.allclose(whole, acc, rtol=2e-6, atol=1e-7)   # one call of 12 = 1 + 4 + 7 up to the order of the sums
    # in whole work items the progressive image is THE image: 16 + 32 + 16 samples = one call of 64
    fr.reset()
    for n in (16, 32, 16):
        fr.add_samples(cam, light, p, n)
    assert np.array_equal(bits(fr.read_pixels()), bits(sc.render_accumulate(cam, light, p, 0, 64)))
    fr.close()
    sc.close()


def test_two_frames_of_one_scene_interleave(api, O):
    # two frames share their scene's scratch memory (queue head, item sums) but run on their own
    # streams: calls that alternate between them must not overlap on the device
    tris = O.load_test_model(2)
    W, H = 256, 192
    cam, light = O.default_camera(W), O.default_light(True)
    pa = O.default_params(W, H, seed=21)
    pb = O.default_params(W, H, seed=22, fog_sigma_t=0.3)
    sc = api.Scene(tris)
    fa, fb = api.Frame(sc, W, H), api.Frame(sc, W, H)
    for _ in range(6):          # launches of a few milliseconds each, queued back to back
        fa.add_samples(cam, light, pa, 48)
        fb.add_samples(cam, light, pb, 16)
    a, b = fa.read_pixels(), fb.read_pixels()
    fa.close(); fb.close()
    ra = rb = None
    for k in range(6):
        ra = sc.render_accumulate(cam, light, pa, 48 * k, 48, accum=ra)
        rb = sc.render_accumulate(cam, light, pb, 16 * k, 16, accum=rb)
    assert np.array_equal(bits(a), bits(ra))
    assert np.array_equal(bits(b), bits(rb))
    # a frame outliving its scene is destroyed without touching the scene
    fc = api.Frame(sc, 8, 8)
    sc.close()
    fc.close()


@pytest.mark.parametrize("with_sphere", [False, True])
def test_maximum_depth_long_paths(api, O, pkg, with_sphere):
    # MIPT_MAX_DEPTH_LIMIT levels with Russian roulette nearly off: the per-lane vertex stack (and
    # the sphere normals beside it) at its largest, every path many bounces long
    tris = O.load_test_model(2)
    sph = api.make_sphere(0.3, [-0.45, 0.7, -0.55, 1], [1, 0, 0]) if with_sphere else None
    W = H = 32
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, max_depth=16, rr_prob=0.02, flags=pkg.abi.FLAG_STATS, seed=4)
    sc = api.Scene(tris, sph)
    img = sc.render_accumulate(cam, light, p, 0, 4)
    st = sc.stats()
    sc.close()
    ref, ost = O.render_accumulate(tris, sph, cam, light, p, 0, 4)
    assert (st.closest_rays, st.shadow_rays) == (ost.closest_rays, ost.shadow_rays)
    assert st.closest_rays > 5 * W * H * 4
    assert np.array_equal(bits(img), bits(ref))
    p.max_depth = 17
    with pytest.raises(api.MiptError):
        api.Scene(tris).render_accumulate(cam, light, p, 0, 1)


def test_full_size_launch_properties(api, O, pkg):
    # BASELINE's quoted configuration in full: GlassCube, 1024 x 1024, 1024 samples per pixel in one
    # launch (2^30 paths). No CPU can check that many paths, so: the shipped kernel against the
    # counting kernel (every triangle in index order, every shadow ray, like the reference) bit for
    # bit; the counters against what the reference's loop structure implies; two half launches
    # against the whole one.
    import torch
    abi = pkg.abi
    tris = O.load_test_model(2)
    S, spp = 1024, 1024
    cam, light = O.default_camera(S), O.default_light(True)
    sc = api.Scene(tris)
    dev = torch.device("cuda", 0)

    def launch(begin, count, flags=0, into=None):
        acc = torch.zeros(S * S * 3, dtype=torch.float32, device=dev) if into is None else into
        p = O.default_params(S, S, seed=0, rng_mode=abi.RNG_COUNTER, fog_sigma_t=0.1, flags=flags)
        sc.render_accumulate_dev(cam, light, p, begin, count, acc)
        torch.cuda.synchronize()
        return acc

    whole = launch(0, spp)
    counted = launch(0, spp, flags=abi.FLAG_STATS)
    st = sc.stats()
    assert torch.equal(whole.view(torch.int32), counted.view(torch.int32))
    assert st.paths == S * S * spp
    # every path casts at least its camera ray and at most MAX_DEPTH + 1 segments and one shadow ray
    assert st.paths <= st.closest_rays <= 11 * st.paths
    assert st.shadow_rays <= st.paths
    assert 3.0 < (st.closest_rays + st.shadow_rays) / st.paths < 6.0
    # closestIntersection looks at every triangle, existsIntersection stops at the first blocker
    assert st.closest_rays * len(tris) <= st.tri_tests <= (st.closest_rays + st.shadow_rays) * len(tris)
    halves = launch(0, spp // 2)
    halves = launch(spp // 2, spp // 2, into=halves)
    sc.close()
    w = whole.cpu().numpy()
    # (sums can be negative: the reference weights a bounce by an unclamped 4-component dot product)
    assert np.isfinite(w).all() and w.sum() > 0
    # two launches of half the samples leave the bits of the one launch
    assert torch.equal(whole.view(torch.int32), halves.view(torch.int32))


def test_scenes_used_from_several_host_threads_at_once(api, O, pkg):
    # the header's threading contract: one scene is for one host thread at a time, different scenes may
    # be used concurrently. Four threads, each with its own scene (linear scan, linear scan with a
    # sphere, a tree, the same tree forced to the linear scan), render, query and trace at the same
    # time, several times over; every result is the one the same calls give when made alone.
    import threading
    import bigscenes
    dtype = pkg.abi.TRIANGLE_DTYPE
    mesh = bigscenes.mesh_in_room(O, dtype, cells=24, nu=32, nv=16)
    sphere = api.make_sphere(0.3, [-0.45, 0.7, -0.55, 1], [1, 0, 0])
    setups = [(O.load_test_model(2), None, {}), (O.load_test_model(0), sphere, {}), (mesh, None, {}), (mesh, None, {"linear_scan": True})]
    W, H = 72, 56
    cam, light = api.default_camera(W), api.default_light(True)
    o, d = raygen.interior_rays(1 << 14)
    lim = np.full(len(o), 0.7, dtype=np.float32)

    def work(k, scene, rounds):
        out = []
        p = api.default_params(W, H, seed=10 + k, fog_sigma_t=0.1 if k % 2 else 0.0)
        for r in range(rounds):
            img = scene.render_accumulate(cam, light, p, 16 * r, 16)
            idx, t, pos = scene.closest_hit(o, d)
            occ = scene.any_hit(o, d, lim)
            out.append((bits(img).copy(), idx.copy(), bits(t[idx >= 0]).copy(), occ.copy()))
        return out

    scenes = [api.Scene(t, sp, **kw) for t, sp, kw in setups]
    alone = [work(k, sc, 3) for k, sc in enumerate(scenes)]
    together = [None] * len(scenes)
    errors = []

    def run(k):
        try:
            together[k] = work(k, scenes[k], 3)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    for _ in range(3):
        threads = [threading.Thread(target=run, args=(k,)) for k in range(len(scenes))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        for k in range(len(scenes)):
            for a, b in zip(alone[k], together[k]):
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), k
    for sc in scenes:
        sc.close()
