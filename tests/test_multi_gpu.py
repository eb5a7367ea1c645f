"""Several GPUs behind the C ABI (mipt_multi_*): the sample split and the RCCL sum-reduce inside
one render call. Without a GPU: the split itself and the argument checks. On the one-GPU box: the
degenerate N = 1 path through RCCL (communicator of one rank, ncclReduce in a group call) against
the plain scene, bit for bit; N > 1 needs several GPUs and is exercised by bench.py there."""
import ctypes as C

import numpy as np
import pytest


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_sample_range_is_the_rank_split_of_the_process_per_gpu_path(pkg):
    from computer_graphics_amd import parallel
    api = pkg.api
    for begin in (0, 5, 4000):
        for count in (0, 1, 7, 1024, 4096, 4099):
            for world in (1, 2, 3, 8):
                spans = [api.multi_sample_range(begin, count, r, world) for r in range(world)]
                assert spans == [parallel.rank_sample_range(begin, count, r, world) for r in range(world)]
                assert sum(c for _, c in spans) == count


def test_item_range_is_the_item_split_of_the_process_per_gpu_path(pkg):
    from computer_graphics_amd import parallel
    api = pkg.api
    for begin in (0, 5, 4000):
        for count in (0, 1, 15, 16, 17, 100, 1024, 4099):
            for world in (1, 2, 3, 8):
                spans = [api.multi_item_range(begin, count, r, world) for r in range(world)]
                assert spans == [parallel.rank_item_range(begin, count, r, world) for r in range(world)]


def test_multi_create_checks_its_arguments(pkg, O):
    abi = pkg.abi
    lib = abi.load_library()
    tris = O.load_test_model(0)
    h = C.c_void_p()
    assert lib.mipt_multi_create(tris.ctypes.data, len(tris), None, 0, None, 1, C.byref(h)) == abi.MIPT_E_INVALID
    devs = (C.c_int * 2)(0, 0)
    assert lib.mipt_multi_create(tris.ctypes.data, len(tris), None, 0, devs, 2, C.byref(h)) == abi.MIPT_E_INVALID
    assert b"twice" in lib.mipt_last_error()
    assert lib.mipt_multi_create(tris.ctypes.data, len(tris), None, 0, devs, 0, C.byref(h)) == abi.MIPT_E_INVALID
    assert lib.mipt_multi_create_opts(tris.ctypes.data, len(tris), None, 0, devs, 2, 0x80, C.byref(h)) == abi.MIPT_E_INVALID
    assert b"option" in lib.mipt_last_error()
    assert lib.mipt_multi_device_count(None) == 0 and lib.mipt_multi_scene(None, 0) is None
    lib.mipt_multi_destroy(None)
    # frames: argument checks that need neither a GPU nor a scene
    f = C.c_void_p()
    assert lib.mipt_multi_frame_create(None, 8, 8, abi.MULTI_FRAME_EXACT, C.byref(f)) == abi.MIPT_E_INVALID
    assert lib.mipt_multi_frame_create(None, 8, 8, 0, None) == abi.MIPT_E_INVALID
    assert lib.mipt_multi_frame_samples(None) == 0 and lib.mipt_multi_frame_device(None) == -1
    assert lib.mipt_multi_frame_reset(None) == abi.MIPT_E_INVALID
    assert lib.mipt_multi_frame_poll(None) == abi.MIPT_E_INVALID
    assert lib.mipt_multi_frame_present(None, None) == abi.MIPT_E_INVALID
    lib.mipt_multi_frame_destroy(None)


@pytest.mark.gpu
def test_one_device_through_rccl_equals_the_plain_scene(pkg, O):
    import torch
    assert torch.cuda.is_available()
    api, abi = pkg.api, pkg.abi
    tris = O.load_test_model(2)
    W, H = 96, 64
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, seed=17, rng_mode=abi.RNG_COUNTER, fog_sigma_t=0.1)
    multi = api.MultiScene(tris, None, devices=(0,))
    assert multi.device_count == 1
    sc = api.Scene(tris)
    got = multi.render_accumulate(cam, light, p, 0, 24)
    got = multi.render_accumulate(cam, light, p, 24, 9, accum=got)       # samples are ADDED
    want = sc.render_accumulate(cam, light, p, 0, 24)
    part = sc.render_accumulate(cam, light, p, 24, 9)
    # the reduce of one image is the image; the second call is added on the host like main.cpp:137
    assert np.array_equal(bits(got), bits(want + part)) and want.sum() > 0
    # the chained variant: upload, fold, download -- the plain scene's bits, also on top of earlier samples
    chained = multi.render_accumulate_chained(cam, light, p, 0, 24)
    chained = multi.render_accumulate_chained(cam, light, p, 24, 9, accum=chained)
    direct = sc.render_accumulate(cam, light, p, 24, 9, accum=want.copy())
    assert np.array_equal(bits(chained), bits(direct))
    # another image size re-allocates the per-device images
    p2 = O.default_params(40, 56, seed=3)
    c2 = O.default_camera(40)
    assert np.array_equal(bits(multi.render_accumulate(c2, light, p2, 0, 5)), bits(sc.render_accumulate(c2, light, p2, 0, 5)))
    multi.close(); sc.close()
    with pytest.raises(api.MiptError) as e:
        api.MultiScene(tris, None, devices=(0, 99))
    assert e.value.code in (abi.MIPT_E_INVALID, abi.MIPT_E_HIP, abi.MIPT_E_RCCL)


@pytest.mark.gpu
def test_bench_with_two_ranks_rehearsed_on_one_gpu(pkg, tmp_path):
    # `python bench.py --gpus 2` as the driver starts it (the script becomes the launcher of two rank
    # processes), with both ranks on GPU 0 and gloo in place of RCCL: sample split, per-rank launches,
    # reduce to rank 0, one JSON line from rank 0. The summed image equals one scene's rendering of
    # the same samples.
    import json
    import os
    import subprocess
    import sys
    import torch
    import bench
    api, abi = pkg.api, pkg.abi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "total.npy")
    spp, steps, warmup = 6, 1, 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse", "--steps", str(steps),
                        "--warmup", str(warmup), "--spp", str(spp), "--no-cpu", "--save-total", out],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and "rehearsal" in line
    assert line["config"]["spp_per_gpu_per_step"] == spp // 2 and line["config"]["global_spp_per_step"] == spp
    got = np.load(out)
    S = bench.SIZE
    tris = api.load_test_model(abi.SCENE_GLASS_CUBE)
    cam, light = api.default_camera(S), api.default_light(glass=True)
    p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER, seed=0, fog_sigma_t=bench.FOG_SIGMA_T,
                           fog_albedo=bench.FOG_ALBEDO)
    sc = api.Scene(tris)
    acc = torch.zeros(S * S * 3, dtype=torch.float32, device="cuda")
    sc.render_accumulate_dev(cam, light, p, 0, spp * (steps + warmup), acc)
    torch.cuda.synchronize()
    want = acc.cpu().numpy()
    sc.close()
    assert want.sum() > 0
    # the same samples, added in another order (per rank, then per step)
    assert np.allclose(got, want, rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_chained_ranks_on_one_gpu_leave_the_bits_of_one_scene(pkg, tmp_path, ranks):
    # `bench.py --gpus N --exact`, rehearsed on one GPU: every rank traces whole work items, the image
    # makes the round of the ranks and each folds its items in. The result is not close to the
    # single-scene image, it IS that image -- with 2 ranks and with 3
    import json
    import os
    import subprocess
    import sys
    import torch
    import bench
    api, abi = pkg.api, pkg.abi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "total.npy")
    spp, steps, warmup = 56, 1, 1       # 4 work items per step (16 + 16 + 16 + 8), 2 steps
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--rehearse", "--exact",
                        "--steps", str(steps), "--warmup", str(warmup), "--spp", str(spp), "--no-cpu", "--save-total", out],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == ranks and "bit-identical to one GPU" in line["config"]["workload"]
    got = np.load(out)
    S = bench.SIZE
    tris = api.load_test_model(abi.SCENE_GLASS_CUBE)
    cam, light = api.default_camera(S), api.default_light(glass=True)
    p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER, seed=0, fog_sigma_t=bench.FOG_SIGMA_T,
                           fog_albedo=bench.FOG_ALBEDO)
    sc = api.Scene(tris)
    acc = torch.zeros(S * S * 3, dtype=torch.float32, device="cuda")
    for k in range(steps + warmup):      # one GPU, the same steps
        sc.render_accumulate_dev(cam, light, p, k * spp, spp, acc)
    torch.cuda.synchronize()
    want = acc.cpu().numpy()
    sc.close()
    assert want.sum() > 0 and np.array_equal(bits(got), bits(want)), np.count_nonzero(bits(got) != bits(want))


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [2, 3])
def test_relay_over_parts_sharing_one_gpu_is_the_one_scene_image(pkg, O, parts):
    # MIPT_MULTI_ALLOW_SHARED_DEVICE: 2 and 3 scene replicas, streams and images on GPU 0. Everything the
    # several-GPU relay does runs here -- the split in whole work items, all parts tracing at once, the
    # event chain, hipMemcpyPeerAsync from part to part, the folds -- and leaves the bits ONE scene leaves.
    import torch
    assert torch.cuda.is_available()
    api, abi = pkg.api, pkg.abi
    tris = O.load_test_model(2)
    W, H = 200, 136
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, seed=5, rng_mode=abi.RNG_COUNTER, fog_sigma_t=0.1)
    multi = api.MultiScene(tris, None, devices=(0,) * parts, allow_shared_device=True)
    assert multi.device_count == parts
    sc = api.Scene(tris)
    # 56 samples = 4 work items over 2 or 3 parts; then 9 more on top (one ragged item: one part works);
    # then 100 (7 items)
    got = want = None
    begin = 0
    for n in (56, 9, 100):
        got = multi.render_accumulate_chained(cam, light, p, begin, n, accum=got)
        want = sc.render_accumulate(cam, light, p, begin, n, accum=want)
        begin += n
        assert np.array_equal(bits(got), bits(want)), (parts, n)
    assert want.sum() > 0
    # the sum-reduce needs one RCCL rank per device: refused on such a scene, and the scene stays usable
    with pytest.raises(api.MiptError) as e:
        multi.render_accumulate(cam, light, p, 0, 8)
    assert e.value.code == abi.MIPT_E_UNSUPPORTED
    again = multi.render_accumulate_chained(cam, light, p, 0, 56)
    assert np.array_equal(bits(again), bits(sc.render_accumulate(cam, light, p, 0, 56)))
    multi.close(); sc.close()


@pytest.mark.gpu
def test_multi_frame_stays_on_the_devices_and_equals_the_one_gpu_frame(pkg, O):
    # mipt_multi_frame_*: the progressive image lives on the devices; only the ARGB screen comes back.
    # EXACT mode over 3 parts on GPU 0 = mipt_frame_* of one scene, bit for bit (pixels and screen),
    # wherever the image ended its round; REDUCE mode with one device through RCCL likewise.
    import torch
    assert torch.cuda.is_available()
    api, abi = pkg.api, pkg.abi
    tris = O.load_test_model(2)
    W, H = 176, 120
    cam, light = O.default_camera(W), O.default_light(True)
    p = O.default_params(W, H, seed=9, rng_mode=abi.RNG_COUNTER, fog_sigma_t=0.1)
    sc = api.Scene(tris)
    one = api.Frame(sc, W, H)
    multi = api.MultiScene(tris, None, devices=(0, 0, 0), allow_shared_device=True)
    fr = api.MultiFrame(multi, W, H, exact=True)
    homes = []
    for n in (48, 16, 64, 32):      # 3, 1, 4, 2 work items: the round ends on a different part each time
        fr.add_samples(cam, light, p, n)
        one.add_samples(cam, light, p, n)
        homes.append(fr.device_index)
    assert fr.samples == one.samples == 160
    assert len(set(homes)) > 1, homes
    polls = 0
    while not fr.poll():
        polls += 1
        assert polls < 10 ** 7
    assert np.array_equal(bits(fr.read_pixels()), bits(one.read_pixels()))
    assert np.array_equal(fr.present(), one.present())
    # reset_evolutionModel: zero sums, sample count 0, and the frame goes on where the image is
    fr.reset(); one.reset()
    assert fr.samples == 0 and not fr.read_pixels().any()
    fr.add_samples(cam, light, p, 32); one.add_samples(cam, light, p, 32)
    assert np.array_equal(bits(fr.read_pixels()), bits(one.read_pixels()))
    with pytest.raises(api.MiptError) as e:     # no communicator on a shared device
        api.MultiFrame(multi, W, H, exact=False)
    assert e.value.code == abi.MIPT_E_UNSUPPORTED
    fr.close(); multi.close()
    # the sum-reduce frame, one device: root renders straight into the frame's image, reduce of one rank
    m1 = api.MultiScene(tris, None, devices=(0,))
    f1 = api.MultiFrame(m1, W, H, exact=False)
    one.reset()
    for n in (16, 40, 7):
        f1.add_samples(cam, light, p, n)
        one.add_samples(cam, light, p, n)
    assert f1.device_index == 0 and f1.samples == 63
    assert np.array_equal(bits(f1.read_pixels()), bits(one.read_pixels()))
    assert np.array_equal(f1.present(), one.present())
    f1.close(); m1.close(); one.close(); sc.close()


@pytest.mark.gpu
def test_bench_through_the_rccl_backend_with_one_rank(pkg):
    # `bench.py --force-dist`: a world of one rank goes through init_process_group("nccl", device_id=...),
    # the barrier, dist.reduce of the image and the MAX all-reduce of the timing -- torch's RCCL backend
    # comes up on this box the way the N-GPU run uses it
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "1", "--warmup", "1",
                        "--spp", "16", "--no-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line.get("collectives") == "nccl"
