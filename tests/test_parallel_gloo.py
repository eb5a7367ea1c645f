"""The sample-split + sum-reduce decomposition with 2 and 3 ranks over gloo, on CPU.

The renderer plugged into parallel.render_distributed here is the CPU oracle (the GPU one has
the same "add samples [b, b+n) into the image" contract); the check is that N ranks together
produce the image one rank produces for the whole range.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import __graft_entry__ as g
    import oracle_lib as O
    pkg = g.load_package()
    from computer_graphics_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 24, 20
    tris = O.load_test_model(2)
    cam, light = O.default_camera(W), O.default_light(True)
    params = O.default_params(W, H, rng_mode=pkg.abi.RNG_COUNTER, seed=4)

    def render_fn(begin, count, image):
        acc = image.numpy().reshape(W, H, 3)
        O.render_accumulate(tris, None, cam, light, params, begin, count, accum=acc)

    rank_image = torch.zeros(W * H * 3, dtype=torch.float32)
    total = torch.zeros(W * H * 3, dtype=torch.float32) if rank == 0 else None
    # two consecutive calls with ranges that do not divide evenly
    parallel.render_distributed(render_fn, 0, 11, rank, world, rank_image, total)
    parallel.render_distributed(render_fn, 11, 6, rank, world, rank_image, total)
    if rank == 0:
        np.save(out_path, total.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _chain_worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import __graft_entry__ as g
    import oracle_lib as O
    pkg = g.load_package()
    from computer_graphics_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 24, 20
    tris = O.load_test_model(2)
    cam, light = O.default_camera(W), O.default_light(True)
    params = O.default_params(W, H, rng_mode=pkg.abi.RNG_COUNTER, seed=4)
    pending = []

    # the CPU stand-in for Scene.render_items_dev / fold_items_dev: the oracle adds sample by sample
    # into the image it is given, which composes exactly like the GPU's fold of item sums does
    def render_items(begin, count):
        pending.append((begin, count))

    def fold(image):
        begin, count = pending.pop()
        O.render_accumulate(tris, None, cam, light, params, begin, count, accum=image.numpy().reshape(W, H, 3))

    image = torch.zeros(W * H * 3, dtype=torch.float32)
    if rank:
        image += 123.0   # only rank 0's contents count on entry
    parallel.render_chained(render_items, fold, 0, 41, rank, world, image)   # 3 items: 16 + 16 + 9
    parallel.render_chained(render_items, fold, 41, 70, rank, world, image)  # the image lives on, on rank 0
    if rank == 0:
        np.save(out_path, image.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_chained_ranks_leave_the_bits_of_one_rank(tmp_path, world, O, pkg):
    # render_chained: whole work items per rank, the image passed from rank to rank -- not close to
    # the single-rank image but IT, whatever the number of ranks (5 ranks for 3 items: two forward only)
    out = str(tmp_path / "img.npy")
    mp.spawn(_chain_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out).reshape(24, 20, 3)
    tris = O.load_test_model(2)
    cam, light = O.default_camera(24), O.default_light(True)
    params = O.default_params(24, 20, rng_mode=pkg.abi.RNG_COUNTER, seed=4)
    want, _ = O.render_accumulate(tris, None, cam, light, params, 0, 111)
    assert want.sum() > 0
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want, dtype=np.float32).view(np.uint32))


def test_rank_item_range_keeps_work_items_whole(pkg):
    from computer_graphics_amd import parallel
    for count in (0, 1, 15, 16, 17, 100, 1024, 4099):
        for world in (1, 2, 3, 8):
            spans = [parallel.rank_item_range(5, count, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == count
            pos = 5
            for b, c in spans:
                assert c == 0 or b == pos
                pos += c
            busy = [c for _, c in spans if c]
            assert all(c % parallel.ITEM_SAMPLES == 0 for c in busy[:-1])


@pytest.mark.parametrize("world", [2, 3])
def test_sample_split_equals_single_rank(tmp_path, world, O, pkg):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out).reshape(24, 20, 3)
    tris = O.load_test_model(2)
    cam, light = O.default_camera(24), O.default_light(True)
    params = O.default_params(24, 20, rng_mode=pkg.abi.RNG_COUNTER, seed=4)
    want, _ = O.render_accumulate(tris, None, cam, light, params, 0, 17)
    assert want.sum() > 0
    # the same 17 samples per pixel, added in a different order
    assert np.allclose(got, want, rtol=1e-5, atol=1e-7)


def test_rank_sample_range_partitions(pkg):
    from computer_graphics_amd import parallel
    for count in (0, 1, 7, 1024, 4096):
        for world in (1, 2, 3, 8):
            spans = [parallel.rank_sample_range(5, count, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == count
            pos = 5
            for b, c in spans:
                assert b == pos
                pos += c
    with pytest.raises(ValueError):
        parallel.rank_sample_range(0, 4, 2, 2)


def test_bench_launcher_starts_the_ranks(tmp_path, O, pkg, capfd):
    # `python bench.py --gpus N` without torchrun: bench.spawn_ranks starts N rank processes with
    # the rendezvous environment set. Here with a CPU stand-in for the rank (gloo, oracle renderer)
    import json
    import bench
    out = str(tmp_path / "total.npy")
    stub = os.path.join(ROOT, "tests", "bench_rank_stub.py")
    steps, warmup, spp = 2, 1, 7
    rc = bench.spawn_ranks(2, [str(steps), str(warmup), str(spp), out], script=stub)
    assert rc == 0
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["value"] > 0
    # rank 0's image holds the samples of all (warmup + steps) steps, from both ranks
    got = np.load(out).reshape(20, 16, 3)
    tris = O.load_test_model(2)
    cam, light = O.default_camera(20), O.default_light(True)
    params = O.default_params(20, 16, rng_mode=pkg.abi.RNG_COUNTER, seed=9)
    want, _ = O.render_accumulate(tris, None, cam, light, params, 0, spp * (steps + warmup))
    assert want.sum() > 0 and np.allclose(got, want, rtol=1e-5, atol=1e-7)
    # a rank that dies takes the job down with its exit code instead of hanging the others
    assert bench.spawn_ranks(2, [str(steps), str(warmup), str(spp), out, "fail-on-rank-1"], script=stub) == 7


def test_bench_launcher_with_eight_ranks_and_a_rank_that_dies_mid_run(tmp_path, O, pkg, capfd):
    # the driver's scaling run is 8 ranks: the launcher with 8 CPU stand-in ranks (gloo) -- sample split
    # 7 over 8 (one rank gets nothing), one JSON line, the 8-rank image -- and then the same with rank 5
    # dying between two collectives: the launcher ends the seven ranks left waiting in the reduce and
    # returns the dead rank's exit code instead of hanging
    import json
    import time
    import bench
    out = str(tmp_path / "total.npy")
    stub = os.path.join(ROOT, "tests", "bench_rank_stub.py")
    steps, warmup, spp = 1, 1, 7
    assert bench.spawn_ranks(8, [str(steps), str(warmup), str(spp), out], script=stub) == 0
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 8
    got = np.load(out).reshape(20, 16, 3)
    tris = O.load_test_model(2)
    cam, light = O.default_camera(20), O.default_light(True)
    params = O.default_params(20, 16, rng_mode=pkg.abi.RNG_COUNTER, seed=9)
    want, _ = O.render_accumulate(tris, None, cam, light, params, 0, spp * (steps + warmup))
    assert want.sum() > 0 and np.allclose(got, want, rtol=1e-5, atol=1e-7)
    t0 = time.time()
    assert bench.spawn_ranks(8, ["2", "1", str(spp), out, "die-mid-run-on-rank-5"], script=stub) == 9
    assert time.time() - t0 < 120


def test_bench_becomes_the_launcher_for_gpus_above_one(monkeypatch, pkg):
    # the driver's `python bench.py --gpus 8`: no WORLD_SIZE in the environment -> 8 ranks are started
    import bench
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv, **kw: seen.update(n=n, argv=list(argv)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen["n"] == 8 and seen["argv"] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
