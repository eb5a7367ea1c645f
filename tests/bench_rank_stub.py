"""One rank of bench.py's launcher, for the CPU test of the N > 1 path: gloo instead of RCCL and the
CPU oracle as the renderer behind parallel.render_distributed, on a tiny image. Everything else --
the environment the launcher sets, the sample split, the sum-reduce, the timing contract and the
one JSON line from rank 0 -- is bench.py's own code."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    steps, warmup, global_spp, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    if sys.argv[5:] == ["fail-on-rank-1"] and os.environ["RANK"] == "1":
        sys.exit(7)
    world, rank, _ = bench.rank_env()
    pkg = bench.graft.load_package()
    from computer_graphics_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 20, 16
    tris = O.load_test_model(2)
    cam, light = O.default_camera(W), O.default_light(True)
    params = O.default_params(W, H, rng_mode=pkg.abi.RNG_COUNTER, seed=9)

    def render(begin, count, image):
        O.render_accumulate(tris, None, cam, light, params, begin, count, accum=image.numpy().reshape(W, H, 3))

    buf = torch.zeros(W * H * 3, dtype=torch.float32)
    total = torch.zeros_like(buf) if rank == 0 else None

    def step(k):
        if sys.argv[5:] == ["die-mid-run-on-rank-5"] and rank == 5 and k == 1:
            os._exit(9)      # between two collectives: the other ranks are left waiting in the next reduce
        parallel.render_distributed(render, k * global_spp, global_spp, rank, world, buf, total)

    def max_over_ranks(dt):
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = bench.timed_steps(step, steps, warmup, dist.barrier, max_over_ranks)
    if rank == 0:
        np.save(out_path, total.numpy())
        print(json.dumps({"n_gpus": world, "steps": steps, "warmup": warmup, "value": W * H * global_spp * steps / dt,
                          "scaling": "strong", "global_spp_per_step": global_spp}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
