#!/usr/bin/env python3
"""Developer micro-benchmark: whole-path rate of one or more builds of libmipt.so on tree scenes, in
ONE process, interleaved (A/B deltas between builds are only meaningful that way).

    python tools/tree_ab.py [lib.so ...] [--scenes c4,wolf] [--spp 64] [--rounds 3]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import bvh_bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--scenes", default="c4")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--fog", type=float, default=0.1)
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    libs = args.libs or [abi.LIB_PATH]
    dev = torch.device("cuda", 0)
    S = args.size
    acc = torch.zeros(S * S * 3, dtype=torch.float32, device=dev)
    for name in args.scenes.split(","):
        tris = bvh_bench.make_scene(name, api, abi.TRIANGLE_DTYPE)
        results = {}
        for r in range(args.rounds + 1):
            for path in libs:
                abi._lib = None
                abi._lib = abi.load_library(path)
                cam, light = api.default_camera(S), api.default_light(True)
                sc = api.Scene(tris)
                p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER, fog_sigma_t=args.fog)
                sc.render_accumulate_dev(cam, light, p, 0, args.spp, acc)
                torch.cuda.synchronize()
                ms = sc.stats().kernel_ms
                if r:
                    results.setdefault(os.path.basename(path), []).append(S * S * args.spp / ms / 1e3)  # Mpaths/s
                sc.close()
        for lib, v in results.items():
            print(f"{name:8s} {lib:28s} med {np.median(v):8.1f} max {np.max(v):8.1f} Mpaths/s", flush=True)


if __name__ == "__main__":
    main()
