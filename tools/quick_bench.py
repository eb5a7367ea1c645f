#!/usr/bin/env python3
"""Developer micro-benchmark: times the path kernel and the batch closest-hit kernel of one or
more builds of libmipt.so in ONE process, interleaved (A/B deltas between builds are only
meaningful that way).

    python tools/quick_bench.py [lib.so ...] [--spp N] [--rounds R]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024)
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    libs = args.libs or [abi.LIB_PATH]
    dev = torch.device("cuda", 0)
    S = args.size
    rng = np.random.default_rng(0)
    n_rays = 1 << 24
    o = np.ones((n_rays, 4), dtype=np.float32)
    o[:, :3] = rng.uniform(-0.95, 0.95, (n_rays, 3))
    d = rng.normal(size=(n_rays, 4)).astype(np.float32)
    d[:, 3] = 1
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t_o, t_d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    t_idx = torch.zeros(n_rays, dtype=torch.int32, device=dev)
    t_t = torch.zeros(n_rays, dtype=torch.float32, device=dev)
    acc = torch.zeros(S * S * 3, dtype=torch.float32, device=dev)
    results = {}
    for r in range(args.rounds + 1):
        for path in libs:
            abi._lib = None
            lib = abi.load_library(path)
            abi._lib = lib
            tris = api.load_test_model(abi.SCENE_GLASS_CUBE)
            cam, light = api.default_camera(S), api.default_light(True)
            sc = api.Scene(tris)
            row = results.setdefault(os.path.basename(path), {})
            for label, sigma in (("paths_fog_off", 0.0), ("paths_fog_on", 0.1)):
                p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER, fog_sigma_t=sigma)
                acc.zero_()
                sc.render_accumulate_dev(cam, light, p, 0, args.spp, acc)
                torch.cuda.synchronize()
                ms = sc.stats().kernel_ms
                if r:
                    row.setdefault(label, []).append(S * S * args.spp / ms / 1e6)  # Gpaths/s
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sc.closest_hit_dev(t_o, t_d, t_idx, t_t)
            e1.record()
            torch.cuda.synchronize()
            if r:
                row.setdefault("closest_Grays", []).append(n_rays / e0.elapsed_time(e1) / 1e6)
            sc.close()
    for name, row in results.items():
        print(name, {k: f"med {np.median(v):.3f} max {np.max(v):.3f}" for k, v in row.items()})


if __name__ == "__main__":
    main()
