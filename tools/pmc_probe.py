#!/usr/bin/env python3
"""One render per kernel flavour, for rocprofv3 --pmc passes:
    rocprofv3 --pmc <counters> -d out -o pmc -- python3 tools/pmc_probe.py [--spp N] [--fog S]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--fog", type=float, default=0.0)
    ap.add_argument("--flags", type=int, nargs="*", default=[0])
    ap.add_argument("--no-batch", action="store_true", help="skip the closest-hit batch")
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    S = args.size
    tris = api.load_test_model(abi.SCENE_GLASS_CUBE)
    cam, light = api.default_camera(S), api.default_light(True)
    sc = api.Scene(tris)
    acc = torch.zeros(S * S * 3, dtype=torch.float32, device="cuda")
    for flags in args.flags:
        p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER, fog_sigma_t=args.fog, flags=flags)
        sc.render_accumulate_dev(cam, light, p, 0, args.spp, acc)
        torch.cuda.synchronize()
        print(f"flags={flags} kernel_ms={sc.stats().kernel_ms:.3f}")
    if args.no_batch:
        sc.close()
        return
    # incoherent closest-hit batch (scan only, no shading)
    import numpy as np
    rng = np.random.default_rng(0)
    n = 1 << 22
    o = np.ones((n, 4), dtype=np.float32)
    o[:, :3] = rng.uniform(-0.95, 0.95, (n, 3))
    d = rng.normal(size=(n, 4)).astype(np.float32)
    d[:, 3] = 1
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t_o, t_d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    t_idx = torch.zeros(n, dtype=torch.int32, device="cuda")
    t_t = torch.zeros(n, dtype=torch.float32, device="cuda")
    sc.closest_hit_dev(t_o, t_d, t_idx, t_t)
    torch.cuda.synchronize()
    sc.close()


if __name__ == "__main__":
    main()
