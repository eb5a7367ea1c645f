#!/usr/bin/env python3
"""Developer measurement: what the tree path kernel costs on rays that never meet the tree. The empty
Cornell room (LoadTestModelNone, 10 triangles) through the linear-scan kernel, against the same room
with a speck of 300 tiny triangles in a corner, which makes it a tree scene (walls in the big-triangle
list, the speck in the LBVH): nearly every ray is a room ray in both.

    python tools/room_overhead.py [--spp 64]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import bigscenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    if args.lib:
        abi._lib = abi.load_library(args.lib)
    S = args.size
    room = api.load_test_model(abi.SCENE_NONE)
    speck = bigscenes.soup(abi.TRIANGLE_DTYPE, 300, size=0.004)
    for k in ("v0", "v1", "v2"):
        speck[k][:, :3] = speck[k][:, :3] * 0.05 + np.float32(0.9)
    acc = torch.zeros(S * S * 3, dtype=torch.float32, device="cuda")
    cam, light = api.default_camera(S), api.default_light(True)
    for label, tris, lin in (("room, linear kernel", room, False), ("room + speck, tree kernel", np.concatenate([room, speck]), False),
                             ("room + speck, linear scan", np.concatenate([room, speck]), True)):
        sc = api.Scene(tris, linear_scan=lin)
        for fog in (0.0, 0.1):
            p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER, fog_sigma_t=fog)
            best = 0.0
            for _ in range(3):
                sc.render_accumulate_dev(cam, light, p, 0, args.spp, acc)
                torch.cuda.synchronize()
                best = max(best, S * S * args.spp / sc.stats().kernel_ms / 1e3)
            print(f"{label:30s} fog {fog}: {best:8.1f} Mpaths/s", flush=True)
        sc.close()


if __name__ == "__main__":
    main()
