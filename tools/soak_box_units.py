#!/usr/bin/env python3
"""Long on-device comparison of the shipped scan with the literal reference arithmetic on scenes of
random parallelepipeds (turned, sheared, faces missing, corners moved within and beyond the box units'
tolerance; tests/test_gpu_parity.py::box_scene): 2^--log2 generated rays per scene. One line per scene.

    python tools/soak_box_units.py [--scenes 6] [--log2 32]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import oracle_lib as O  # noqa: E402
from test_gpu_parity import box_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=6)
    ap.add_argument("--log2", type=int, default=32)
    args = ap.parse_args()
    pkg = graft.load_package()
    api, abi = pkg.api, pkg.abi
    for k in range(args.scenes):
        sizes = (-1.3, 0.0) if k % 3 != 2 else (-3.0, -1.0)
        tris = box_scene(O, abi.TRIANGLE_DTYPE, 24, 100 + k, sizes)
        sc = api.Scene(tris)
        t0 = time.time()
        _, _, c, a = sc.selfcheck(1 << args.log2, seed=500 + k)
        print(json.dumps({"scene": f"24 random boxes, seed {100 + k}, sizes 10^{sizes}", "triangles": len(tris),
                          "pairs": sc.pair_units(), "boxes": sc.box_units(), "rays": 1 << args.log2,
                          "closest_mismatches": c, "any_hit_mismatches": a, "seconds": round(time.time() - t0, 1)}), flush=True)
        sc.close()


if __name__ == "__main__":
    main()
