#!/usr/bin/env python3
"""One-off soak: the device self-check (tree walk against the literal linear scan, closest- and any-hit
rule) on tree scenes with many more rays than the collected tests use. One JSON line per scene.

    python tools/soak_tree.py [--log2-rays 28] [--scenes wolf,mesh9k,mesh140k,c4]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import bvh_bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-rays", type=int, default=28)
    ap.add_argument("--scenes", default="wolf,mesh140k,c4")
    args = ap.parse_args()
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    for name in args.scenes.split(","):
        tris = bvh_bench.make_scene(name, api, abi.TRIANGLE_DTYPE)
        sc = api.Scene(tris)
        t0 = time.time()
        rcp_bad, _, closest_bad, any_bad = sc.selfcheck(1 << args.log2_rays, seed=17)
        redo, overflow = sc.last_selfcheck_bvh
        print(json.dumps({"scene": name, "triangles": len(tris), "tree": sc.tree_info(), "rays": 1 << args.log2_rays,
                          "closest_differ": closest_bad, "any_hit_differ": any_bad, "exhaustive_reruns": redo,
                          "stack_overflows": overflow, "grazing_rays_that_differ": sc.last_selfcheck_grazing,
                          "seconds": round(time.time() - t0, 1)}), flush=True)
        sc.close()


if __name__ == "__main__":
    main()
