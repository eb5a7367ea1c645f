#!/usr/bin/env python3
"""Long on-device comparison of the shipped scene search with the literal reference arithmetic:
many seeds x 2^32 generated rays per Cornell scene, 2^26 per tree scene. Prints one line per scene.

    python tools/soak_selfcheck.py [--seeds 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import bigscenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=16)
    ap.add_argument("--log2", type=int, default=32)
    args = ap.parse_args()
    pkg = graft.load_package()
    api, abi = pkg.api, pkg.abi
    dt = abi.TRIANGLE_DTYPE
    scenes = {f"test model {k}": (api.load_test_model(k), args.log2) for k in range(4)}
    scenes["mesh9k"] = (bigscenes.mesh_in_room(api, dt, cells=48, nu=64, nv=32), 26)
    scenes["soup20k"] = (np.concatenate([bigscenes.room(api), bigscenes.soup(dt, 20000)]), 25)
    scenes["glassWolf"] = (np.fromfile(os.path.join(ROOT, "tests", "golden", "obj", "glasswolf_scene.bin"), dtype=dt), 28)
    for name, (tris, lg) in scenes.items():
        sc = api.Scene(tris)
        t0 = time.time()
        bad_c = bad_a = graz = redo = ovf = 0
        for seed in range(args.seeds):
            _, _, c, a = sc.selfcheck(1 << lg, seed=1000 + seed)
            bad_c += c; bad_a += a
            graz += sc.last_selfcheck_grazing
            redo += sc.last_selfcheck_bvh[0]; ovf += sc.last_selfcheck_bvh[1]
            print(f"  {name} seed {seed}: {c} {a}", file=sys.stderr, flush=True)
        print(json.dumps({"scene": name, "triangles": len(tris), "tree": sc.bvh_info()[0], "rays": args.seeds << lg,
                          "closest_mismatches": bad_c, "any_hit_mismatches": bad_a, "grazing_family_mismatches": graz,
                          "exhaustive_reruns": redo, "overflows": ovf, "seconds": round(time.time() - t0, 1)}), flush=True)
        sc.close()


if __name__ == "__main__":
    main()
