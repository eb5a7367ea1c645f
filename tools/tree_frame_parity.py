#!/usr/bin/env python3
"""One-off check on the big tree scenes: the GPU's image (LBVH walk, shipped path kernel) against the CPU
oracle's linear scan of the same triangles, the oracle adding a pixel's samples 16 at a time like the GPU's
work items. Prints the number of differing channel sums per scene (expected 0).

    python tools/tree_frame_parity.py [--scenes c4,c5,wolf,mesh140k] [--threads 16]
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
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as graft  # noqa: E402
import oracle_lib as O  # noqa: E402
from bvh_bench import make_scene  # noqa: E402

# image size and samples per scene, sized so that the oracle's linear scan takes about half a minute
SHAPES = {"wolf": (256, 32), "mesh9k": (128, 32), "mesh140k": (96, 16), "c4": (112, 16), "c5": (40, 16)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="wolf,mesh140k,c4,c5")
    ap.add_argument("--threads", type=int, default=int(os.environ.get("OMP_NUM_THREADS", "8")))
    args = ap.parse_args()
    pkg = graft.load_package()
    api, abi = pkg.api, pkg.abi
    for name in args.scenes.split(","):
        S, spp = SHAPES[name]
        tris = make_scene(name, api, abi.TRIANGLE_DTYPE)
        cam, light = api.default_camera(S), api.default_light(glass=True)
        p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER, seed=1, fog_sigma_t=0.1)
        sc = api.Scene(tris)
        acc = sc.render_accumulate(cam, light, p, 0, spp)
        has_tree = sc.bvh_info()[0]
        sc.close()
        t0 = time.time()
        want = None
        for b in range(0, spp, 16):
            img, _ = O.render_accumulate(tris, None, cam, light, p, b, min(16, spp - b), threads=args.threads)
            want = img.copy() if want is None else (want + img).astype(np.float32)
        a = np.ascontiguousarray(acc).view(np.uint32)
        w = np.ascontiguousarray(want).view(np.uint32)
        print(json.dumps({"scene": name, "triangles": int(len(tris)), "tree": bool(has_tree), "image": f"{S}x{S}", "spp": spp,
                          "paths": S * S * spp, "channel_sums": int(a.size), "differing": int(np.count_nonzero(a != w)),
                          "image_sum": float(want.sum()), "oracle_seconds": round(time.time() - t0, 1),
                          "oracle_threads": args.threads}), flush=True)


if __name__ == "__main__":
    main()
