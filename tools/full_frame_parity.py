#!/usr/bin/env python3
"""One-off check at the bench's full size: the GPU's image of BASELINE configs[1] (GlassCube, 1024 x 1024,
fog, counter RNG; --spp samples, default 1024) against the CPU oracle's, the oracle adding a pixel's samples
16 at a time like the GPU's work items. Prints the number of differing channel sums (expected 0).

    python tools/full_frame_parity.py [--spp 1024] [--size 1024] [--threads 16]
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
import bench  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--size", type=int, default=bench.SIZE)
    ap.add_argument("--threads", type=int, default=int(os.environ.get("OMP_NUM_THREADS", "8")))
    ap.add_argument("--scene", type=int, default=2, help="test model: 0 Cornell, 1 glass cube + room for the sphere, 2 GlassCube (bench), 3 empty room")
    ap.add_argument("--rng", choices=["counter", "minstd"], default="counter")
    ap.add_argument("--fog", type=float, default=bench.FOG_SIGMA_T)
    ap.add_argument("--sphere", action="store_true", help="add the reference's sphere (main.cpp:75)")
    args = ap.parse_args()
    pkg = graft.load_package()
    api, abi = pkg.api, pkg.abi
    S = args.size
    tris = api.load_test_model(args.scene)
    spheres = api.make_sphere(0.3, [-0.45, 0.7, -0.55, 1], [1, 0, 0]) if args.sphere else None
    cam, light = api.default_camera(S), api.default_light(glass=args.scene != 0)
    p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER if args.rng == "counter" else abi.RNG_MINSTD_PATH, seed=0,
                           fog_sigma_t=args.fog, fog_albedo=bench.FOG_ALBEDO)
    sc = api.Scene(tris, spheres)
    t0 = time.time()
    acc = sc.render_accumulate(cam, light, p, 0, args.spp)
    t_gpu = time.time() - t0
    sc.close()
    t0 = time.time()
    want = None
    for b in range(0, args.spp, 16):
        img, _ = O.render_accumulate(tris, spheres, cam, light, p, b, min(16, args.spp - b), threads=args.threads)
        want = img.copy() if want is None else (want + img).astype(np.float32)
        print(f"  oracle: {b + 16} of {args.spp} samples, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    a = np.ascontiguousarray(acc).view(np.uint32)
    w = np.ascontiguousarray(want).view(np.uint32)
    print(json.dumps({"workload": f"test model {args.scene}{' + sphere' if args.sphere else ''} {S}x{S}, {args.spp} spp, fog sigma_t {args.fog}, {args.rng} RNG", "paths": S * S * args.spp,
                      "channel_sums": int(a.size), "differing": int(np.count_nonzero(a != w)),
                      "gpu_seconds_incl_copies": round(t_gpu, 3), "oracle_seconds": round(time.time() - t0, 1),
                      "oracle_threads": args.threads}), flush=True)


if __name__ == "__main__":
    main()
