#!/usr/bin/env python3
"""Developer probe: box units of the test-model scenes, device self-check, shipped vs counting image."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

def main():
    import torch  # noqa: F401
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
    for which, glass in ((0, False), (2, True)):
        tris = api.load_test_model(which)
        sc = api.Scene(tris)
        print("scene", which, "pairs", sc.pair_units(), "axis", sc.axis_pair_units(), "boxes", sc.box_units(), flush=True)
        print("  selfcheck", sc.selfcheck(n, seed=3), flush=True)
        W = H = 128
        cam, light = api.default_camera(W), api.default_light(glass)
        for fog in (0.0, 0.2):
            a = sc.render_accumulate(cam, light, api.default_params(W, H, seed=5, fog_sigma_t=fog), 0, 32)
            b = sc.render_accumulate(cam, light, api.default_params(W, H, seed=5, fog_sigma_t=fog, flags=abi.FLAG_STATS), 0, 32)
            print("  fog", fog, "shipped == counting:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(a.sum()), flush=True)
        sc.close()

if __name__ == "__main__":
    main()
