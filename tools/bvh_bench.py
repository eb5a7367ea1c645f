#!/usr/bin/env python3
"""Developer benchmark of the LBVH path on synthetic scenes standing in for the OBJ meshes the
reference's repository does not ship: tree build time, batch closest-hit / any-hit rates on
coherent and incoherent rays, and whole-path rates.

    python tools/bvh_bench.py [--scenes mesh140k,soup1m] [--spp 16] [--lib path/to/libmipt.so]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import bigscenes  # noqa: E402
import raygen  # noqa: E402


def make_scene(name, api, dtype):
    if name == "mesh9k":
        return bigscenes.mesh_in_room(api, dtype, cells=48, nu=64, nv=32)
    if name == "mesh140k":
        return bigscenes.mesh_in_room(api, dtype, cells=192, nu=256, nv=128)
    if name == "mesh1m":
        return bigscenes.mesh_in_room(api, dtype, cells=512, nu=768, nv=384)
    if name == "soup100k":
        return np.concatenate([bigscenes.room(api), bigscenes.soup(dtype, 100000, size=0.02)])
    if name == "soup1m":
        return np.concatenate([bigscenes.room(api), bigscenes.soup(dtype, 1000000, size=0.01)])
    if name == "wolf":  # the reference's glassWolf scene (main.cpp:78-89), from the committed fixture
        return np.fromfile(os.path.join(ROOT, "tests", "golden", "obj", "glasswolf_scene.bin"), dtype=dtype)
    if name == "c4":
        return bigscenes.config4(api, dtype)
    if name == "c5":
        return bigscenes.config5(dtype)
    raise SystemExit(f"unknown scene {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="c4,c5")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--rays", type=int, default=1 << 22)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    import torch
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    if args.lib:
        abi._lib = abi.load_library(args.lib)
    dev = torch.device("cuda", 0)
    n = args.rays
    S = args.size

    def timed(fn, reps=3):
        best = 1e30
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    for name in args.scenes.split(","):
        n = 3840 * 2160 if name == "c5" else args.rays  # configs[4] names a 3840x2160 ray batch
        tris = make_scene(name, api, abi.TRIANGLE_DTYPE)
        api.Scene(tris[:1000]).close()  # first use of the sort pays one-off set-up
        sc = api.Scene(tris)
        has, nodes, build_ms = sc.bvh_info()
        row = {"scene": name, "triangles": len(tris), "bvh": has, "build_ms": round(build_ms, 3)}
        idx = torch.zeros(n, dtype=torch.int32, device=dev)
        tt = torch.zeros(n, dtype=torch.float32, device=dev)
        occ = torch.zeros(n, dtype=torch.uint8, device=dev)
        for kind, (o, d) in {"primary": raygen.primary_rays(n), "incoherent": raygen.interior_rays(n)}.items():
            to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
            ms = timed(lambda: sc.closest_hit_dev(to, td, idx, tt))
            row[f"closest_{kind}_Mrays_s"] = round(n / ms / 1e3, 1)
            row[f"closest_{kind}_hit_frac"] = round(float((idx >= 0).float().mean()), 3)
        o, d, r = raygen.shadow_rays(tris, n)
        to, td, tr = (torch.from_numpy(a).to(dev) for a in (o, d, r))
        ms = timed(lambda: sc.any_hit_dev(to, td, tr, occ))
        row["any_Mrays_s"] = round(n / ms / 1e3, 1)
        cam, light = api.default_camera(S), api.default_light(True)
        acc = torch.zeros(S * S * 3, dtype=torch.float32, device=dev)
        p = api.default_params(S, S, rng_mode=abi.RNG_COUNTER, fog_sigma_t=0.1)
        sc.render_accumulate_dev(cam, light, p, 0, 1, acc)
        torch.cuda.synchronize()
        sc.render_accumulate_dev(cam, light, p, 1, args.spp, acc)
        torch.cuda.synchronize()
        ms = sc.stats().kernel_ms
        row["paths_Mpaths_s"] = round(S * S * args.spp / ms / 1e3, 1)
        p.flags = abi.FLAG_STATS if hasattr(abi, "FLAG_STATS") else 1
        sc.render_accumulate_dev(cam, light, p, 0, 2, acc)
        torch.cuda.synchronize()
        st = sc.stats()
        rays_per_path = (st.closest_rays + st.shadow_rays) / max(st.paths, 1)
        row["rays_per_path"] = round(rays_per_path, 2)
        row["paths_Mrays_s"] = round(row["paths_Mpaths_s"] * rays_per_path, 1)
        row["bvh_redo_per_Mray"] = round(1e6 * st.bvh_redo / max(st.closest_rays, 1), 2)
        row["bvh_overflow"] = int(st.bvh_overflow)
        n_rays_st = max(st.closest_rays + st.shadow_rays, 1)
        row["nodes_per_ray"] = round(st.bvh_nodes / n_rays_st, 1)
        row["leaves_per_ray"] = round(st.tri_tests / n_rays_st, 1)
        # algorithmic bytes of the traversal: 64 B per four-wide node fetched, 64 B per triangle record looked at
        bytes_per_ray = (st.bvh_nodes * 64 + st.tri_tests * 64) / n_rays_st
        row["paths_algorithmic_GB_s"] = round(row["paths_Mrays_s"] * 1e6 * bytes_per_ray / 1e9, 1)
        sc.close()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
