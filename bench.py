#!/usr/bin/env python3
"""Throughput of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of samples: the ranks together render
`global spp per step` samples of every pixel of the GlassCube Cornell box at 1024 x 1024
(glass block, mirror, fog) into device-resident accumulation buffers.

  N = 1   BASELINE configs[1]: 1024 spp per step on the one GPU.
  N > 1   BASELINE configs[2] (default, --scaling strong): 4096 spp per step, the sample range split
          into N contiguous parts, one per rank, and the per-rank images sum-reduced to rank 0 over
          RCCL; --scaling weak keeps 1024 spp per rank per step instead.

The image lives in HBM for the whole timed region; nothing crosses PCIe. One process per GPU:
started by torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or by this script
itself -- `python bench.py --gpus N` without those variables starts N rank processes before it
touches a GPU and waits for them. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402

SIZE = 1024
SPP_CONFIG1 = 1024   # BASELINE configs[1]
SPP_CONFIG2 = 4096   # BASELINE configs[2], split over the ranks
FOG_SIGMA_T = 0.1    # builder's choice for the "+ fog" of configs[1]; the reference has no fog code
FOG_ALBEDO = 0.8
C5_ALGORITHMIC_BYTES_PER_RAY = 3963   # 55.7 nodes + 5.6 triangle records of 64 B + 40 B per ray (stats kernel, configs[4])
# algorithmic work of the reference's triangle test: 26 flop for the determinant-ratio gate,
# 79 more when the gate passes (inverse, mat*vec, barycentric sum, position)
FLOP_GATE, FLOP_INVERSE = 26, 79
PEAK_FP32_VECTOR = 157.3   # TFLOP/s, MI355X vector peak, FMA counted as 2
PEAK_FP32_NONFUSED = 78.6  # the same issue rate with one flop per lane-instruction (no FMA allowed here)


# ---------------------------------------------------------------------------------------------
# launcher: N rank processes on one node

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, script=None, extra_env=None):
    """Start `n` fresh processes of `script` (default: this file) with argv, one per rank, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set; nothing in THIS process touches
    a GPU. Their output passes through (rank 0 prints the JSON line). Returns 0 if all ranks
    succeeded, else the first non-zero exit code; a failed rank ends the others."""
    script = script or os.path.abspath(__file__)
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:      # the others would wait for the dead rank in a collective
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def rank_env():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def timed_steps(step, steps, warmup, barrier, max_over_ranks):
    """The timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by
    barrier() (a collective barrier + device synchronisation) on both sides; returns the elapsed
    seconds, maximum over the ranks."""
    for w in range(warmup):
        step(w)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    barrier()
    return max_over_ranks(time.perf_counter() - t0)


# ---------------------------------------------------------------------------------------------

def cpu_baseline(abi, tris, light, params):
    """The strict CPU oracle on the host cores, on bounded samples of the workloads: the bench
    scene at 256 x 256, and BASELINE configs[0] (Cornell test model, 256 x 256, 16 spp, diffuse only,
    no fog) in full. All threads the process may use, one thread, and the -Ofast build the
    reference's Makefile would produce (CW/raytracer/Makefile:12-14)."""
    import oracle_lib as O
    visible = len(os.sched_getaffinity(0))
    # a GPU box hands each job a share of its cores and says so in OMP_NUM_THREADS
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = min(visible, share) if share > 0 else visible
    crop = 256
    p = abi.MiptParams.from_buffer_copy(bytes(params))
    p.width = p.height = crop
    p.flags = 0
    c = O.default_camera(crop)

    def rate(tr, cam, lt, par, spp, thr, fast):
        t0 = time.perf_counter()
        O.render_accumulate(tr, None, cam, lt, par, 0, spp, threads=thr, fast=fast)
        return par.width * par.height * spp / (time.perf_counter() - t0)

    out = {"strict_mt": rate(tris, c, light, p, 256, threads, False),
           "strict_1t": rate(tris, c, light, p, 16, 1, False),
           "ofast_mt": rate(tris, c, light, p, 256, threads, True),
           "ofast_1t": rate(tris, c, light, p, 16, 1, True)}
    # configs[0]: the reference's own CPU-runnable case, whole
    t0 = O.load_test_model(0)
    p0 = O.default_params(crop, crop, rng_mode=abi.RNG_MINSTD_PATH, seed=0)
    l0 = O.default_light(False)
    c0 = {"workload": "BASELINE configs[0]: Cornell test model (30 triangles) 256x256, 16 spp, diffuse only, no fog, whole",
          "strict_all_threads_paths_per_s": rate(t0, c, l0, p0, 16, threads, False),
          "strict_one_thread_paths_per_s": rate(t0, c, l0, p0, 16, 1, False),
          "ofast_all_threads_paths_per_s": rate(t0, c, l0, p0, 16, threads, True),
          "ofast_one_thread_paths_per_s": rate(t0, c, l0, p0, 16, 1, True)}
    return {"value": out["strict_mt"], "unit": "paths/s", "cores": threads, "kind": "port",
            "host_threads_visible": visible,
            "sample": f"GlassCube {crop}x{crop}, 256 spp, fog sigma_t={FOG_SIGMA_T}, strict-IEEE C oracle "
                      f"(oracle/pt_oracle.c, gcc -O2 -ffp-contract=off -fopenmp), {threads} threads "
                      f"(per-path random streams, so the threads do not share the reference's one racy engine)",
            "one_thread_paths_per_s": out["strict_1t"],
            "ofast_build_paths_per_s": out["ofast_mt"],
            "ofast_build_one_thread_paths_per_s": out["ofast_1t"],
            "config0": c0}


def load_counters(name):
    """Per-launch hardware counters of a kernel from the committed rocprofv3 --pmc passes
    (tools/profile_round.sh -> tools/counters_json.py). The file names the hash of the sources it was
    measured on; counters of other code are not reported: returns (counters or None, stale)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            hw = json.load(f)
    except (OSError, ValueError):
        return None, False
    if hw.get("source_hash") != graft.kernel_source_hash():
        return None, True
    return hw, False


_COLLECTIVES = None   # backend of the process group, if one was made


def init_ranks(rehearse=False, force_dist=False):
    """rehearse: every rank on GPU 0 and gloo for the collectives (RCCL refuses two ranks on one device):
    the N-rank code path on a one-GPU machine, for checking it -- the timings mean nothing.
    force_dist: make the process group even for a world of one rank, so that the RCCL backend's
    initialisation, barrier, reduce and all-reduce run on a one-GPU machine exactly as with N ranks."""
    global _COLLECTIVES
    import torch
    import torch.distributed as dist
    world, rank, local_rank = rank_env()
    if rehearse:
        local_rank = 0
    grouped = world > 1 or force_dist
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        _COLLECTIVES = "gloo" if rehearse else "nccl"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if grouped:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    return world, rank, dev, barrier, max_over_ranks


def finish_ranks(world):
    if _COLLECTIVES:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def tree_workload(args):
    """configs[3]: whole paths through the room + 102 400-triangle mesh, samples split over ranks + one
    sum-reduce (like the Cornell workload). configs[4]: a 3840 x 2160 batch of incoherent closest-hit
    rays against 10^6 random triangles, rays sharded over ranks, no collective."""
    import torch
    import bigscenes
    import raygen

    world, rank, dev, barrier, max_over_ranks = init_ranks(args.rehearse, args.force_dist)
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    from computer_graphics_amd import parallel

    if args.workload == "c4":
        tris = bigscenes.config4(api, abi.TRIANGLE_DTYPE)
        strong = args.scaling == "strong"
        global_spp = args.spp if args.spp else (SPP_CONFIG1 if strong else SPP_CONFIG1 * world)
        cam, light = api.default_camera(SIZE), api.default_light(glass=True)
        params = api.default_params(SIZE, SIZE, rng_mode=abi.RNG_COUNTER, seed=0, fog_sigma_t=FOG_SIGMA_T,
                                    fog_albedo=FOG_ALBEDO)
        scene = api.Scene(tris, None, device=dev.index)
        buf = torch.zeros(SIZE * SIZE * 3, dtype=torch.float32, device=dev)
        total = torch.zeros_like(buf) if rank == 0 or args.exact else None

        def render(begin, count, image):
            scene.render_accumulate_dev(cam, light, params, begin, count, image)

        def step(k):
            if args.exact:   # the image relayed through the ranks: the one-GPU image bit for bit
                parallel.render_chained(lambda b, c: scene.render_items_dev(cam, light, params, b, c),
                                        scene.fold_items_dev, k * global_spp, global_spp, rank, world, total,
                                        via_host=args.rehearse)
            else:
                parallel.render_distributed(render, k * global_spp, global_spp, rank, world, buf, total,
                                            via_host=args.rehearse, always=args.force_dist)
        units_per_step, unit = SIZE * SIZE * global_spp, "paths/s"
        label = (f"empty Cornell room + 102 400-triangle glass mesh (LBVH), {SIZE}x{SIZE}, {global_spp} spp per step"
                 f"{' split over the GPUs' if world > 1 else ''}, fog on; stand-in for BASELINE configs[3]")
    else:
        tris = bigscenes.config5(abi.TRIANGLE_DTYPE)
        n_all = 3840 * 2160
        lo, cnt = parallel.rank_sample_range(0, n_all, rank, world)   # rays sharded like samples: contiguous runs
        o, d = raygen.interior_rays(n_all, seed=42)
        to, td = torch.from_numpy(o[lo:lo + cnt]).to(dev), torch.from_numpy(d[lo:lo + cnt]).to(dev)
        idx = torch.zeros(cnt, dtype=torch.int32, device=dev)
        tt = torch.zeros(cnt, dtype=torch.float32, device=dev)
        scene = api.Scene(tris, None, device=dev.index)

        def step(k):
            scene.closest_hit_dev(to, td, idx, tt)
        units_per_step, unit = n_all, "rays/s"
        label = ("10^6 random triangles (LBVH), 3840x2160 incoherent closest-hit rays per step"
                 f"{', rays sharded over the GPUs' if world > 1 else ''}; BASELINE configs[4]")

    dt = timed_steps(step, args.steps, args.warmup, barrier, max_over_ranks)
    if rank == 0:
        value = units_per_step * args.steps / dt
        line = {"metric": "paths/sec (+ Mrays/sec), tree scenes, MI355X" if args.workload == "c4" else "Mrays/sec, LBVH closest hit, MI355X",
                "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                "scaling": args.scaling if args.workload == "c4" else "strong", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic", "config": {"workload": label, "parallelism": f"x{world}"}}
        if args.rehearse:
            line["rehearsal"] = "all ranks on one GPU, gloo through the host: not a measurement"
        hw, stale = load_counters(f"counters_bvh_{args.workload}.json")
        if args.workload == "c4":
            sp = abi.MiptParams.from_buffer_copy(bytes(params))
            sp.flags = abi.FLAG_STATS
            scene.render_accumulate_dev(cam, light, sp, 0, 4, buf)
            torch.cuda.synchronize()
            st = scene.stats()
            rays = st.closest_rays + st.shadow_rays
            bytes_per_unit = (st.bvh_nodes * 64 + st.tri_tests * 64 + rays * 40) / st.paths
            line["mrays_per_s"] = value * rays / st.paths / 1e6
            line["nodes_per_ray"] = st.bvh_nodes / rays
            line["leaves_per_ray"] = st.tri_tests / rays
        else:
            bytes_per_unit = C5_ALGORITHMIC_BYTES_PER_RAY
        hw = hw or {}
        # what the memory system actually moved: HBM bytes per launch from the counter passes over the
        # launch's duration in those passes (null when the counters are of other code)
        traffic = hw.get("hbm_bytes_per_launch")
        measured = traffic / (hw["kernel_us"] * 1e-6) / 1e9 if traffic and hw.get("kernel_us") else None
        lanes, issue = hw.get("active_lanes"), hw.get("valu_issue_frac")
        line["roofline"] = {
            "bound": "latency", "achieved": measured, "peak": 8000.0, "unit": "GB/s",
            "frac": measured / 8000.0 if measured else None, "traffic": traffic,
            "algorithmic_gbps": value / world * bytes_per_unit / 1e9,
            "l2_hit_rate": hw.get("l2_hit_rate"), "active_lanes": lanes, "valu_issue_frac": issue,
            "lane_slot_frac": issue * lanes / 64.0 if issue and lanes else None,
            "counters_stale": stale,
            "note": "The walk's gathers are dependent (a node's address comes out of the previous node): the kernel "
                    "is bound by L2 / HBM LATENCY at the occupancy its registers allow, not by bandwidth. achieved "
                    "= HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, profiles/counters_bvh_*.json) "
                    "/ the launch's duration in those passes; frac = achieved / the 8 TB/s peak. algorithmic_gbps "
                    "= 64 B per four-wide node fetched + 64 B per triangle record looked at + 40 B per ray, times "
                    "the measured rate (what a cache-less walk would move; most of it hits L2). lane_slot_frac = "
                    "valu_issue_frac x active_lanes / 64: the share of the machine's lane slots doing work. "
                    "counters_stale: the committed counters are of other sources than this build (fields null)"}
        print(json.dumps(line), flush=True)
    scene.close()
    finish_ranks(world)


def cornell_workload(args):
    import torch

    world, rank, dev, barrier, max_over_ranks = init_ranks(args.rehearse, args.force_dist)
    pkg = graft.load_package()
    abi, api = pkg.abi, pkg.api
    from computer_graphics_amd import parallel

    tris = api.load_test_model(abi.SCENE_GLASS_CUBE)
    cam = api.default_camera(SIZE)
    light = api.default_light(glass=True)
    params = api.default_params(SIZE, SIZE, rng_mode=abi.RNG_COUNTER, seed=0,
                                fog_sigma_t=FOG_SIGMA_T, fog_albedo=FOG_ALBEDO)
    scene = api.Scene(tris, None, device=dev.index)
    strong = args.scaling == "strong"
    if args.spp:
        global_spp = args.spp if strong else args.spp * world
    elif world == 1:
        global_spp = SPP_CONFIG1
    else:
        global_spp = SPP_CONFIG2 if strong else SPP_CONFIG1 * world
    n_img = SIZE * SIZE * 3
    step_buf = torch.zeros(n_img, dtype=torch.float32, device=dev)
    # --exact: the running image exists on every rank (it travels through them); else rank 0 sums
    total = torch.zeros(n_img, dtype=torch.float32, device=dev) if rank == 0 or args.exact else None
    split = parallel.rank_item_range if args.exact else parallel.rank_sample_range
    my_spp = split(0, global_spp, rank, world)[1]

    def render(begin, count, image):
        scene.render_accumulate_dev(cam, light, params, begin, count, image)

    kernel_ms = []

    def one_step(k):
        if args.exact:
            # every rank traces its whole work items at once, then the image makes the round of the ranks
            # and each folds its items in: the bits ONE GPU would leave, whatever the number of ranks
            parallel.render_chained(lambda b, c: scene.render_items_dev(cam, light, params, b, c), scene.fold_items_dev,
                                    k * global_spp, global_spp, rank, world, total, via_host=args.rehearse)
        else:
            # the step's global sample range is split into `world` contiguous sub-ranges; one sum-reduce
            parallel.render_distributed(render, k * global_spp, global_spp, rank, world, step_buf, total,
                                        via_host=args.rehearse, always=args.force_dist)
        if k >= args.warmup and rank == 0:
            kernel_ms.append(scene.stats().kernel_ms)  # HIP events around the path kernel on its launch stream

    dt = timed_steps(one_step, args.steps, args.warmup, barrier, max_over_ranks)
    paths_per_step = SIZE * SIZE * global_spp
    value = paths_per_step * args.steps / dt

    if rank == 0:
        # exact work counters of one launch of rank 0's share (stats variant of the kernel, untimed)
        sp = abi.MiptParams.from_buffer_copy(bytes(params))
        sp.flags = abi.FLAG_STATS
        step_buf.zero_()
        scene.render_accumulate_dev(cam, light, sp, 0, my_spp, step_buf)
        torch.cuda.synchronize()
        st = scene.stats()
        flops = FLOP_GATE * st.tri_tests + FLOP_INVERSE * st.gate_passes
        k_ms = float(np.mean(kernel_ms))
        achieved = flops / (k_ms * 1e-3) / 1e12
        rays_per_path = (st.closest_rays + st.shadow_rays) / st.paths
        # the same workload without fog (the part of the path the reference pins)
        nf = abi.MiptParams.from_buffer_copy(bytes(params))
        nf.fog_sigma_t = 0.0
        step_buf.zero_()
        scene.render_accumulate_dev(cam, light, nf, 0, my_spp, step_buf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        scene.render_accumulate_dev(cam, light, nf, my_spp, my_spp, step_buf)
        torch.cuda.synchronize()
        fog_off = SIZE * SIZE * my_spp / (time.perf_counter() - t1)

        # hardware counters per launch of this workload, from the committed rocprofv3 --pmc passes
        hw, stale = load_counters("counters_k_paths.json")
        hw = hw or {}
        scale = my_spp / hw["spp"] if hw.get("size") == SIZE and hw.get("spp") else None
        traffic = (hw["fetch_bytes_per_launch"] + hw["write_bytes_per_launch"]) * scale if scale else None

        cfg = "BASELINE configs[1]" if world == 1 else "BASELINE configs[2]" if strong else "weak scaling of configs[1]"
        line = {
            "metric": "paths/sec (+ Mrays/sec), Cornell box 1024^2, MI355X",
            "value": value, "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"GlassCube Cornell box (22 triangles: red room, glass block, mirror) "
                                   f"{SIZE}x{SIZE}, {global_spp} spp per step, max depth 10, fog sigma_t="
                                   f"{FOG_SIGMA_T} albedo={FOG_ALBEDO}, counter-based RNG; {cfg}"
                                   + ((f": samples split over {world} GPUs ({my_spp} on rank 0) in whole work items; the image "
                                       f"makes the round of the GPUs and each folds its items in -- bit-identical to one GPU")
                                      if world > 1 and args.exact else
                                      (f": samples split over {world} GPUs ({my_spp} each) + one RCCL sum-reduce "
                                       f"of the fp32 image per step") if world > 1 else ""),
                       "spp_per_gpu_per_step": my_spp, "global_spp_per_step": global_spp,
                       "parallelism": f"spp-split x{world}"},
            "mrays_per_s": value * rays_per_path / 1e6,
            "rays_per_path": rays_per_path,
            "paths_per_s_fog_off_1gpu": fog_off,
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VECTOR, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VECTOR, "traffic": traffic,
                         "peak_nonfused": PEAK_FP32_NONFUSED, "frac_nonfused": achieved / PEAK_FP32_NONFUSED,
                         "valu_issue_frac": hw.get("valu_issue_frac"), "active_lanes": hw.get("active_lanes"),
                         # executed work: share of the machine's lane-operations per second that the kernel's own
                         # VALU instructions fill (issue slots used x lanes active), no reference flops involved
                         "executed_lane_op_frac": (hw["valu_issue_frac"] * hw["active_lanes"] / 64.0
                                                   if hw.get("valu_issue_frac") and hw.get("active_lanes") else None),
                         "counters_stale": stale,
                         "kernel": "k_paths", "kernel_ms": k_ms, "flops_per_launch": flops,
                         "tri_tests": st.tri_tests, "gate_passes": st.gate_passes,
                         "note": "achieved = algorithmic flops of the REFERENCE's scan for this launch (26 per "
                                 "triangle test + 79 per gate pass, counted by the stats variant of the kernel, which "
                                 "tests every triangle like intersections.cpp does) / the shipped kernel's time; the "
                                 "shipped kernel reaches the same results with fewer executed flops (candidate "
                                 "filter), so this is a rate of reference work, not an instruction count. FMA "
                                 "contraction is forbidden by the bit-exact contract: 78.6 TFLOP/s is the issue-rate "
                                 "roof. valu_issue_frac = wave-level VALU instructions x 2 cycles / (SIMDs x cycles), "
                                 "active_lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64), both from the "
                                 "committed rocprofv3 --pmc passes (profiles/counters_k_paths.json), like traffic = "
                                 "HBM bytes per launch (12 bytes written per 16-sample work item; the 22 triangles "
                                 "stay in the scalar cache and LDS). counters_stale: that file names the hash of the "
                                 "sources it was measured on; when this build's differ the counter fields are null"},
        }
        if args.rehearse:
            line["rehearsal"] = "all ranks on one GPU, gloo through the host: not a measurement"
        if _COLLECTIVES:
            line["collectives"] = _COLLECTIVES
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(abi, tris, light, params)
        print(json.dumps(line), flush=True)
        if args.save_total:
            np.save(args.save_total, total.cpu().numpy())

    scene.close()
    finish_ranks(world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = BASELINE configs[2], a fixed 4096 spp per step split over the GPUs; "
                         "weak = 1024 spp per GPU per step")
    ap.add_argument("--spp", type=int, default=0,
                    help="override: samples per pixel per step (strong: in total; weak: per GPU)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a one-GPU machine: all ranks on GPU 0, gloo instead of RCCL, images reduced through "
                         "the host. Checks the N-rank code path; the line says so and its timings mean nothing")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: go through torch.distributed all the same (process group over RCCL, barrier, reduce of "
                         "the image, MAX all-reduce of the timing) -- the N-rank plumbing on a one-GPU machine")
    ap.add_argument("--exact", action="store_true",
                    help="cornell and c4, N > 1: instead of the sum-reduce, pass the image from rank to rank, each folding "
                         "its work items in (parallel.render_chained): the N-GPU image is the 1-GPU image bit for bit")
    ap.add_argument("--save-total", default=None, metavar="FILE.npy",
                    help="cornell: rank 0 saves the image summed over all warm-up and timed steps (for checks)")
    ap.add_argument("--workload", choices=["cornell", "c4", "c5"], default="cornell",
                    help="cornell = BASELINE configs[1]/[2] (the bench line the driver records); c4 / c5 = the "
                         "tree-scene stand-ins for configs[3] / configs[4] (extra lines, same JSON shape)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher, before anything touches a GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = rank_env()[0]
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running {world} rank(s)", file=sys.stderr)
    if args.workload != "cornell":
        return tree_workload(args)
    return cornell_workload(args)


if __name__ == "__main__":
    main()
