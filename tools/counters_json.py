#!/usr/bin/env python3
"""Per-launch hardware counters of one kernel, from the rocprofv3 --pmc passes that
tools/profile_round.sh leaves in a directory, as the small JSON file bench.py reads:

    python tools/counters_json.py gpurun_out/prof_<tag> <kernel substring> <out.json> [key=value ...]

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests
at 64 bytes). valu_issue_frac = wave-level VALU instructions x 2 issue cycles / (1024 SIMDs x
2.4e9 Hz x the kernel's duration in the un-instrumented pass). active_lanes =
SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64).
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def per_kernel(db_path, substr):
    db = sqlite3.connect(db_path)
    c = db.cursor()
    out = {}
    for k, cn, nd, v in c.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) "
                                  "from counters_collection group by kernel_name, counter_name"):
        if substr in k:
            out[cn] = (v, nd)
    return out, None


def main():
    d, substr, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    extra = dict(kv.split("=", 1) for kv in sys.argv[4:])
    sq, _ = per_kernel(f"{d}/sq_results.db", substr)
    fe, _ = per_kernel(f"{d}/fetch_results.db", substr)
    wr, _ = per_kernel(f"{d}/write_results.db", substr)
    l2, _ = per_kernel(f"{d}/l2_results.db", substr)
    # average duration of the kernel in the SQ-counter pass itself (microseconds)
    db = sqlite3.connect(f"{d}/sq_results.db")
    avg_us = None
    for name, avg in db.cursor().execute("select name, average from top_kernels"):
        if substr in name:
            avg_us = avg
    # the code these counters belong to: bench.py drops them when the sources have changed since
    res = {"kernel": substr, "source_hash": graft.kernel_source_hash(), "source": f"{d}: rocprofv3 --kernel-trace --pmc passes of `bench.py --steps 3 --warmup 1 --no-cpu` "
                                       "(tools/profile_round.sh); per launch = sum over dispatches / dispatches"}
    n = lambda t: t[0] / t[1]
    # fetch_scale=1 for kernels whose reads are 64-byte gathers (tree nodes, triangle records): their
    # requests are 64 bytes and the counter is exact (L2 misses x 64 B agrees with it); 2 for streams
    fetch_scale = float(extra.pop("fetch_scale", 2))
    if "FETCH_SIZE" in fe:
        res["fetch_bytes_per_launch"] = n(fe["FETCH_SIZE"]) * 1024 * fetch_scale
        res["fetch_scale"] = fetch_scale
    if "WRITE_SIZE" in wr:
        res["write_bytes_per_launch"] = n(wr["WRITE_SIZE"]) * 1024
    if "fetch_bytes_per_launch" in res and "write_bytes_per_launch" in res:
        res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
    if "TCC_HIT_sum" in l2 and "TCC_MISS_sum" in l2:
        h, m = l2["TCC_HIT_sum"][0], l2["TCC_MISS_sum"][0]
        res["l2_hit_rate"] = h / (h + m) if h + m else None
        res["l2_requests_per_launch"] = n(l2["TCC_REQ_sum"]) if "TCC_REQ_sum" in l2 else None
    if "SQ_ACTIVE_INST_VALU" in sq and "SQ_THREAD_CYCLES_VALU" in sq:
        res["active_lanes"] = sq["SQ_THREAD_CYCLES_VALU"][0] / sq["SQ_ACTIVE_INST_VALU"][0]
    if "SQ_INSTS_VALU" in sq and avg_us:
        res["valu_insts_per_launch"] = n(sq["SQ_INSTS_VALU"])
        res["kernel_us"] = avg_us
        res["valu_issue_frac"] = res["valu_insts_per_launch"] * 2 / (1024 * 2.4e9 * avg_us * 1e-6)
    for k, v in extra.items():
        try:
            res[k] = json.loads(v)
        except ValueError:
            res[k] = v
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
