#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite output (ROCm 7.2 default) into the plain-text summaries kept
under profiles/: per-kernel call count / total / average duration (the `--stats` view) and,
for --pmc runs, per-kernel sums of every collected counter.

    python tools/rocpd_summary.py [--only SUBSTRING] <results.db> [more.db ...] > profiles/<name>.txt

--only keeps the kernels whose name contains SUBSTRING (e.g. "mipt::").

Kernel names are cut to 160 characters (rocPRIM's template names run to kilobytes).
"""
import sqlite3
import sys


def short(name, width=160):
    return name if len(name) <= width else name[:width - 3] + "..."


def main():
    args = sys.argv[1:]
    only = None
    if args and args[0] == "--only":
        only, args = args[1], args[2:]
    for path in args:
        db = sqlite3.connect(path)
        c = db.cursor()
        print(f"# {path}")
        rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            print("## kernel-trace --stats (durations in microseconds)")
            print(f"{'calls':>6} {'total_us':>14} {'avg_us':>14} {'pct':>7}  name")
            for name, calls, total, avg, pct in rows:
                if only and only not in name:
                    continue
                print(f"{calls:6d} {total:14.3f} {avg:14.3f} {pct:7.3f}  {short(name)}")
        try:
            pmc = list(c.execute(
                "select kernel_name, counter_name, count(distinct dispatch_id), sum(value), "
                "max(vgpr_count), max(sgpr_count), max(lds_block_size), max(grid_size), max(workgroup_size) "
                "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"))
        except sqlite3.OperationalError:
            pmc = []
        if pmc:
            print("## pmc (sum over dispatches)")
            last = None
            for k, cn, nd, v, vg, sg, lds, grid, wg in pmc:
                if only and only not in k:
                    continue
                if k != last:
                    print(f"kernel {short(k)}\n  dispatches={nd} vgpr={vg} sgpr={sg} lds={lds} grid={grid} wg={wg}")
                    last = k
                print(f"    {cn:28s} {v:.6g}")
        print()


if __name__ == "__main__":
    main()
