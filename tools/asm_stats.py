#!/usr/bin/env python3
"""Developer tool: instruction mix of one kernel in the ISA listing `make asm` leaves
(computer-graphics_amd/csrc/mipt_api.s): vector / scalar instructions, scalar-register spill
traffic (v_readlane / v_writelane), scratch and LDS accesses.

    python tools/asm_stats.py <mangled-name substring> [listing.s]
"""
import re
import sys


def main():
    sub = sys.argv[1]
    path = sys.argv[2] if len(sys.argv) > 2 else "computer-graphics_amd/csrc/mipt_api.s"
    lines = open(path).read().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for n, i in enumerate(starts):
        name = lines[i].split(":")[0]
        if sub not in name:
            continue
        end = starts[n + 1] if n + 1 < len(starts) else len(lines)
        body = [l.strip() for l in lines[i:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        def count(pat):
            return sum(1 for l in body if re.match(pat, l))
        print(name)
        print("  instructions", len(body), " v_", count(r"v_"), " s_", count(r"s_"), " readlane", count(r"v_readlane"),
              " writelane", count(r"v_writelane"), " s_load", count(r"s_load"), " scratch", count(r"scratch_|buffer_(load|store)"),
              " ds_", count(r"ds_"), " global/flat", count(r"global_|flat_"), " s_cbranch", count(r"s_cbranch"),
              " v_mov", count(r"v_mov_b32"))


if __name__ == "__main__":
    main()
