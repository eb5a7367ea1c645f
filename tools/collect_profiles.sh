#!/bin/bash
# After `bash tools/profile_round.sh <tag>_<workload> <workload>` ran on the GPU box for cornell, c4 and c5:
#   bash tools/collect_profiles.sh <tag> <round>
# copies the summaries and counter files from gpurun_out/prof_<tag>_* into profiles/ (named <round>_*).
# The bench lines are taken afterwards (bench.py reads the counter files it finds in profiles/).
set -eu
TAG=${1:?tag}
ROUND=${2:?round prefix, e.g. r3}
for w in cornell c4 c5; do
  d=gpurun_out/prof_${TAG}_$w
  for f in kernel_stats pmc_sq pmc_hbm; do cp "$d/$f.txt" "profiles/${ROUND}_${w}_$f.txt"; done
done
cp gpurun_out/prof_${TAG}_cornell/counters_k_paths.json gpurun_out/prof_${TAG}_c4/counters_bvh_c4.json \
   gpurun_out/prof_${TAG}_c5/counters_bvh_c5.json profiles/
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import __graft_entry__ as g
h = g.kernel_source_hash()
for f in ("counters_k_paths", "counters_bvh_c4", "counters_bvh_c5"):
    d = json.load(open(f"profiles/{f}.json"))
    print(f, "hash matches the tree" if d["source_hash"] == h else "STALE", d.get("kernel_us"))
PY
