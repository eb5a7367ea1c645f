#!/bin/bash
# Collects, on the GPU box, the evidence behind one bench line:
#   bash tools/profile_round.sh <tag> [cornell|c4|c5]
# -> gpurun_out/prof_<tag>/: kernel-trace --stats of `bench.py --steps 3 --warmup 1 --no-cpu`, and one
# rocprofv3 --pmc pass each for the SQ counters, FETCH_SIZE, WRITE_SIZE and the L2 hit counters
# (a pass cannot hold them all), every pass a run of its own under its own time limit.
set -u
TAG=${1:?tag}
WL=${2:-cornell}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
# the stats pass times the bench command itself; the counter passes run the same launches alone
# (bench.py adds a counting launch and two fog-off launches of the same kernel template after its timed steps)
BENCH="bench.py --steps 3 --warmup 1 --no-cpu --workload $WL"
case $WL in
  cornell) PROBE="tools/pmc_probe.py --spp 1024 --fog 0.1 --flags 0 0 0 --no-batch" ;;
  *) PROBE="$BENCH" ;;
esac
run() {  # name, rocprofv3 options...
  local name=$1; shift
  local ARGS=$PROBE
  [ "$name" = stats ] && ARGS=$BENCH
  timeout -s KILL 300 rocprofv3 "$@" -d "$OUT/$name" -o "$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  find "$OUT/$name" -name "*_results.db" -exec cp {} "$OUT/${name}_results.db" \;
}
run stats --kernel-trace --stats && \
run sq --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM && \
run fetch --kernel-trace --pmc FETCH_SIZE && \
run write --kernel-trace --pmc WRITE_SIZE && \
run l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
python3 tools/rocpd_summary.py --only mipt:: "$OUT"/stats_results.db > "$OUT/kernel_stats.txt" && \
python3 tools/rocpd_summary.py --only mipt:: "$OUT"/sq_results.db > "$OUT/pmc_sq.txt" && \
python3 tools/rocpd_summary.py --only mipt:: "$OUT"/fetch_results.db "$OUT"/write_results.db "$OUT"/l2_results.db > "$OUT/pmc_hbm.txt" && \
grep '^{' "$OUT/stats.log" > "$OUT/bench_line.json"
case $WL in
  cornell) python3 tools/counters_json.py "$OUT" "k_paths<1, false, false, 0>" "$OUT/counters_k_paths.json" size=1024 spp=1024 ;;
  c4) python3 tools/counters_json.py "$OUT" "k_paths<1, false, false, 2>" "$OUT/counters_bvh_c4.json" fetch_scale=1 size=1024 spp=1024 ;;
  c5) python3 tools/counters_json.py "$OUT" "k_bvh_batch<false>" "$OUT/counters_bvh_c5.json" fetch_scale=1 algorithmic_bytes_per_ray=3963 ;;
esac > /dev/null
rm -rf "$OUT"/stats "$OUT"/sq "$OUT"/fetch "$OUT"/write "$OUT"/l2 "$OUT"/*.db
ls "$OUT"
