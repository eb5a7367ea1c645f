set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3/pmc_ta
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- python3 bench.py --steps 2 --warmup 1 --no-cpu --workload c5 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }; find $OUT/$name -name "*_results.db" -exec cp {} $OUT/${name}_results.db \; ; }
run a TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE && \
run b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE && \
python3 tools/rocpd_summary.py --only k_bvh_batch $OUT/a_results.db $OUT/b_results.db > $OUT/summary.txt
rm -rf $OUT/a $OUT/b $OUT/*.db
grep -A 14 "^kernel" $OUT/summary.txt | cut -c1-120
