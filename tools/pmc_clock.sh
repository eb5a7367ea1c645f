# The clock the chip actually runs at under each workload: GRBM_GUI_ACTIVE (cycles the graphics engine
# was active; the counter is summed over the 8 XCDs) / 8 / the kernels' run time, one rocprofv3 --pmc pass each.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3/pmc_clock
mkdir -p $OUT
for wl in cornell c4 c5; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/$wl -o $wl -- python3 bench.py --steps 3 --warmup 1 --no-cpu --workload $wl > $OUT/$wl.log 2>&1 || { echo "pass $wl failed"; tail -3 $OUT/$wl.log; exit 1; }
  find $OUT/$wl -name "*_results.db" -exec cp {} $OUT/${wl}_results.db \;
done
python3 tools/rocpd_summary.py --only mipt:: $OUT/cornell_results.db $OUT/c4_results.db $OUT/c5_results.db > $OUT/summary.txt
rm -rf $OUT/cornell $OUT/c4 $OUT/c5 $OUT/*.db
grep -E "^kernel.*(k_paths|k_bvh_batch)|GRBM|calls|k_paths|k_bvh_batch" $OUT/summary.txt | cut -c1-140
