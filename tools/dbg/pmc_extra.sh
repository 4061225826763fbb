set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmcx; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in quadtank bank; do for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c -d $OUT/pmc_${c}_$w -o p -- python $ROOT/bench.py --workload $w --steps 1 --warmup 0 --T 100 --no-cpu-baseline > $OUT/pmc_${c}_$w.log 2>&1
done; done
cd $ROOT
for w in quadtank bank; do python tools/rocprof_pmc_summary.py $OUT/pmc_traffic_$w.txt $(find $OUT/pmc_FETCH_SIZE_$w -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE_$w -name "*.db" | head -1); rm -rf $OUT/pmc_FETCH_SIZE_$w $OUT/pmc_WRITE_SIZE_$w; done
grep -h "k_step\|k_resprop\|k_norm\|k_resample" $OUT/pmc_traffic_*.txt | cut -c1-175
