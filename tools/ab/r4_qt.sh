set -u
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
timeout 120 python tests/../tools/dbg/qt_small_check.py > $OUT/small_check.txt 2>&1; echo "small rc=$?" >> $OUT/small_check.txt
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?" >> $OUT/gpu_suite.log
rm -f $OUT/qt_ab.txt
for v in "LLPF_SOURCE_FX=1" "LLPF_SOURCE_FX=0"; do for rep in 1 2; do
env $v timeout 120 python bench.py --workload quadtank --steps 2 --warmup 1 --T 500 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep', '%.3e' % d['value'], d['ms_per_step'], {k.split('(')[0]: (round(v,2) if v else v) for k,v in d['kernel_us'].items()})" >> $OUT/qt_ab.txt 2>&1
done; done
LLPF_LIB=$PWD/lib_steptiming.so timeout 120 python tools/dbg/qt_phases.py > $OUT/qt_phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $OUT/kt_qt -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload quadtank --steps 2 --no-cpu-baseline > $OUT/kt_qt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $OUT/kt_qt -name "*.db" | head -1) > $OUT/kernel_stats_qt.txt
rm -rf $OUT/kt_qt
cat $OUT/small_check.txt; tail -3 $OUT/gpu_suite.log; cat $OUT/qt_ab.txt; head -8 $OUT/kernel_stats_qt.txt; cat $OUT/qt_phases.txt
