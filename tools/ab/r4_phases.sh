set -u
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
LLPF_LIB=$PWD/lib_steptiming.so python tools/dbg/qt_phases.py > $OUT/qt_phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq_qt -o p -- python $GRAFT_REPO_ROOT/bench.py --workload quadtank --steps 1 --warmup 0 --T 100 --no-cpu-baseline > $OUT/pmc_sq_qt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_pmc_summary.py $OUT/pmc_sq_quadtank.txt $(find $OUT/pmc_sq_qt -name "*.db" | head -1)
rm -rf $OUT/pmc_sq_qt
cat $OUT/qt_phases.txt; cat $OUT/pmc_sq_quadtank.txt
