mkdir -p gpurun_out/r07h
cd /tmp && export TMPDIR=/tmp
for lib in default tpb4w6nl tpb4w4; do
  if [ $lib = default ]; then unset LLPF_LIB; else export LLPF_LIB=$GRAFT_REPO_ROOT/lib_$lib.so; fi
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r07h/p_$lib -o p -- python $GRAFT_REPO_ROOT/bench.py --workload bank --steps 1 --warmup 0 --T 100 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
  db=$(find $GRAFT_REPO_ROOT/gpurun_out/r07h/p_$lib -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocprof_pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/r07h/pmc_$lib.txt $db > /dev/null
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r07h/p_$lib
  echo "== $lib"; grep -E "k_resprop" $GRAFT_REPO_ROOT/gpurun_out/r07h/pmc_$lib.txt | cut -c1-30,70-200
done
