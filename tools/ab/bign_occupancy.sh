mkdir -p gpurun_out/r06l
for rep in 1 2; do
for lib in default w5 w6 w8; do
for sch in split merged; do
  if [ $lib = default ]; then unset LLPF_LIB; else export LLPF_LIB=$PWD/lib_$lib.so; fi
  echo "== $lib $sch rep $rep"
  LLPF_SCHEDULE=$sch python tools/bench_n.py --sizes 4000000,16000000 --passes 2 | grep -E '"particles"|us_per_timestep"|roofline_frac|loglik'
done; done; done
