for rep in 1 2; do for abl in 0 1 4 5; do
  c2=$(LLPF_ABLATE=$abl LLPF_LIB=$PWD/lib_dev.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']))")
  echo "ablate=$abl rep$rep C2_us_per_step=$c2"
done; done
