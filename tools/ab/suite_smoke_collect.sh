mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final/gpu_suite.log 2>&1; tail -4 gpurun_out/final/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1700 bash tools/collect_profiles.sh ${1:-final} > gpurun_out/final/collect.log 2>&1; tail -3 gpurun_out/final/collect.log
