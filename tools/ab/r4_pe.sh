mkdir -p gpurun_out/r4_pe
timeout 1200 python -m pytest tests/test_parameter_estimation.py tests/test_quantile.py -x -q -m gpu > gpurun_out/r4_pe/tests.log 2>&1; tail -25 gpurun_out/r4_pe/tests.log
