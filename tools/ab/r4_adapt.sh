mkdir -p gpurun_out/r4_adapt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forms or chooses or quadtank" > gpurun_out/r4_adapt/new_tests.log 2>&1; tail -5 gpurun_out/r4_adapt/new_tests.log
timeout 900 python tools/dbg/qt_regimes.py > gpurun_out/r4_adapt/regimes.txt 2>&1; cat gpurun_out/r4_adapt/regimes.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4_adapt/gpu_suite.log 2>&1; tail -5 gpurun_out/r4_adapt/gpu_suite.log
