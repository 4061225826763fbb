timeout 600 python -m pytest tests/test_user_models.py tests/test_gpu_parity.py -m gpu -x -q -k "bound or weighted_cov or noise" 2>&1 | tail -15
