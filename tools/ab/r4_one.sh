timeout 600 python -m pytest tests/test_user_models.py tests/test_independent_oracle.py tests/test_reference_fixtures.py -m gpu -x -q 2>&1 | tail -8
