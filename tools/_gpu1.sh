cd $GRAFT_REPO_ROOT
timeout -k 10 1700 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_configs.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -8
