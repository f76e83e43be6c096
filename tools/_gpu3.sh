cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_fused_png.py -x -q -m gpu 2>&1 | tail -4
