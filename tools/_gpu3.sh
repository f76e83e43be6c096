cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r4f/gpu_tests.txt
