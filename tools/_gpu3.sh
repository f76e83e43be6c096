cd $GRAFT_REPO_ROOT
timeout -k 10 1000 bash tools/refresh_profiles.sh 2>&1 | tail -8
mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4g/gpu_tests_final.txt
