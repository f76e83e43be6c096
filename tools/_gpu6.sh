cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
bash tools/refresh_profiles.sh 2>&1 | tee gpurun_out/r4i/refresh.log
cd $GRAFT_REPO_ROOT
echo "[tests]"
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r4i/gpu_tests.txt
