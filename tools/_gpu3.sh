cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout -k 10 600 python -m pytest tests/test_fused_png.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r4f/hybrid_tests.txt
grep -q passed gpurun_out/r4f/hybrid_tests.txt && ! grep -q failed gpurun_out/r4f/hybrid_tests.txt || exit 1
{ DEBIG_BENCH_FUSED=1 timeout -k 10 300 python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu.ids | tail -6
python tools/probe_hybrid_parts.py 2>&1 | grep -v amdgpu.ids
python tools/bench_fused_probe.py 2>&1 | grep -v amdgpu.ids; } | tee gpurun_out/r4f/hybrid_cfg3.txt
