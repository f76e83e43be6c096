cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
python tools/probe_fused_files.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/fused_files.txt
python tools/bench_fused_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/fused_vs_pair.txt
