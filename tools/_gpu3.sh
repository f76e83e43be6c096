cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
{ python tools/probe_fused_files.py 2>&1 | grep -v amdgpu.ids | grep -v "structuredart\|fs_b\|fs_c"
DEBIG_BENCH_FUSED=1 timeout -k 10 300 python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu.ids | tail -3
python tools/bench_fused_probe.py 2>&1 | grep -v amdgpu.ids | grep "x    64\|x   256\|x  1024\|small"; } | tee gpurun_out/r4g/rows_weight.txt
