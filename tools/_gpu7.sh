cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "png or defilter or cfg or fused or dropin" 2>&1 | tail -3
{
for n in 32 16 64; do
DEBIG_BENCH_FUSED=0 timeout -k 10 200 python tools/bench_png.py cfg4 8192 $n 2>&1 | grep -v amdgpu.ids | tail -5
done
} 2>&1 | tee gpurun_out/r4k/cfg4_mask_select.txt
