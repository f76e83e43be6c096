cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "png or defilter or cfg or fused" 2>&1 | tail -3 | tee gpurun_out/r4j/gpu_tests_px.txt
{
for px in 1 0; do
echo "== DEBIG_DEFILTER_PXSKEW=$px"
DEBIG_DEFILTER_PXSKEW=$px DEBIG_BENCH_FUSED=0 timeout -k 10 200 python tools/bench_png.py cfg4 8192 32 2>&1 | grep -v amdgpu.ids | tail -5
done
for n in 8 16; do
echo "== PX, $n images"
DEBIG_BENCH_FUSED=0 timeout -k 10 200 python tools/bench_png.py cfg4 8192 $n 2>&1 | grep -v amdgpu.ids | tail -5
done
} 2>&1 | tee gpurun_out/r4j/defilter_px.txt
