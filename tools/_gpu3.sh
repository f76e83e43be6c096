cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
{ echo "#### with the image-rows hint (product)"; python tools/bench_decode_png_call.py 2>&1 | grep -v amdgpu.ids
echo "#### without (DEBIG_CHUNKED_ROWS_MIN_IN_BYTES = never)"; DEBIG_LIB=$PWD/debigulator_amd/lib/libdebigulator_hip_ab_norowshint.so python tools/bench_decode_png_call.py 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r4f/decode_png_call.txt
timeout -k 10 900 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -3
