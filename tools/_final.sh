mkdir -p gpurun_out/r3z
python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu
python tools/bench_png.py cfg4 8192 32 2>&1 | grep -v amdgpu
python tools/bench_png.py cfg4 8192 256 2>&1 | grep -v amdgpu
python tools/bench_gz.py 2>&1 | grep -v amdgpu | tail -3
python tools/bench_chunked.py dynamic 256 1 2>&1 | grep -v amdgpu | tail -3
python tools/bench_variant.py dynamic 4096 0x10 2>&1 | grep -v amdgpu
python tools/bench_variant.py png 4096 0x10 2>&1 | grep -v amdgpu
python tools/bench_host_api.py 2>&1 | grep -v amdgpu | tail -4
