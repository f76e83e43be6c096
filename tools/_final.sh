mkdir -p gpurun_out/r3z gpurun_out/refresh
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3z/gpu_tests.txt 2>&1; tail -2 gpurun_out/r3z/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
{
python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu
python tools/bench_png.py cfg4 8192 32 2>&1 | grep -v amdgpu
python tools/bench_png.py cfg4 8192 256 2>&1 | grep -v amdgpu
python tools/bench_gz.py 2>&1 | grep -v amdgpu | tail -3
python tools/bench_chunked.py dynamic 256 1 2>&1 | grep -v amdgpu | tail -3
python tools/bench_variant.py dynamic 4096 0x10 2>&1 | grep -v amdgpu
python tools/bench_variant.py png 4096 0x10 2>&1 | grep -v amdgpu
python tools/bench_host_api.py 2>&1 | grep -v amdgpu | tail -4
} > gpurun_out/r3z/final.txt 2>&1
cat gpurun_out/r3z/final.txt
bash tools/refresh_profiles.sh 2>&1 | tee gpurun_out/refresh/log.txt | cut -c1-200
