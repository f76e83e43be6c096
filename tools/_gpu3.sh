cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
F=tests/golden/resources
{
for i in 1 2; do for k in fixed dynamic png; do python tools/bench_variant.py $k 4096 0x10 2>&1 | tail -1; done; done
python tools/bench_file_stream.py $F/fs_angrymob.png 128 0x13 2>&1 | tail -1
python tools/bench_variant.py dynamic 512 0x13 1048576 2>&1 | tail -1
DEBIG_BENCH_FUSED=0 timeout -k 10 300 python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu.ids | tail -2
} 2>&1 | tee gpurun_out/r4f/search_skip.txt
