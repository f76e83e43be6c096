cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
L=debigulator_amd/lib
F=tests/golden/resources
{
for v in base; do
  echo "#### $v"
  for f in fs_angrymob.png gimp_test.png phoebus.png purpleback.png; do python tools/bench_file_stream.py $F/$f 128 0x13 2>&1 | tail -1; done
  python tools/bench_file_stream.py $F/fs_angrymob.png 365 0x13 2>&1 | tail -1
  for k in fixed dynamic png; do python tools/bench_variant.py $k 4096 0x10 2>&1 | tail -1; done
  python tools/bench_variant.py dynamic 512 0x13 1048576 2>&1 | tail -1
  python tools/bench_variant.py dynamic 256 0x13 1048576 2>&1 | tail -1
  python tools/bench_variant.py png 768 0x13 1048576 2>&1 | tail -1
  DEBIG_BENCH_FUSED=1 timeout -k 10 300 python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu.ids | tail -4
  DEBIG_BENCH_FUSED=0 timeout -k 10 300 python tools/bench_png.py cfg4 8192 8 2>&1 | grep -v amdgpu.ids | tail -3
done
PROF_WIDTH=0x12 timeout -k 10 200 python tools/prof_split_png_file.py $F/fs_angrymob.png 2>&1 | grep -v amdgpu.ids | tail -12
} 2>&1 | tee gpurun_out/r4e/medium_copy.txt
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4e/bench_line.json 2> gpurun_out/r4e/bench_err.txt; tail -c 3000 gpurun_out/r4e/bench_line.json
