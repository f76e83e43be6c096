cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
F=tests/golden/resources
{
for big in 0 default; do
  if [ $big = 0 ]; then export DEBIG_PIPE_BIG_TILE_STREAMS=0; else unset DEBIG_PIPE_BIG_TILE_STREAMS; fi
  echo "#### DEBIG_PIPE_BIG_TILE_STREAMS=$big"
  for f in fs_angrymob.png gimp_test.png; do python tools/bench_file_stream.py $F/$f 128 0x13 2>&1 | tail -1; done
  for n in 256 512 768 1024 1280 2048; do python tools/bench_variant.py dynamic $n 0x13 1048576 2>&1 | tail -1; done
  python tools/bench_variant.py png 768 0x13 1048576 2>&1 | tail -1
  python tools/bench_variant.py fixed 1024 0x13 65536 2>&1 | tail -1
  python tools/bench_variant.py png 1024 0x13 65536 2>&1 | tail -1
done
unset DEBIG_PIPE_BIG_TILE_STREAMS
python tools/bench_variant.py dynamic 256 8 1048576 2>&1 | tail -1
DEBIG_BENCH_FUSED=1 timeout -k 10 300 python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu.ids | tail -2
python tools/probe_hybrid_parts.py 2>&1 | grep -v amdgpu.ids
python tools/bench_fused_probe.py 2>&1 | grep -v amdgpu.ids | grep "x    64\|x   128\|x   256\|x   512\|small"
} 2>&1 | tee gpurun_out/r4g/big_tile.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4g/gpu_tests_bigtile.txt
