cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
L=debigulator_amd/lib
F=tests/golden/resources
{
for v in base group1; do
  if [ $v = base ]; then unset DEBIG_LIB; else export DEBIG_LIB=$PWD/$L/libdebigulator_hip_ab_$v.so; fi
  echo "#### $v"
  for f in fs_angrymob.png gimp_test.png purpleback.png; do python tools/bench_file_stream.py $F/$f 128 0x13 2>&1 | tail -1; done
  python tools/bench_file_stream.py $F/fs_bribery.png 365 0x20 2>&1 | tail -1
  python tools/bench_file_stream.py $F/fs_bribery.png 1 0x20 2>&1 | tail -1
  for k in fixed dynamic png; do python tools/bench_variant.py $k 4096 0x10 2>&1 | tail -1; done
  python tools/bench_variant.py dynamic 512 0x13 1048576 2>&1 | tail -1
  python tools/bench_variant.py png 768 0x13 1048576 2>&1 | tail -1
  DEBIG_BENCH_FUSED=1 timeout -k 10 300 python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu.ids | tail -3
  DEBIG_BENCH_FUSED=0 timeout -k 10 300 python tools/bench_png.py cfg4 8192 8 2>&1 | grep -v amdgpu.ids | tail -4 | head -2
done
} 2>&1 | tee gpurun_out/r4f/group2_ab.txt
