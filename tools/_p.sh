mkdir -p gpurun_out/r3y
for v in nofx new nofx new; do
  if [ $v = new ]; then unset DEBIG_LIB; else export DEBIG_LIB=$PWD/debigulator_amd/lib/libdebigulator_hip_ab_$v.so; fi
  echo "== $v"
  python tools/bench_variant.py fixed 4096 0x10 2>&1 | grep -v amdgpu
  python tools/bench_variant.py dynamic 4096 0x10 2>&1 | grep -v amdgpu
done
unset DEBIG_LIB
python bench.py --no-cfg5 --no-kinds --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
python tools/prof_split.py fixed 2>&1 | grep -v amdgpu | head -9
