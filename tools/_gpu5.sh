cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 8 16 32; do
  DEBIG_BENCH_FUSED=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4_$n -- python3 $GRAFT_REPO_ROOT/tools/bench_png.py cfg4 8192 $n > $O/prof_cfg4_$n.log 2>&1
  f=$(find $O/prof_cfg4_$n -name "*kernel_stats.csv" | head -1)
  echo "== cfg4 8192 $n"; grep -v amdgpu.ids $O/prof_cfg4_$n.log | tail -6; cut -d, -f1-5 $f | head -16
done 2>&1 | tee $O/cfg4_kernels_by_n.txt
