cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout -k 10 600 python -m pytest tests/test_fused_png.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4e/fused_tests.txt
cat gpurun_out/r4e/fused_tests.txt
grep -q passed gpurun_out/r4e/fused_tests.txt && ! grep -q failed gpurun_out/r4e/fused_tests.txt || exit 1
for wpe in 2 3; do
echo "---- cfg3 DEBIG_FUSED_WPE=$wpe"
DEBIG_FUSED_WPE=$wpe timeout -k 10 300 python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu.ids | tail -5
done | tee gpurun_out/r4e/cfg3_fused.txt
