cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4m/gpu_tests.txt
{
for n in 8 16 32 64 128; do
DEBIG_BENCH_FUSED=0 timeout -k 10 300 python tools/bench_png.py cfg4 8192 $n 2>&1 | grep -v amdgpu.ids | tail -5
done
} 2>&1 | tee gpurun_out/r4m/cfg4_final.txt
python bench.py --config cfg4 --images 32 --steps 5 --warmup 1 > gpurun_out/r4m/bench_cfg4_32.json 2> gpurun_out/r4m/bench_cfg4_32.err
tail -1 gpurun_out/r4m/bench_cfg4_32.json | cut -c1-300
echo "[pmc traffic]"
bash tools/pmc_traffic.sh > gpurun_out/r4m/pmc_traffic.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py > gpurun_out/r4m/pmc_traffic.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/r4m/pmc_traffic.json
tail -3 gpurun_out/r4m/pmc_traffic.txt
echo "[bench line]"
python3 bench.py > gpurun_out/r4m/bench_line.json 2> gpurun_out/r4m/bench_stderr.log
tail -1 gpurun_out/r4m/bench_line.json | cut -c1-200
