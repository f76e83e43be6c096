cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout -k 10 900 python bench.py --config cfg4 --images 32 --no-cpu-baseline > gpurun_out/r4f/cfg4_32.json 2> gpurun_out/r4f/cfg4_32.err; tail -c 1500 gpurun_out/r4f/cfg4_32.json
