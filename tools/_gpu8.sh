cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
run() { n=$1; shift; echo "== n=$n $*"; env "$@" DEBIG_BENCH_FUSED=0 timeout -k 10 200 python tools/bench_png.py cfg4 8192 $n 2>&1 | grep "de-filter only\|two launches\|rror\|assert"; }
{
run 32 DEBIG_DEFILTER_WGS=8 DEBIG_DEFILTER_WG_WAVES=4
run 32 DEBIG_DEFILTER_WGS=8 DEBIG_DEFILTER_WG_WAVES=8
run 64 DEBIG_DEFILTER_WGS=4 DEBIG_DEFILTER_WG_WAVES=4
} 2>&1 | tee gpurun_out/r4k/mask_select_shapes.txt
