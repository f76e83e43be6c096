cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
run() { # n, env...
  n=$1; shift
  echo "== n=$n $*"
  env "$@" DEBIG_BENCH_FUSED=0 timeout -k 10 200 python tools/bench_png.py cfg4 8192 $n 2>&1 | grep "de-filter only\|two launches\|Error\|error\|assert" 
}
{
run 32 DEBIG_DEFILTER_WGS=8 DEBIG_DEFILTER_WG_WAVES=8
run 32 DEBIG_DEFILTER_WGS=8 DEBIG_DEFILTER_WG_WAVES=4
run 16 DEBIG_DEFILTER_WGS=16 DEBIG_DEFILTER_WG_WAVES=4
run 16 DEBIG_DEFILTER_WGS=8 DEBIG_DEFILTER_WG_WAVES=8
run 64 X=1
run 64 DEBIG_DEFILTER_WGS=4 DEBIG_DEFILTER_WG_WAVES=8
run 64 DEBIG_DEFILTER_PXSKEW=0
run 128 X=1
run 128 DEBIG_DEFILTER_PXSKEW=0
run 128 DEBIG_DEFILTER_WGS=2 DEBIG_DEFILTER_WG_WAVES=4
} 2>&1 | tee gpurun_out/r4j/defilter_px_shapes.txt
