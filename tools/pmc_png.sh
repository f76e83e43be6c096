#!/bin/bash
# HBM traffic of a decode_png batch (config 3 or the config 4 shape) per kernel, FETCH_SIZE and
# WRITE_SIZE in separate passes; summary: tools/pmc_png_summary.py.  usage: pmc_png.sh cfg3 | cfg4 SIDE COUNT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_png_fetch -- python3 $R/tools/bench_png.py "$@" > $R/gpurun_out/pmc_png_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_png_write -- python3 $R/tools/bench_png.py "$@" > $R/gpurun_out/pmc_png_write.log 2>&1
