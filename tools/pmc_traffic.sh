#!/bin/bash
# HBM traffic of the bench's launches from PMC counters (separate passes: FETCH_SIZE needs 3
# TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output: CSVs under
# gpurun_out/pmc_traffic_{fetch,write}; summarised by tools/pmc_summary.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_traffic_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg5 --no-cfg3 --no-kinds > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_traffic_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg5 --no-cfg3 --no-kinds > $R/gpurun_out/pmc_write.log 2>&1
