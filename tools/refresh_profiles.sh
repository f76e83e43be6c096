#!/bin/bash
# End-of-round evidence, in one GPU call: PMC traffic passes (feed bench.py's roofline.traffic),
# rocprofv3 kernel stats of the bench command, instruction counters and phase shares of the
# scan / LZ77 kernel pair, and the bench line itself.  Everything lands in gpurun_out/refresh/;
# copy what is judged to profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
python3 tools/pmc_summary.py > $O/pmc_traffic.txt 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 20 --no-cpu-baseline > $O/prof_bench.log 2>&1 )
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
bash tools/pmc_split.sh fixed > $O/pmc_split_fixed.txt 2>&1 || true
python3 tools/prof_split.py fixed > $O/phases_split_fixed.txt 2>&1 || true
python3 tools/prof_split.py dynamic > $O/phases_split_dynamic.txt 2>&1 || true
python3 bench.py > $O/bench_line.json 2> $O/bench_stderr.log
tail -1 $O/bench_line.json | cut -c1-600
