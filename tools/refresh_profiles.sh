#!/bin/bash
# End-of-round evidence, in one GPU call: PMC traffic passes (feed bench.py's roofline.traffic),
# rocprofv3 kernel statistics of the bench command and of each stream kind alone, the FETCH_SIZE
# calibration of the LZ77 kernel's access shapes, instruction counters and phase shares of the
# scan / LZ77 kernel pair, and the bench line itself.  Everything lands in gpurun_out/refresh/;
# copy what is judged to profiles/ (named per round).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
echo "[refresh] pmc traffic"; bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
python3 tools/pmc_summary.py > $O/pmc_traffic.txt 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
echo "[refresh] kernel stats"
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_blend -- python3 $R/bench.py --steps 20 --no-cpu-baseline --no-cfg5 --no-cfg3 --no-kinds > $O/prof_blend.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fixed -- python3 $R/tools/bench_variant.py fixed 4096 0 > $O/prof_fixed.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stored -- python3 $R/tools/bench_variant.py stored 4096 0 > $O/prof_stored.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_cal -- $R/tools/bin/ubench_fetch > $O/fetch_cal.log 2>&1 )
for k in blend fixed stored; do find $O/prof_$k -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$k.csv \; ; done
python3 - <<PY > $O/fetch_calibration.txt 2>&1
import csv, glob
print(open("$O/fetch_cal.log").read())
for f in glob.glob("$O/fetch_cal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Kernel_Name'].split('(')[0]:12s} dispatch {r['Dispatch_Id']:>3s}  {r['Counter_Name']} = {float(r['Counter_Value'])*1024/1e6:10.1f} MB (raw counter x 1 KiB)")
PY
echo "[refresh] counters and phases"
bash tools/pmc_split.sh fixed > $O/pmc_split_fixed.txt 2>&1 || true
python3 tools/prof_split.py fixed 4096 65536 0x10 > $O/phases_split_fixed.txt 2>&1 || true
python3 tools/prof_split.py dynamic 4096 65536 0x10 > $O/phases_split_dynamic.txt 2>&1 || true
echo "[refresh] bench line"
python3 bench.py > $O/bench_line.json 2> $O/bench_stderr.log
tail -1 $O/bench_line.json | cut -c1-400
echo "[refresh] the long-segment scan (DEBIG_WAVES_STRAND): counters, phases"
bash tools/pmc_width.sh fixed 0x12 debig_strand_kernel > $O/pmc_strand_fixed.txt 2>&1 || true
python3 tools/prof_split.py fixed 4096 65536 0x12 > $O/phases_strand_fixed.txt 2>&1 || true
python3 tools/prof_split.py dynamic 2048 1048576 0x12 > $O/phases_strand_dynamic_1m.txt 2>&1 || true
