# lane utilisation of the inflate kernel: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-fixed}
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN --output-format csv -d $R/gpurun_out/pmc_lanes_$K -- python3 $R/tools/bench_variant.py $K 4096 > $R/gpurun_out/pmc_lanes_$K.log 2>&1
cd $R
python3 - <<PY
import csv, glob, os
f = max(glob.glob("gpurun_out/pmc_lanes_$K/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "debig_inflate_kernel" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows)
v = {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last}
for k in sorted(v): print(f"{k:28s} {v[k]:.6g}")
if "SQ_THREAD_CYCLES_VALU" in v and "SQ_ACTIVE_INST_VALU" in v:
    print("active lanes per VALU issue cycle:", v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"])
PY
