# lane utilisation of the inflate kernel: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-fixed}
# one --pmc pass: at most 8 SQ counters (more abort in rocprofiler_create_counter_config: "Request exceeds the
# capabilities of the hardware to collect", gpurun_out/r3w/pmc_c.log of round 3)
pmc_pass() {
    local out=$1; shift
    local n=0 a
    for a in "$@"; do [ "$a" = "--" ] && break; n=$((n + 1)); done
    if [ $n -gt 8 ]; then echo "pmc_pass: $n counters in one pass (limit 8)" >&2; exit 2; fi
    local counters=("${@:1:$n}"); shift $((n + 1))
    rocprofv3 --kernel-trace --pmc "${counters[@]}" --output-format csv -d "$out" -- "$@"
}

pmc_pass $R/gpurun_out/pmc_lanes_$K SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN -- python3 $R/tools/bench_variant.py $K 4096 > $R/gpurun_out/pmc_lanes_$K.log 2>&1
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
