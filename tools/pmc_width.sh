# PMC passes over one inflate width: tools/bench_variant.py $1 (kind) 4096 $2 (width), kernel name $3
# (<= 8 SQ counters per pass: more abort in rocprofiler_create_counter_config)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-fixed}
W=${2:-0x12}
KERN=${3:-debig_strand_kernel}
T=$R/gpurun_out/pmcw_${K}_$W
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

pmc_pass ${T}_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -- python3 $R/tools/bench_variant.py $K 4096 $W > ${T}_a.log 2>&1
pmc_pass ${T}_b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- python3 $R/tools/bench_variant.py $K 4096 $W > ${T}_b.log 2>&1
pmc_pass ${T}_c SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH -- python3 $R/tools/bench_variant.py $K 4096 $W > ${T}_c.log 2>&1
cd $R
python3 - <<PY
import csv, glob, os
kern = "$KERN"
print("==", kern, "$K", "$W")
tot = {}
for pas in "abc":
    fs = glob.glob("gpurun_out/pmcw_${K}_${W}_%s/**/*counter_collection.csv" % pas, recursive=True)
    if not fs: continue
    f = max(fs, key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    if not rows: continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last: tot[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(tot): print(f"  {k:26s} {tot[k]:.6g}")
w = tot.get("SQ_WAVES", 4096)
if "SQ_INSTS_VALU" in tot:
    print(f"  per stream: VALU {tot['SQ_INSTS_VALU']/w:.0f}  SALU {tot.get('SQ_INSTS_SALU',0)/w:.0f}  LDS {tot.get('SQ_INSTS_LDS',0)/w:.0f}  VMEM {tot.get('SQ_INSTS_VMEM',0)/w:.0f}  branch {tot.get('SQ_INSTS_BRANCH',0)/w:.0f}")
if "SQ_THREAD_CYCLES_VALU" in tot and tot.get("SQ_ACTIVE_INST_VALU"):
    print(f"  active lanes per VALU issue cycle: {tot['SQ_THREAD_CYCLES_VALU']/tot['SQ_ACTIVE_INST_VALU']:.1f}")
if "SQ_LDS_BANK_CONFLICT" in tot and tot.get("SQ_LDS_IDX_ACTIVE"):
    print(f"  LDS conflict share: {tot['SQ_LDS_BANK_CONFLICT']/tot['SQ_LDS_IDX_ACTIVE']:.2f}")
PY
