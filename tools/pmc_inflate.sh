cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/pmc_list.txt 2>&1
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

pmc_pass $R/gpurun_out/pmc_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -- python3 $R/tools/bench_variant.py fixed 4096 > $R/gpurun_out/pmc_a.log 2>&1
pmc_pass $R/gpurun_out/pmc_b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- python3 $R/tools/bench_variant.py fixed 4096 > $R/gpurun_out/pmc_b.log 2>&1
cd $R; find gpurun_out/pmc_a gpurun_out/pmc_b -name "*counter_collection*" | head
