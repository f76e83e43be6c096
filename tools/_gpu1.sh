cd $GRAFT_REPO_ROOT
python tools/ab_variants.py run kinds=fixed,dynamic width=0x12
for w in 0x10 0x12; do python tools/bench_variant.py dynamic 2048 $w 1048576 | tail -1; done
for w in 0x10 0x12; do python tools/bench_variant.py fixed 8192 $w 65536 | tail -1; done
for w in 0x10 0x12; do python tools/bench_variant.py fixed 16384 $w 65536 | tail -1; done
