cd $GRAFT_REPO_ROOT
for k in fixed dynamic png; do for w in 0x10 0x12; do python tools/bench_variant.py $k 4096 $w 65536 2>&1 | tail -1; done; done
for k in fixed png; do for w in 0x10 0x12; do python tools/bench_variant.py $k 2048 $w 65536 2>&1 | tail -1; done; done
for w in 0x10 0x12; do python tools/bench_variant.py dynamic 2048 $w 1048576 2>&1 | tail -1; done
for w in 0x10 0x12; do python tools/bench_gz.py 8192 1048576 $w 2>&1 | grep "inflate  "; done
