cd $GRAFT_REPO_ROOT
for k in png fixed; do
for n in 768 1024; do for w in 2 0x10 0x12; do python tools/bench_variant.py $k $n $w 65536 2>&1 | tail -1; done; done
for n in 2048 3072; do for w in 0x10 0x12; do python tools/bench_variant.py $k $n $w 65536 2>&1 | tail -1; done; done
done
for n in 768 1024; do for w in 2 0x10 0x12; do python tools/bench_variant.py png $n $w 1048576 2>&1 | tail -1; done; done
for w in 0x10 0x12; do python tools/bench_variant.py png 2048 $w 1048576 2>&1 | tail -1; done
