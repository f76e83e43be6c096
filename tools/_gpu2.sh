cd $GRAFT_REPO_ROOT
for n in 320 384 512; do for w in 4 0x13; do python tools/bench_variant.py dynamic $n $w 1048576 2>&1 | tail -1; done; done
for n in 384 512 768; do for w in 4 2 0x13; do python tools/bench_variant.py png $n $w 1048576 2>&1 | tail -1; done; done
for n in 512 768; do for w in 4 2 0x13; do python tools/bench_variant.py fixed $n $w 65536 2>&1 | tail -1; done; done
echo "---- cfg3 pipe"
DEBIG_WAVES_PER_STREAM=0x13 DEBIG_WORKSPACE_MB=16000 python tools/bench_png.py cfg3 2>&1 | grep -v amdgpu.ids | tail -3
F=tests/golden/resources
for f in fs_angrymob.png gimp_test.png; do for n in 128 365; do python tools/bench_file_stream.py $F/$f $n 0x13 2>&1 | tail -1; done; done
