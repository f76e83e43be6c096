cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/ab_variants.py run kinds=fixed,dynamic,png width=0x12
done
python tools/ab_variants.py run kinds=fixed,dynamic,png width=0x10 only=base
python tools/ab_variants.py run kinds=fixed,png width=0x12 n=2048
