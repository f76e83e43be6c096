cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/ab_variants.py run kinds=fixed,dynamic width=0x12
python tools/ab_variants.py run kinds=fixed width=0x10 only=base
done
