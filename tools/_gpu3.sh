cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
python tools/bench_fused_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4f/hybrid_cfg3.txt
{
for spec in "dynamic 1048576" "png 1048576"; do set -- $spec
for n in 128 256 512 1024; do
  for w in 0 0x20; do python tools/bench_variant.py $1 $n $w $2 2>&1 | tail -1; done
done; done
} | tee gpurun_out/r4f/chunked_grid.txt
