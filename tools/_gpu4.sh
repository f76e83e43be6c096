cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
{
timeout -k 10 500 python tools/probe_png_pipeline.py 8192 32 16x2 8x2 8x4 4x4 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r4h/png_pipeline_lanes.txt
