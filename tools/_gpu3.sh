cd $GRAFT_REPO_ROOT
for cb in 98304 131072 196608 262144; do echo -n "DEBIG_CHUNK_BYTES=$cb "; DEBIG_CHUNK_BYTES=$cb python tools/probe_hybrid_parts.py 2>&1 | grep -v amdgpu.ids | head -2 | tr '\n' ' '; echo; done 2>&1 | tee -a gpurun_out/r4f/chunk_bytes.txt
