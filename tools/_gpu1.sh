cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cfg5 --no-cpu-baseline > gpurun_out/r4b/bench_perm.json 2>/dev/null
python -c "
import json; l=json.load(open('gpurun_out/r4b/bench_perm.json')); r=l['roofline']
print('value', l['value'], 'ms', l['ms_per_step'], 'frac', r['frac'], 'with_plan', r.get('ms_with_plan'))
print('huff', l['roofline_huffman']['decompressed_GBps'], 'stored', l['roofline_stored']['frac'], 'interleaved ms', l['roofline_interleaved']['avg_step_ms'], l['roofline_interleaved']['frac'])
"
python tools/bench_mixed_order.py 2>&1 | grep -v amdgpu.ids | tail -12
