cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
( time timeout -k 10 900 python bench.py > gpurun_out/r4g/bench_line.json 2> gpurun_out/r4g/bench_err.txt ) 2>&1 | tail -3
python3 -c "
import json
d=json.loads(open('gpurun_out/r4g/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
print(json.dumps(d['cfg3_png'], indent=1)[:1500])"
tail -3 gpurun_out/r4g/bench_err.txt
