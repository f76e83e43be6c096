import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.batch import DeviceBatch
kind = sys.argv[1]; n = int(sys.argv[2]); width = int(sys.argv[3], 0); size = int(sys.argv[4])
pairs = workload.make_streams(kind, n, size, threads=16)
raws = [p[0] for p in pairs]; caps = [max(size + 1, len(r)) for r in raws]
b = DeviceBatch.from_streams(raws, caps)
for it in range(3):
    b.launch(waves_per_stream=width)
    torch.cuda.synchronize()
    res = b.results()
    bad = [i for i in range(n) if res["good"][i] != 1 or res["final_size"][i] != size]
    wrong = [i for i in range(n) if i not in bad and b.output(i, res) != pairs[i][1].tobytes()]
    print(f"iter {it}: n={n} bad={len(bad)} wrong={len(wrong)}", [(i, int(res['good'][i]), int(res['status'][i]), int(res['final_size'][i]), int(res['n_blocks'][i]), int(res['n_windows'][i])) for i in bad[:6]], wrong[:6], flush=True)
    for i in wrong[:2]:
        a = np.frombuffer(b.output(i, res), dtype=np.uint8); p = pairs[i][1]
        d = np.nonzero(a != p)[0]
        print("  stream", i, "diffs", len(d), "first", d[:4], "last", d[-4:])
