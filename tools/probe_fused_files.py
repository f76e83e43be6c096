import glob, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd.png_device import DevicePngBatch
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
files = [f for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "resources", "*.png"))) if not f.endswith("backgrounddetailed1.png")]
for f in files:
    d = open(f, "rb").read()
    b = DevicePngBatch([d] * 73)
    t_pair = timeit(lambda: b.launch(fused=False))
    t_f = timeit(b.launch_fused)
    os.environ["DEBIG_FUSED_FLAGS"] = "3"
    b.launch_fused(); res, ires = b.results()
    os.environ.pop("DEBIG_FUSED_FLAGS")
    nretry = int((res["status"] == 0x7fffffff).sum()) if False else int((res["good"] == 0).sum())
    print(f"{os.path.basename(f):26s} x73  pair {t_pair:8.3f}  fused {t_f:8.3f}   handed back {nretry}  blocks {int(res['n_blocks'][0])} windows {int(res['n_windows'][0])} in {len(b.items[0]['raw'])} out {int(res['final_size'][0])}", flush=True)
