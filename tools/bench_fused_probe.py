#!/usr/bin/env python3
"""Diagnostic: the fused inflate -> de-filter kernel on subsets of config 3 and with larger workspaces."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
datas = {os.path.basename(f): open(f, "rb").read() for f in files}
names = sorted(datas)
big = [n for n in names if n.startswith("fs_")]
small = [n for n in names if not n.startswith("fs_")]
for label, sel, count in (("all 14 x 1024", names, 1024), ("5 fs_* x 365", big, 365), ("9 small x 659", small, 659), ("5 fs_* x 128", big, 128)):
    pngs = [datas[sel[i % len(sel)]] for i in range(count)]
    b = DevicePngBatch(pngs)
    t_pair = timeit(b.launch)
    t_f = timeit(b.launch_fused)
    f = b.fused
    big_ws = torch.empty(3 * f["ws_bytes"], dtype=torch.uint8, device="cuda")
    f["d_ws"], f["ws_bytes"] = big_ws, big_ws.numel()
    t_f3 = timeit(b.launch_fused)
    res, ires = b.results()
    assert (res["good"] == 1).all() and (ires["good"] == 1).all()
    print(f"{label:18s} pair {t_pair:8.3f} ms   fused {t_f:8.3f} ms   fused, 3 x workspace {t_f3:8.3f} ms   windows/stream {res['n_windows'].mean():.1f}", flush=True)
    del b, big_ws
    torch.cuda.empty_cache()
