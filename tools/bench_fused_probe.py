#!/usr/bin/env python3
"""Diagnostic: the fused inflate -> de-filter kernel against the pair of launches, config 3's mix of sample files at
several batch sizes (and its two halves: the five photo-like fs_* files, the nine smaller ones)."""
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
cases = [("all 14", names, c) for c in (14, 28, 42, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 4096)] + [("5 fs_*", big, 365), ("9 small", small, 659)]
for label, sel, count in cases:
    pngs = [datas[sel[i % len(sel)]] for i in range(count)]
    b = DevicePngBatch(pngs)
    t_pair = timeit(lambda: b.launch(fused=False, hybrid=False))
    t_f = timeit(b.launch_fused)
    res, ires = b.results()
    assert (res["good"] == 1).all() and (ires["good"] == 1).all()
    t_auto = timeit(b.launch)
    how = "hybrid" if b.last_hybrid else "fused" if b.last_fused else "pair"
    t_h = timeit(lambda: b.launch(hybrid=True))
    t_h2 = timeit(lambda: b.launch(hybrid=True))
    t_h = f"{t_h:.3f} / {t_h2:.3f}"
    how2 = "hybrid" if b.last_hybrid else "fused" if b.last_fused else "pair"
    print(f"{label:8s} x {count:5d}   pair {t_pair:8.3f} ms   fused {t_f:8.3f} ms   launch() {t_auto:8.3f} ms ({how})   hybrid {t_h} ms   {b.rgba_bytes / t_auto / 1e6:7.1f} GB/s of RGBA", flush=True)
    del b
    torch.cuda.empty_cache()
