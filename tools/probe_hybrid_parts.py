#!/usr/bin/env python3
"""Diagnostic: the parts of DevicePngBatch.launch_hybrid on config 3, each alone and together."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import _native as N
from debigulator_amd.png_device import DevicePngBatch


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


files = [f for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "resources", "*.png"))) if not f.endswith("backgrounddetailed1.png")]
datas = [open(f, "rb").read() for f in files]
b = DevicePngBatch([datas[i % len(datas)] for i in range(1024)])
print(f"hybrid together {timeit(b.launch_hybrid):8.3f} ms")
for key, idx, sub, side in b.hybrid["parts"]:
    if key == "long":
        fn = lambda sub=sub: sub.launch(waves_per_stream=N.WAVES_CHUNKED, fused=False)
        t_inf = timeit(lambda sub=sub: sub.inflate.launch(waves_per_stream=N.WAVES_CHUNKED))
        extra = f"(inflate alone {t_inf:.3f})"
    elif key == "rest":
        fn, extra = sub.launch_fused, ""
    else:
        fn, extra = (lambda sub=sub: sub.launch(fused=False, hybrid=False)), ""
    print(f"  part {key:5s} {sub.n:5d} images  alone {timeit(fn):8.3f} ms {extra}", flush=True)
