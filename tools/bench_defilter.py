#!/usr/bin/env python3
"""Diagnostic: de-filter kernel alone (4 all-Paeth RGBA images per shape, one wavefront each):
time per macro-step (4 pixels of each of 64 rows) as a function of the image shape."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.png_device import DevicePngBatch
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for w,h in [(256,16384),(1024,4096),(4096,1024),(4096,4096)]:
    png = workload.make_png(77, w, h, ct=6, ftype=4, noise=24, enc="fixed")[0]
    b = DevicePngBatch([png]*4)
    b.launch(); torch.cuda.synchronize()
    t = timeit(b.launch_defilter_only)
    steps = ((h+63)//64) * ((w+3)//4 + 63)
    print(f"{w}x{h}: de-filter {t:8.3f} ms, {steps} macro-steps/image, {t*1e3/steps:6.3f} us per macro-step, row stride {4*w+1} B", flush=True)
