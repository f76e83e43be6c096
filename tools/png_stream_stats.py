#!/usr/bin/env python3
"""Diagnostic: per-file decode counters of the reference's sample PNGs (blocks, input windows,
speculation rounds) and the time each stream takes alone, 1 vs 4 wavefronts."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd.png_device import DevicePngBatch

files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "resources", "*.png")))
print(f"{'file':28s} {'C KB':>8s} {'S KB':>8s} {'blocks':>6s} {'win':>6s} {'rnd/win':>7s} {'C/blk KB':>8s} {'ms w=1':>8s} {'ms w=4':>8s}")
for f in files:
    data = open(f, "rb").read()
    b = DevicePngBatch([data])
    ts = {}
    for w in (1, 4):
        for _ in range(2):
            b.inflate.launch(waves_per_stream=w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.inflate.launch(waves_per_stream=w); e1.record(); torch.cuda.synchronize()
        ts[w] = e0.elapsed_time(e1)
    r = b.inflate.results()[0]
    c = b.c_bytes
    print(f"{os.path.basename(f):28s} {c/1e3:8.1f} {b.s_bytes/1e3:8.1f} {r['n_blocks']:6d} {r['n_windows']:6d} "
          f"{r['n_rounds']/max(1,r['n_windows']):7.2f} {c/1e3/max(1,r['n_blocks']):8.2f} {ts[1]:8.3f} {ts[4]:8.3f}")
