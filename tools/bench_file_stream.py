#!/usr/bin/env python3
"""Diagnostic: N copies of ONE sample PNG's IDAT stream through a given inflate width, timed
(DEBIG_LIB selects a library variant).  usage: bench_file_stream.py FILE.png [copies=256] [width=0x10]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, zlib
from debigulator_amd.batch import DeviceBatch
from debigulator_amd.png_device import split_png
f = sys.argv[1]
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 256
width = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x10
it = split_png(open(f, "rb").read())
raw = it["raw"]; est = 4 * it["w"] * it["h"] + it["h"] + 1
b = DeviceBatch.from_streams([raw] * copies, [est] * copies)
if width in (0x10, 0x12, 0x13):
    b.d_ws = torch.empty(20 * len(raw) * copies + (1 << 26), dtype=torch.uint8, device="cuda")
for _ in range(2): b.launch(waves_per_stream=width)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.launch(waves_per_stream=width); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
res = b.results()
want = zlib.decompress(raw, -15)
n = int(res["final_size"][0])
ok = bool((res["good"] == 1).all()) and b.output(0, res)[:n] == want[:n] and b.output(copies - 1, res)[:n] == want[:n]
ms = float(np.median(ts))
print(f"{os.environ.get('DEBIG_LIB','default').split('/')[-1]:32s} {os.path.basename(f):22s} x{copies} width={width:#x} {ms:9.3f} ms  {copies*n/ms/1e6:7.1f} GB/s  exact={ok}", flush=True)
