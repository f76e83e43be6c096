#!/usr/bin/env python3
"""Diagnostic: time one library variant (DEBIG_LIB=...) on 64 KiB streams of one kind.
usage: bench_variant.py [kind=fixed] [n=4096] [width=0 (library's choice) | 1 | 0x10 ...] [size=65536]
Prints kernel time per launch (events on the launch stream), GB/s decompressed, exactness."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.batch import DeviceBatch
kind = sys.argv[1] if len(sys.argv) > 1 else "fixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
width = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
size = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
pairs = workload.make_streams(kind, n, size, threads=16)
raws = [p[0] for p in pairs]; caps = [max(size + 1, len(r)) for r in raws]
b = DeviceBatch.from_streams(raws, caps)
for _ in range(3): b.launch(waves_per_stream=width)
torch.cuda.synchronize()
ts = []
cold = os.environ.get("COLD") == "1"
junk = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda") if cold else None
for _ in range(10):
    if cold:
        junk.add_(1)  # 2 GiB of traffic: evicts L2 and the 256 MiB Infinity Cache
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.launch(waves_per_stream=width); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
res = b.results()
if os.environ.get("NOCHECK") == "1":  # timing-only builds (DEBIG_ABLATE): the output is wrong on purpose
    ok = None
else:
    assert (res["good"] == 1).all() and (res["final_size"] == size).all()
    ok = all(b.output(i, res) == pairs[i][1].tobytes() for i in range(0, n, max(1, n // 32)))
ms = float(np.median(ts))
print(f"{os.environ.get('DEBIG_LIB','default').split('/')[-1]:28s} {kind:8s} n={n:6d} width={width:#x} {ms:8.3f} ms  {n*size/ms/1e6:8.1f} GB/s  exact={ok} cold={cold}", flush=True)
