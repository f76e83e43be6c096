#!/usr/bin/env python3
"""Diagnostic: time one library variant (DEBIG_LIB=...) on cfg2 fixed/dynamic streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.batch import DeviceBatch
kind = sys.argv[1] if len(sys.argv) > 1 else "fixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
pairs = workload.make_streams(kind, n, 65536)
raws = [p[0] for p in pairs]; caps = [max(65537, len(r)) for r in raws]
b = DeviceBatch.from_streams(raws, caps)
for _ in range(3): b.launch()
torch.cuda.synchronize()
ts = []
cold = os.environ.get("COLD") == "1"
junk = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda") if cold else None
for _ in range(10):
    if cold:
        junk.add_(1)  # 2 GiB of traffic: evicts L2 and the 256 MiB Infinity Cache
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.launch(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
res = b.results(); assert (res["good"] == 1).all() and (res["final_size"] == 65536).all()
ok = all(b.output(i, res) == pairs[i][1].tobytes() for i in range(0, n, max(1, n // 32)))
ms = float(np.median(ts))
print(f"{os.environ.get('DEBIG_LIB','default').split('/')[-1]:28s} {kind:8s} {ms:8.3f} ms  {n*65536/ms/1e6:8.1f} GB/s  exact={ok} cold={cold}")
