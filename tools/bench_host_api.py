#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in batch call (host buffers in, host buffers out):
debig_inflate_batch over config-2 streams.  Reported in DESIGN.md; never bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: see _native.lib)
from debigulator_amd import api, workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for kind in ("stored", "fixed"):
    pairs = workload.make_streams(kind, n, 65536)
    raws = [p[0] for p in pairs]
    caps = [max(65537, len(r)) for r in raws]
    api.inflate_batch(raws[:64], caps[:64])
    t0 = time.perf_counter()
    res = api.inflate_batch(raws, caps)
    dt = time.perf_counter() - t0
    assert all(g == 1 and f == 65536 for g, f, _ in res)
    assert res[7][2] == pairs[7][1].tobytes()
    print(f"host-buffer batch, {kind:6s}: {n} streams, {dt*1e3:8.1f} ms, {n*65536/dt/1e9:6.2f} GB/s decompressed "
          f"(H2D + kernel + D2H + host packing)")
