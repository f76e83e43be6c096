#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in batch call (host buffers in, host buffers out):
debig_inflate_batch over config-2 streams, timing ONLY the C call (buffers and pointer arrays
are prepared before; the caller's buffers are ordinary pageable memory).  Reported in DESIGN.md;
never bench.py's `value`.   usage: bench_host_api.py [n=2048]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (first: see _native.lib)
from debigulator_amd import _native as N, workload

L = N.lib()
L.debig_inflate_batch.restype = C.c_int
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NOT_SET = 0xFFFFFFFFFFFFFFFF
for kind in ("stored", "fixed", "dynamic"):
    pairs = workload.make_streams(kind, n, 65536)
    ins = [np.frombuffer(p[0], dtype=np.uint8) for p in pairs]
    caps_l = [max(65537, len(a)) for a in ins]
    outs = [np.zeros(c, dtype=np.uint8) for c in caps_l]
    in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins])
    out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    in_sizes = (C.c_uint64 * n)(*[len(a) for a in ins])
    caps = (C.c_uint64 * n)(*caps_l)
    finals = (C.c_uint64 * n)(*([NOT_SET] * n))
    goods = (C.c_uint32 * n)()
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        rc = L.debig_inflate_batch(out_ptrs, caps, finals, in_ptrs, in_sizes, goods, n, 0)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    assert all(goods[i] == 1 and finals[i] == 65536 for i in range(n))
    assert outs[7][:65536].tobytes() == pairs[7][1].tobytes() and outs[n - 1][:65536].tobytes() == pairs[n - 1][1].tobytes()
    dt = min(ts[1:])
    print(f"host-buffer batch, {kind:7s}: {n} streams, {dt*1e3:8.1f} ms, {n*65536/dt/1e9:6.2f} GB/s decompressed "
          f"(C call only: pack + H2D + kernels + D2H + unpack, best of {len(ts)-1})", flush=True)
