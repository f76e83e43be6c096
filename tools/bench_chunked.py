#!/usr/bin/env python3
"""Measurement: a few LARGE streams through the chunk-parallel path (DEBIG_WAVES_CHUNKED) against
the library's own choice for the batch.  kind: png (all-Paeth RGBA scanline streams, config 4's
shape) | dynamic | fixed (text-like payload).
    python tools/bench_chunked.py KIND COUNT MBYTES_EACH"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload, _native as N
from debigulator_amd.batch import DeviceBatch
from debigulator_amd.png_device import split_png

kind = sys.argv[1] if len(sys.argv) > 1 else "png"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mbytes = float(sys.argv[3]) if len(sys.argv) > 3 else 16
distinct = min(count, 4)
t0 = time.time()
raws, plains = [], []
for s in range(distinct):
    if kind == "png":
        side = int((mbytes * 1e6 / 4) ** 0.5) // 64 * 64
        png, _ = workload.make_png(9000 + s, side, side, ct=6, ftype=4, noise=workload.CFG4_NOISE, enc="dynamic")
        raw = split_png(png)["raw"]
        raws.append(raw); plains.append(zlib.decompress(raw, -15))
    else:
        r, p = workload.make_stream(kind, 100 + s, size=int(mbytes * 1e6))
        raws.append(bytes(r)); plains.append(bytes(p))
raws = [raws[i % distinct] for i in range(count)]
plains_i = [i % distinct for i in range(count)]
caps = [len(plains[j]) for j in plains_i]
S, Cb = sum(caps), sum(len(r) for r in raws)
print(f"{kind}: {count} streams, {Cb/1e6:.1f} MB -> {S/1e6:.1f} MB (ratio {S/Cb:.2f}), made in {time.time()-t0:.0f} s", flush=True)


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for name, w in (("auto", 0), ("chunked", N.WAVES_CHUNKED)):
    b = DeviceBatch.from_streams(raws, caps)
    ms = timeit(lambda: b.launch(waves_per_stream=w))
    res = b.results()
    ok = bool((res["good"] == 1).all() and (res["final_size"] == np.array(caps)).all())
    same = all(b.output(i, res) == plains[plains_i[i]] for i in sorted({0, count // 2, count - 1}))
    extra = f" groups {len(b.chunk_groups)}" if w else ""
    print(f"  {name:8s} {ms:10.3f} ms  {S/ms/1e6:8.1f} GB/s of output   good={ok} bytes_exact={same}{extra}", flush=True)
    del b
    torch.cuda.empty_cache()
