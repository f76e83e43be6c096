#!/usr/bin/env python3
"""Config 4 shape: the de-filter of one group of images beside the inflate of the next (DevicePngBatch.launch_pipelined)
against the two launches one after the other.  usage: probe_png_pipeline.py [side] [count] [images per group[xlanes] ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from debigulator_amd import workload
from debigulator_amd.png_device import DevicePngBatch

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
count = int(sys.argv[2]) if len(sys.argv) > 2 else 32
groups = [tuple(int(x) for x in a.split("x")) + (1,) * (2 - len(a.split("x"))) for a in sys.argv[3:]] or [(16, 1), (8, 1), (4, 1), (8, 2), (4, 2)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(min(ts))


t0 = time.time()
with ThreadPoolExecutor(4) as ex:
    made = list(ex.map(lambda s: workload.make_png(9000 + s, side, side, ct=6, ftype=4, noise=workload.CFG4_NOISE,
                                                   enc="dynamic", idat_chunk=65536), range(min(count, 4))))
pngs = [made[i % len(made)][0] for i in range(count)]
print(f"{count} x {side}x{side} RGBA all-Paeth PNGs (generated in {time.time()-t0:.0f} s)", flush=True)
b = DevicePngBatch(pngs)
P = b.rgba_bytes


def check(label):
    res, ires = b.results()
    assert (res["good"] == 1).all() and (ires["good"] == 1).all(), label
    for i in sorted({0, 1 % count, 2 % count, 3 % count, count - 1}):
        assert np.array_equal(b.rgba(i), np.asarray(made[i % len(made)][1]).reshape(-1)), f"{label}: image {i} differs"
    b.d_rgba.zero_()


os.environ["DEBIG_PNG_PIPELINE"] = "0"
t, tm = timeit(b.launch)
check("two launches")
print(f"  inflate, then de-filter ({len(b.inflate.chunk_groups)} workspace groups)  {t:9.3f} ms (min {tm:.3f})  {P/t/1e6:8.1f} GB/s of RGBA", flush=True)
t, tm = timeit(b.launch_inflate_only)
print(f"  inflate only                                   {t:9.3f} ms (min {tm:.3f})", flush=True)
t, tm = timeit(b.launch_defilter_only)
print(f"  de-filter only                                 {t:9.3f} ms (min {tm:.3f})", flush=True)
os.environ["DEBIG_PNG_PIPELINE"] = "1"
for g, ln in groups:
    t, tm = timeit(lambda: b.launch_pipelined(group_images=g, lanes=ln))
    check(f"pipelined {g} x {ln}")
    print(f"  pipelined, groups of {g:3d} images ({len(b.inflate.chunk_groups)} groups), {ln} inflate lanes  {t:9.3f} ms (min {tm:.3f})  {P/t/1e6:8.1f} GB/s of RGBA; pixels of 5 images checked", flush=True)
t, tm = timeit(b.launch)
check("launch()")
print(f"  launch() default                               {t:9.3f} ms (min {tm:.3f})  {P/t/1e6:8.1f} GB/s of RGBA", flush=True)
