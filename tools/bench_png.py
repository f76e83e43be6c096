#!/usr/bin/env python3
"""Secondary measurements (not bench.py's headline): PNG decode with everything resident in
HBM -- BASELINE config 3 (1024 PNGs cycled from the reference's sample files) and the config 4
shape (all-Paeth RGBA 8192x8192, a few images instead of 256)."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.png_device import DevicePngBatch


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if which == "cfg3":
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "resources", "*.png")))
             if not f.endswith("backgrounddetailed1.png")]
    datas = [open(f, "rb").read() for f in files]
    pngs = [datas[i % len(datas)] for i in range(1024)]
    label = "cfg3: 1024 PNGs cycled from 14 reference sample files"
else:
    side = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    t0 = time.time()
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:  # BASELINE config 4: >= 4 distinct seeds, ratio about 3:1
        made = list(ex.map(lambda s: workload.make_png(9000 + s, side, side, ct=6, ftype=4, noise=workload.CFG4_NOISE,
                                                       enc="dynamic", idat_chunk=65536), range(min(count, 4))))
    distinct = [m[0] for m in made]
    pixels = [m[1] for m in made]  # what must come out: [h, w*4] bytes
    pngs = [distinct[i % len(distinct)] for i in range(count)]
    label = f"cfg4 shape: {count} x {side}x{side} RGBA all-Paeth PNGs (generated in {time.time()-t0:.0f} s)"
b = DevicePngBatch(pngs)
t_all = timeit(lambda: b.launch(fused=False))
t_inf = timeit(b.launch_inflate_only)
t_def = timeit(b.launch_defilter_only)
res, ires = b.results()
assert (res["good"] == 1).all() and (ires["good"] == 1).all()
t_fused = None
if os.environ.get("DEBIG_BENCH_FUSED", "1") != "0" and len(pngs) <= 4096:  # one kernel per image batch
    b.launch(fused=False)
    torch.cuda.synchronize()
    want = [b.rgba(i).copy() for i in sorted(set([0, 1 % len(pngs), 5 % len(pngs), 13 % len(pngs), len(pngs) - 1]))]
    b.d_rgba.zero_()
    t_fused = timeit(b.launch_fused)
    rf, irf = b.results()
    assert (rf["good"] == 1).all() and (irf["good"] == 1).all()
    for k, i in enumerate(sorted(set([0, 1 % len(pngs), 5 % len(pngs), 13 % len(pngs), len(pngs) - 1]))):
        assert np.array_equal(b.rgba(i), want[k]), f"fused: image {i} differs"
checked = ""
if which != "cfg3":  # the generator's own pixels are the expected output (round-trip property)
    b.launch(fused=False)
    for i in sorted(set([0, 1 % count, 2 % count, 3 % count, count - 1])):
        assert np.array_equal(b.rgba(i), np.asarray(pixels[i % len(pixels)]).reshape(-1)), f"image {i} differs"
    checked = "; images 0-3 and last byte-exact vs the generator's pixels"
P, Cb, Sb = b.rgba_bytes, b.c_bytes, b.s_bytes
print(label + checked)
print(f"  two launches     {t_all:9.3f} ms  {P/t_all/1e6:8.1f} GB/s of RGBA   (C={Cb/1e6:.1f} MB, S={Sb/1e6:.1f} MB, P={P/1e6:.1f} MB)")
print(f"  inflate only     {t_inf:9.3f} ms  {Sb/t_inf/1e6:8.1f} GB/s of scanline stream")
print(f"  de-filter only   {t_def:9.3f} ms  {(Sb+P)/t_def/1e6:8.1f} GB/s (S+P)")
t_def_l = timeit(b.launch)
print(f"  launch() default {t_def_l:9.3f} ms  {P/t_def_l/1e6:8.1f} GB/s of RGBA   (what DevicePngBatch picks for this batch)")
if t_fused is not None:
    print(f"  fused kernel     {t_fused:9.3f} ms  {P/t_fused/1e6:8.1f} GB/s of RGBA   (debig_hip_png_decode_fused_batch, pixels of 5 images compared)")
