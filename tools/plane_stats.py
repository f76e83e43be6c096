#!/usr/bin/env python3
"""Diagnostic: in the chunk-parallel path every task k >= 1 is replayed in 16-bit elements {a, b}
(synthetic histories A and B); an element with a != b derives from the 32 KiB window in front of
the task.  How far into a task's output do such bytes reach?  (If they die out, the second plane is
dead weight from there on.)   python tools/plane_stats.py KIND COUNT MBYTES_EACH"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload, _native as N
from debigulator_amd.batch import DeviceBatch
from debigulator_amd.png_device import split_png

kind, count, mbytes = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
raws, caps = [], []
for s in range(count):
    if kind == "png":
        side = int((mbytes * 1e6 / 4) ** 0.5) // 64 * 64
        png, _ = workload.make_png(9000 + s, side, side, ct=6, ftype=4, noise=workload.CFG4_NOISE, enc="dynamic")
        raw = split_png(png)["raw"]
        raws.append(raw); caps.append(side * (side * 4 + 1))
    elif kind.endswith(".png"):
        it = split_png(open(kind, "rb").read())
        raws.append(it["raw"]); caps.append(4 * it["w"] * it["h"] + it["h"] + 1)
    else:
        r, p = workload.make_stream(kind, 100 + s, size=int(mbytes * 1e6))
        raws.append(bytes(r)); caps.append(len(p))
b = DeviceBatch.from_streams(raws, caps)
b.launch(waves_per_stream=N.WAVES_CHUNKED)
res = b.results()
assert (res["good"] == 1).all()
ws = b.d_ws_chunked.cpu().numpy()
n = count
hdr, h64 = ws[:128].view(np.uint32), ws[:128].view(np.uint64)
n_tasks, max_tasks, planes_off = int(hdr[0]), int(hdr[2]), int(h64[6])
SD = np.dtype([("first", "<u4"), ("n", "<u4"), ("state", "<u4"), ("final", "<u4"), ("total", "<u8"), ("end_bit", "<u8"),
               ("need", "<u8"), ("base", "<u8"), ("nb", "<u4"), ("nw", "<u4"), ("nr", "<u4"), ("bad", "<u4")])
TD = np.dtype([("stream", "<u4"), ("k", "<u4"), ("found", "<u8"), ("start", "<u8"), ("stop", "<u8"), ("out_off", "<u8"),
               ("plane_rel", "<u8"), ("live", "<u4"), ("rescan", "<u4"), ("pad1", "<u8")])
SL = np.dtype([("row0", "<u8"), ("rows", "<u4"), ("recs", "<u4"), ("rec0", "<u8"), ("state", "<u4"), ("flags", "<u4"),
               ("out_total", "<u8"), ("end_bit", "<u8"), ("nb", "<u4"), ("nw", "<u4"), ("nr", "<u4"), ("pad", "<u4")])
al = lambda v: (v + 255) // 256 * 256
t_off = al(128 + n * 64)
s_off = t_off + max_tasks * 64
cs = ws[128:128 + n * 64].view(SD)
tk = ws[t_off:t_off + max_tasks * 64].view(TD)
sl = ws[s_off:s_off + max_tasks * 64].view(SL)
H = 32768
tot_bytes = tot_diff = 0
lasts, sizes, dead = [], [], 0
for i in range(n):
    c = cs[i]
    for k in range(1, int(c["final"]) + 1):
        t, s = tk[c["first"] + k], sl[c["first"] + k]
        if not t["live"]:
            continue
        tot = int(s["out_total"])
        blk = planes_off + int(c["base"]) + int(t["plane_rel"])
        pw = blk + H + 2 * H  # the true window, then the wide plane: H synthetic elements, then the task's
        el = ws[pw:pw + 2 * tot].view("<u2")
        A, B = el & 255, el >> 8
        d = np.flatnonzero(A != B)
        tot_bytes += tot
        tot_diff += d.size
        last = int(d[-1]) + 1 if d.size else 0
        lasts.append(last); sizes.append(tot)
        dead += max(0, tot - (last + H))  # bytes behind a clean 32 KiB stretch after the last differing byte
lasts, sizes = np.array(lasts), np.array(sizes)
print(f"{kind}: {len(lasts)} tasks behind a first task, output per task {sizes.mean()/1e3:.0f} KB (min {sizes.min()/1e3:.0f}, max {sizes.max()/1e3:.0f})")
print(f"  bytes that derive from the window in front of the task: {100*tot_diff/tot_bytes:.2f} % of the task output")
print(f"  last such byte: median {np.median(lasts)/1e3:.1f} KB into the task, 90 % {np.percentile(lasts,90)/1e3:.1f} KB, max {lasts.max()/1e3:.1f} KB")
print(f"  output that lies more than 32 KiB behind the last such byte (a second plane adds nothing there): {100*dead/tot_bytes:.1f} %")
