#!/usr/bin/env python3
"""Diagnostic: what the chunk-parallel path (DEBIG_WAVES_CHUNKED) made of a batch -- reads the
workspace tables back after a launch (layout: csrc/inflate_chunk_kernel.inc).
    python tools/dump_chunked.py KIND COUNT MBYTES_EACH"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload, _native as N
from debigulator_amd.batch import DeviceBatch
from debigulator_amd.png_device import split_png

kind, count, mbytes = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
raws, caps = [], []
for s in range(min(count, 4)):
    if kind == "png":
        side = int((mbytes * 1e6 / 4) ** 0.5) // 64 * 64
        png, _ = workload.make_png(9000 + s, side, side, ct=6, ftype=4, noise=workload.CFG4_NOISE, enc="dynamic")
        raw = split_png(png)["raw"]
        raws.append(raw); caps.append(side * (side * 4 + 1))
    else:
        r, p = workload.make_stream(kind, 100 + s, size=int(mbytes * 1e6))
        raws.append(bytes(r)); caps.append(len(p))
raws = [raws[i % len(raws)] for i in range(count)]; caps = [caps[i % len(caps)] for i in range(count)]
b = DeviceBatch.from_streams(raws, caps)
b.launch(waves_per_stream=N.WAVES_CHUNKED)
res = b.results()
ws = b.d_ws_chunked.cpu().numpy()
n = count
hdr = ws[:128].view(np.uint32)
h64 = ws[:128].view(np.uint64)
n_tasks, C, max_tasks = int(hdr[0]), int(hdr[1]), int(hdr[2])
print(f"tasks {n_tasks} of {max_tasks}, C {C}, rows {h64[2]} recs {h64[3]} planes {h64[7] / 1e6:.1f} MB; ws {ws.size / 1e6:.1f} MB")
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
NONE = (1 << 64) - 1
for i in range(n):
    c = cs[i]
    print(f"stream {i}: in {len(raws[i])} tasks {c['first']}+{c['n']} state {c['state']} final {c['final']} total {c['total']} bad {c['bad']} "
          f"need {c['need'] / 1e6:.1f} MB | result good {res[i]['good']} status {res[i]['status']}")
    shown = 0
    for k in range(int(c["n"])):
        t, s = tk[c["first"] + k], sl[c["first"] + k]
        empty = t["start"] == t["stop"]
        odd = (not empty) and (s["state"] != 0 or not (s["flags"] & 3))
        if (not empty and shown < 4) or odd:
            f = lambda v: -1 if v == NONE else int(v)
            print(f"   task {k}: found {f(t['found'])} start {f(t['start'])} stop {f(t['stop'])} live {t['live']} | state {s['state']} flags {s['flags']} "
                  f"out {s['out_total']} end {f(s['end_bit'])} rows {s['rows']} recs {s['recs']}")
            shown += 1
    live = sum(1 for k in range(int(c["n"])) if tk[c["first"] + k]["start"] != tk[c["first"] + k]["stop"])
    print(f"   non-empty tasks: {live}")
