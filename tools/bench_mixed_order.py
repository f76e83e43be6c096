#!/usr/bin/env python3
"""Diagnostic: config 2's batch (4096 fixed-Huffman + 4096 stored streams of 64 KiB) in different stream
ORDERS inside the batch: all fixed then all stored (what bench.py times), alternating, alternating in
groups of 8 / 64 / 512, stored first -- one workgroup per stream (DEBIG_WAVES_SPLIT) against persistent
workgroups with a work queue (DEBIG_WAVES_SPLIT_QUEUED)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.batch import DeviceBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fx = workload.make_streams("fixed", n, 65536)
st = workload.make_streams("stored", n, 65536, first=n)
def run(label, seq):
    raws = [p[0] for p in seq]; caps = [max(65537, len(r)) for r in raws]
    b = DeviceBatch.from_streams(raws, caps)
    out = []
    for width in (0x10, 0x11):  # DEBIG_WAVES_SPLIT (one workgroup per stream), DEBIG_WAVES_SPLIT_QUEUED
        for _ in range(3): b.launch(waves_per_stream=width)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); b.launch(waves_per_stream=width); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res = b.results()
        ok = bool((res["good"] == 1).all()) and all(b.output(i, res) == seq[i][1].tobytes() for i in (0, 1, len(seq) // 2, len(seq) - 1))
        out.append((float(np.median(ts)), ok))
    print(f"{label:34s} per-stream workgroups {out[0][0]:7.3f} ms   queued {out[1][0]:7.3f} ms   exact={out[0][1] and out[1][1]}", flush=True)
tiny = workload.make_streams("stored", n, 256, first=3 * n)
def alt(a, b, g):
    seq = []
    for i in range(0, n, g): seq += a[i:i + g] + b[i:i + g]
    return seq
if len(sys.argv) > 2:  # what is it about mixing: the other kind's traffic, or where the workgroups land?
    run("fixed only", fx)
    run("fixed..., tiny stored...", fx + tiny)
    run("fixed / tiny stored alternating", alt(fx, tiny, 1))
    run("fixed / fixed alternating", alt(fx[: n // 2], fx[n // 2:], 1))
    dy = workload.make_streams("dynamic", n, 65536, first=5 * n)
    run("fixed..., dynamic...", fx + dy)
    run("fixed / dynamic alternating", alt(fx, dy, 1))
    sys.exit(0)
run("fixed..., stored...", fx + st)
run("stored..., fixed...", st + fx)
for g in (1, 8, 64, 512):
    seq = []
    for i in range(0, n, g): seq += fx[i:i + g] + st[i:i + g]
    run(f"alternating groups of {g}", seq)
