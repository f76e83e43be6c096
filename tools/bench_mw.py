#!/usr/bin/env python3
"""Diagnostic: wavefronts-per-stream sweep.  For each (kind, stream size, batch size) time
debig_hip_inflate_batch_ex with 1, 2 and 4 wavefronts per stream and check the output bytes.
Usage: bench_mw.py [kinds=fixed,dynamic,png] [sizes=65536,1048576] [counts=1,16,64,256,1024,4096]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.batch import DeviceBatch

kinds = (sys.argv[1] if len(sys.argv) > 1 else "fixed,dynamic,png").split(",")
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "65536,1048576").split(",")]
counts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,16,64,256,1024,4096").split(",")]
MAX_BYTES = 512 << 20
WIDTHS = (1, 2, 4, 8)
print(f"{'kind':8s} {'size':>9s} {'n':>5s} " + " ".join(f"{'w=%d GB/s' % w:>11s}" for w in WIDTHS), flush=True)
for kind in ([] if "mixed" in sys.argv else kinds):
    for size in sizes:
        nmax = max(c for c in counts if c * size <= MAX_BYTES)
        uniq = min(nmax, 64)  # distinct streams; larger batches repeat them
        pairs = workload.make_streams(kind, uniq, size)
        for n in counts:
            if n * size > MAX_BYTES:
                continue
            raws = [pairs[i % uniq][0] for i in range(n)]
            caps = [max(size + 1, len(r)) for r in raws]
            b = DeviceBatch.from_streams(raws, caps)
            row = []
            for w in WIDTHS:
                b.d_out.zero_()
                for _ in range(2):
                    b.launch(waves_per_stream=w)
                torch.cuda.synchronize()
                ts = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); b.launch(waves_per_stream=w); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                res = b.results()
                ok = bool((res["good"] == 1).all() and (res["final_size"] == size).all())
                for i in sorted(set([0, n // 2, n - 1])):
                    ok = ok and b.output(i, res) == pairs[i % uniq][1].tobytes()
                ms = float(np.median(ts))
                row.append(f"{n * size / ms / 1e6:9.2f}{'' if ok else '!!'}")
            print(f"{kind:8s} {size:9d} {n:5d} " + " ".join(f"{r:>11s}" for r in row), flush=True)
            del b

# a few large streams among thousands of small ones: the shape the mixed modes are for
if "mixed" in sys.argv:
    print("mixed shape: n_large x 4 MiB + 2048 x 64 KiB (dynamic); ms per launch", flush=True)
    small = workload.make_streams("dynamic", 64, 65536)
    large = workload.make_streams("dynamic", 8, 4 << 20)
    for n_large in (0, 8, 32, 128):
        pairs = [large[i % 8] for i in range(n_large)] + [small[i % 64] for i in range(2048)]
        raws = [p[0] for p in pairs]
        caps = [len(p[1]) + 1 for p in pairs]
        b = DeviceBatch.from_streams(raws, caps)
        row = []
        for w in (1, 2, 4, 0x41, 0x42):
            for _ in range(2):
                b.launch(waves_per_stream=w)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); b.launch(waves_per_stream=w); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res = b.results()
            ok = bool((res["good"] == 1).all()) and b.output(0, res) == pairs[0][1].tobytes() \
                and b.output(len(pairs) - 1, res) == pairs[-1][1].tobytes()
            row.append(f"w={w:#x}: {float(np.median(ts)):7.2f}{'' if ok else '!!'}")
        print(f"  n_large={n_large:4d}  " + "  ".join(row), flush=True)
        del b
