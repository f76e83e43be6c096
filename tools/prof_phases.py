#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the inflate kernel (needs the
-DDEBIG_PROFILE build: python tools/prof_phases.py builds it).  Shares only -- the
instrumented build is slower than the product build; never quote its run time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from debigulator_amd.build import build  # noqa: E402

_pre = os.path.join(ROOT, "debigulator_amd", "lib", "libdebigulator_hip_prof0.so")  # (built here by tools/prof_split_png_file.py PROF_BUILD=1: travels to the GPU box)
lib = _pre if os.path.exists(_pre) and not os.environ.get("PROF_REBUILD") else build(extra_defs=("DEBIG_PROFILE",), out="libdebigulator_hip_prof.so")
os.environ["DEBIG_LIB"] = lib
import numpy as np  # noqa: E402
import torch  # noqa: E402
from debigulator_amd import workload  # noqa: E402
from debigulator_amd.batch import DeviceBatch  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "fixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # wavefronts per stream (1, 2, 4)
size = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
pairs = workload.make_streams(kind, n, size)
raws = [p[0] for p in pairs]
caps = [max(size + 1, len(r)) for r in raws]
b = DeviceBatch.from_streams(raws, caps)
for _ in range(3):
    b.launch(waves_per_stream=waves)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); b.launch(waves_per_stream=waves); e1.record(); torch.cuda.synchronize()
res = b.results()
assert (res["good"] == 1).all()
prof = res["prof"].astype(np.float64) * 16
names = ["stage", "pass1 scan", "pass2 decode", "resolve(near)", "flush", "hdr+tables", "TOTAL", "far copy"]
tot = prof[:, 6].mean()
print(f"{kind}: {n} streams x {size} B, {waves} wavefront(s)/stream, kernel {e0.elapsed_time(e1):.3f} ms (instrumented), "
      f"windows/stream {res['n_windows'].mean():.2f}, rounds/window {res['n_rounds'].sum()/max(1,res['n_windows'].sum()):.2f}")
for i, nm in [(j, names[j]) for j in (0, 1, 2, 7, 3, 4, 5, 6)]:
    print(f"  {nm:14s} {prof[:, i].mean():12.0f} cyc/stream  {100*prof[:, i].mean()/tot:5.1f} %")
print(f"  other          {tot - prof[:, :6].mean(0).sum() - prof[:, 7].mean():12.0f} cyc/stream")
