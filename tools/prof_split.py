#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the scan / LZ77 kernel pair (DEBIG_WAVES_SPLIT).
Builds two -DDEBIG_PROFILE libraries (one reports the scan kernel's phases, one the LZ77 kernel's)
and runs each in a child process.  Shares only: instrumented builds run 3 waves per SIMD and are
slower than the product; never quote their run time.
usage: prof_split.py [kind=fixed] [n=4096] [size=65536] [width=0x10 | 0x12]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.batch import DeviceBatch
kind, n, size, which, width = %(kind)r, %(n)d, %(size)d, %(which)d, %(width)d
pairs = workload.make_streams(kind, n, size, threads=16)
raws = [p[0] for p in pairs]; caps = [max(size + 1, len(r)) for r in raws]
b = DeviceBatch.from_streams(raws, caps)
for _ in range(3): b.launch(waves_per_stream=width)
torch.cuda.synchronize()
res = b.results(); assert (res["good"] == 1).all()
prof = res["prof"].astype(np.float64) * 16
names = (["-", "lead (positions)", "main pass (tokens)", "header + tables", "chain + re-decode + records", "-", "TOTAL", "-"] if which == 0 and width == 0x12 else
         ["stage window", "position rounds", "full rounds (tokens)", "header + tables", "window records", "-", "TOTAL", "-"] if which == 0 else
         ["token replay", "far copy", "near resolve", "flush", "-", "-", "TOTAL", "-"])
tot = prof[:, 6].mean()
print(f"{'scan kernel' if which == 0 else 'LZ77 kernel'}: {kind}, {n} streams x {size} B; windows/stream {res['n_windows'].mean():.2f}, "
      f"rounds/window {res['n_rounds'].sum()/max(1,res['n_windows'].sum()):.2f}")
acc = 0
for i, nm in enumerate(names):
    if nm == "-" or i == 6: continue
    acc += prof[:, i].mean()
    print(f"  {nm:22s} {prof[:, i].mean():12.0f} cyc/stream  {100*prof[:, i].mean()/tot:5.1f} %%")
print(f"  {'other':22s} {tot-acc:12.0f} cyc/stream  {100*(tot-acc)/tot:5.1f} %%")
print(f"  {'TOTAL':22s} {tot:12.0f} cyc/stream")
'''

if __name__ == "__main__":
    from debigulator_amd.build import build

    kind = sys.argv[1] if len(sys.argv) > 1 else "fixed"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    width = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0x10
    for which in (0, 1):
        lib = build(extra_defs=("DEBIG_PROFILE", f"DEBIG_PROFILE_LZ={which}"), out=f"libdebigulator_hip_prof{which}.so")
        env = dict(os.environ, DEBIG_LIB=lib)
        subprocess.check_call([sys.executable, "-c", CHILD % {"root": ROOT, "kind": kind, "n": n, "size": size, "which": which, "width": width}], env=env)
