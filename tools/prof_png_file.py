#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the inflate kernel on ONE sample PNG's stream
(-DDEBIG_PROFILE build).  Usage: prof_png_file.py <file.png> [waves=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from debigulator_amd.build import build
lib = build(extra_defs=("DEBIG_PROFILE",), out="libdebigulator_hip_prof.so")
os.environ["DEBIG_LIB"] = lib
import numpy as np, torch
from debigulator_amd.png_device import DevicePngBatch
f = sys.argv[1]
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1
b = DevicePngBatch([open(f, "rb").read()])
for _ in range(2):
    b.inflate.launch(waves_per_stream=w)
r = b.inflate.results()[0]
prof = r["prof"].astype(np.float64) * 16
names = ["stage", "pass1 scan", "pass2 decode", "resolve(near)", "flush", "hdr+tables", "TOTAL", "far copy"]
print(f"{os.path.basename(f)}: {w} wavefront(s), blocks {r['n_blocks']}, windows {r['n_windows']}, rounds/window {r['n_rounds']/max(1,r['n_windows']):.2f}")
for i in (0, 1, 2, 7, 3, 4, 5, 6):
    print(f"  {names[i]:14s} {prof[i]:12.0f} cyc  {100*prof[i]/prof[6]:5.1f} %   per block {prof[i]/max(1,r['n_blocks']):9.0f}")
print(f"  other          {prof[6]-prof[:6].sum()-prof[7]:12.0f} cyc")
