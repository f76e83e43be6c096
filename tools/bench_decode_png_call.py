#!/usr/bin/env python3
"""The drop-in SINGLE call: decode_png() of the reference's header (host buffer in, host buffer out) on each sample file,
wall time of the C call (container walk, CRC check, H2D, inflate, de-filter, D2H) and its digest against tests/golden.
usage: bench_decode_png_call.py [DEBIG_NO_ROWS_HINT=1 in the environment: without the image-rows hint]"""
import glob, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from debigulator_amd import api

gold = json.load(open(os.path.join(ROOT, "tests", "golden", "resources.json")))["png"]
api.decode_png_init()
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "resources", "*.png"))):
    name = os.path.basename(f)
    if name == "backgrounddetailed1.png":
        continue  # (colour type 2: the reference's output depends on the caller's prior buffer)
    data = open(f, "rb").read()
    ts = []
    for it in range(5):
        t0 = time.perf_counter()
        good, rgba = api.decode_png(data)
        ts.append(time.perf_counter() - t0)
    ok = good == 1 and hashlib.sha256(bytes(rgba)).hexdigest() == gold[name]["rgba_sha256"]
    print(f"decode_png({name:26s}) {min(ts[1:])*1e3:8.3f} ms   {len(data):8d} B in, {len(rgba):8d} B out   digest {'ok' if ok else 'DIFFERS'}", flush=True)
