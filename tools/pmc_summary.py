#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs of bench.py runs: per launch kind (stored / fixed-Huffman,
told apart by launch order: bench.py launches stored then fixed each step)."""
import csv, glob, sys, collections
out = {}
for name in ("fetch", "write"):
    files = glob.glob(f"gpurun_out/pmc_traffic_{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    rows = [r for r in csv.DictReader(open(files[0])) if "debig_inflate_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    # warmup(1) + steps(3) => 8 launches: even index = stored, odd = fixed
    st, fx = vals[0::2], vals[1::2]
    out[name] = (sum(st[-3:]) / 3, sum(fx[-3:]) / 3, rows[0]["Counter_Name"])
for name, (st, fx, cn) in out.items():
    print(f"{cn}: stored launch {st:.1f} KB, fixed-Huffman launch {fx:.1f} KB (per launch, raw counter)")
if "fetch" in out and "write" in out:
    for i, kind in enumerate(("stored", "fixed")):
        f, w = out["fetch"][i], out["write"][i]
        # gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM)
        print(f"{kind}: HBM traffic ~= 2*FETCH + WRITE = {(2*f + w) * 1024 / 1e6:.1f} MB per launch "
              f"(FETCH {f*1024/1e6:.1f} MB raw, WRITE {w*1024/1e6:.1f} MB)")
