#!/usr/bin/env python3
"""Per-kernel HBM traffic of a decode_png batch from the CSVs of tools/pmc_png.sh: bytes of the LAST
launch of every debig_* kernel (FETCH_SIZE raw: gfx950 reports half of wide coalesced reads)."""
import csv, glob, os
per = {}
for name in ("fetch", "write"):
    files = glob.glob(f"gpurun_out/pmc_png_{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    f = max(files, key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "debig_" in r["Kernel_Name"] and "tables" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    last = {}
    for r in rows:
        last[r["Kernel_Name"].split("(")[0]] = float(r["Counter_Value"])  # KB
    for k, v in last.items():
        per.setdefault(k, {})[name] = v
tot_f = sum(d.get("fetch", 0) for d in per.values())
tot_w = sum(d.get("write", 0) for d in per.values())
for k, d in sorted(per.items(), key=lambda kv: -(kv[1].get("fetch", 0) + kv[1].get("write", 0))):
    print(f"  {k:36s} FETCH_SIZE {d.get('fetch', 0) / 1e3:10.1f} MB raw   WRITE_SIZE {d.get('write', 0) / 1e3:10.1f} MB")
print(f"  {'all debig_* kernels of one decode':36s} FETCH_SIZE {tot_f / 1e3:10.1f} MB raw   WRITE_SIZE {tot_w / 1e3:10.1f} MB")
