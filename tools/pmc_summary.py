#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs made by tools/pmc_traffic.sh (bench.py: every launch of
debig_inflate_kernel is the same batch) and write profiles/pmc_traffic.json for bench.py."""
import csv, glob, json, os
out = {}
for name in ("fetch", "write"):
    files = glob.glob(f"gpurun_out/pmc_traffic_{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    f = max(files, key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "debig_inflate_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    out[name] = (sum(vals[-3:]) / 3, rows[0]["Counter_Name"], len(vals))
for name, (v, cn, n) in out.items():
    print(f"{cn}: {v:.1f} KB per launch (raw counter, mean of the last 3 of {n} launches)")
if "fetch" in out and "write" in out:
    f, w = out["fetch"][0], out["write"][0]
    b = (2 * f + w) * 1024
    # gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM)
    print(f"HBM traffic ~= 2*FETCH + WRITE = {b/1e6:.1f} MB per launch (FETCH {f*1024/1e6:.1f} MB raw, WRITE {w*1024/1e6:.1f} MB)")
    json.dump({"bytes_per_launch": b, "fetch_size_kb_raw": f, "write_size_kb": w,
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py "
                      "(tools/pmc_traffic.sh); bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 reports half of a wide "
                      "coalesced read stream (MI355X_MICROARCH.md, HBM); the LZ77 history reads are 4-byte L2 loads, for "
                      "which the factor 2 is uncalibrated (upper estimate)"},
              open("profiles/pmc_traffic.json", "w"), indent=1)
