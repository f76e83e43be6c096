#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs made by tools/pmc_traffic.sh (bench.py: every step is the
same batch) and write profiles/pmc_traffic.json for bench.py.  A step is several launches since
round 2 (workspace plan, scan kernel, LZ77 kernel, the one-kernel path for handed-back streams):
the bytes of all debig_* kernels between two plan launches are one step."""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out, per_kernel = {}, {}
for name in ("fetch", "write"):
    files = glob.glob(f"gpurun_out/pmc_traffic_{name}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    f = max(files, key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "debig_" in r["Kernel_Name"] and "tables" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # steps: split at every plan kernel (or every one-kernel launch when the plan kernel is absent)
    steps, cur = [], []
    # round 3: the workspace is carved once per batch object, a step is scan + lz + hand-back
    marker = next((m for m in ("debig_scanlz_kernel", "debig_scan_kernel", "debig_inflate_kernel")
                   if any(m in r["Kernel_Name"] for r in rows)), "debig_inflate_kernel")
    rows = [r for r in rows if "debig_split_plan_kernel" not in r["Kernel_Name"]]
    for r in rows:
        if marker in r["Kernel_Name"] and cur:
            steps.append(cur)
            cur = []
        cur.append(r)
    if cur:
        steps.append(cur)
    # (older runs also launched each stream kind alone, smaller grids): keep the whole-batch steps
    big = max(max(int(r["Grid_Size"]) for r in st) for st in steps)
    steps = [st for st in steps if max(int(r["Grid_Size"]) for r in st) == big]
    last = steps[-3:]
    out[name] = (sum(sum(float(r["Counter_Value"]) for r in st) for st in last) / len(last), rows[0]["Counter_Name"], len(steps))
    for r in last[-1]:
        k = r["Kernel_Name"].split("(")[0]
        per_kernel.setdefault(k, {})[name] = per_kernel.get(k, {}).get(name, 0.0) + float(r["Counter_Value"])
for name, (v, cn, n) in out.items():
    print(f"{cn}: {v:.1f} KB per step (raw counter, mean of the last 3 of {n} steps)")
for k, d in per_kernel.items():
    print(f"  {k:28s} FETCH_SIZE {d.get('fetch', 0):12.1f} KB raw   WRITE_SIZE {d.get('write', 0):12.1f} KB")
if "fetch" in out and "write" in out:
    f, w = out["fetch"][0], out["write"][0]
    # gfx950: FETCH_SIZE tallies 64 bytes per 128-byte line request, for coalesced streams (MI355X_MICROARCH.md, HBM)
    # and -- calibrated in round 3 with tools/ubench_fetch.hip, profiles/r03_fetch_calibration.txt -- for the LZ77
    # kernel's shapes too: 4 B/lane token rows are reported at 1/2, a random 16-byte history gather is one line
    # request and moves a whole 128-byte line.  HBM read bytes = 2 x FETCH_SIZE for every kernel.
    wide = f
    b = (2 * f + w) * 1024
    print(f"HBM traffic ~= 2 x FETCH + WRITE = {b/1e6:.1f} MB per step (FETCH {f*1024/1e6:.1f} MB raw, WRITE {w*1024/1e6:.1f} MB)")
    for k, d in per_kernel.items():
        print(f"  {k:28s} reads {2*d.get('fetch', 0)*1024/1e6:9.1f} MB  writes {d.get('write', 0)*1024/1e6:9.1f} MB")
    from bench import kernel_sources_digest
    json.dump({"bytes_per_launch": b, "fetch_size_kb_raw": f, "write_size_kb": w,
               "per_kernel_kb_raw": per_kernel, "kernel_sources_sha256": kernel_sources_digest(),
               "source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py (tools/pmc_traffic.sh)",
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py "
                      "(tools/pmc_traffic.sh), summed over the kernels of one whole-batch step; FETCH_SIZE doubled for every "
                      "kernel: gfx950 tallies 64 bytes per 128-byte line request (MI355X_MICROARCH.md, HBM), calibrated for the "
                      "LZ77 kernel's token rows and 16-byte history gathers with tools/ubench_fetch.hip "
                      "(profiles/r03_fetch_calibration.txt)"},
              open("profiles/pmc_traffic.json", "w"), indent=1)
