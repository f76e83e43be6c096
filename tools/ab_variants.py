#!/usr/bin/env python3
"""Diagnostic: A/B several library builds (compile-time -D variants) in one GPU call.

  build here (no GPU needed):   python tools/ab_variants.py build  name=DEF1,DEF2=3 name2=...
  run on the GPU box:           python tools/ab_variants.py run [kinds=fixed,dynamic] [n=4096] [prof=1] [only=a,b]
                                (prof=1: each run under rocprofv3 --kernel-trace --stats, per-kernel averages printed)

`build` writes debigulator_amd/lib/libdebigulator_hip_ab_<name>.so (they travel with the snapshot);
`run` times every such library with tools/bench_variant.py in a child process each ("base" = the
product library).  Nothing here is on the product path."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "debigulator_amd", "lib")


def main():
    if len(sys.argv) < 2 or sys.argv[1] not in ("build", "run", "clean"):
        print(__doc__)
        return 2
    if sys.argv[1] == "clean":
        for f in glob.glob(os.path.join(LIBDIR, "libdebigulator_hip_ab_*.so")):
            os.remove(f)
        return 0
    if sys.argv[1] == "build":
        from debigulator_amd.build import build
        for spec in sys.argv[2:]:
            name, _, defs = spec.partition("=")
            defs = tuple(d for d in defs.split(",") if d)
            print(name, defs, build(force=True, extra_defs=defs, out=f"libdebigulator_hip_ab_{name}.so"), flush=True)
        return 0
    kinds, n, width, prof, only = ["fixed"], "4096", "0x10", False, None
    for a in sys.argv[2:]:
        if a == "prof=1":
            prof = True
        if a.startswith("only="):
            only = a[5:].split(",")
        if a.startswith("kinds="):
            kinds = a[6:].split(",")
        elif a.startswith("n="):
            n = a[2:]
        elif a.startswith("width="):
            width = a[6:]
    libs = [("base", None)] + [(os.path.basename(f)[len("libdebigulator_hip_ab_"):-3], f)
                               for f in sorted(glob.glob(os.path.join(LIBDIR, "libdebigulator_hip_ab_*.so")))]
    if only:
        libs = [x for x in libs if x[0] in only]
    for kind in kinds:
        for name, lib in libs:
            env = dict(os.environ)
            if lib:
                env["DEBIG_LIB"] = lib
            cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_variant.py"), kind, n, width]
            pdir = f"/tmp/ab_prof/{kind}_{name}_{n}_{os.getpid()}"
            if prof:
                env["TMPDIR"] = "/tmp"
                cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", pdir, "--"] + cmd
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd="/tmp" if prof else None)
            out = [l for l in r.stdout.splitlines() if "GB/s" in l or "rror" in l or "ssert" in l]
            print(f"{name:24s} {out[-1] if out else r.stdout[-300:]}", flush=True)
            if prof:
                import csv
                for f in glob.glob(pdir + "/**/*kernel_stats.csv", recursive=True):
                    for row in csv.DictReader(open(f)):
                        nm = row.get("Name", "")
                        if "debig" in nm:
                            print(f"    {nm.split('(')[0][:40]:40s} calls {row.get('Calls'):>4s} avg {float(row.get('AverageNs', 0))/1e3:9.1f} us"
                                  f"  min {float(row.get('MinNs', 0))/1e3:9.1f}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
