import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from debigulator_amd.batch import DeviceBatch
from debigulator_amd.png_device import split_png
which = %(which)d
it = split_png(open(%(f)r, "rb").read())
raw = it["raw"]; est = 4*it["w"]*it["h"] + it["h"] + 1
NCOPY = int(os.environ.get("PROF_COPIES", "64")); raws=[raw]*NCOPY; caps=[est]*NCOPY
b = DeviceBatch.from_streams(raws, caps)
import ctypes
ws = torch.empty(20*len(raw)*64 + (1<<26), dtype=torch.uint8, device="cuda")
b.d_ws = ws
for _ in range(2): b.launch(waves_per_stream=int(os.environ.get("PROF_WIDTH", "0x10"), 0))
torch.cuda.synchronize()
res = b.results(); print("good", int(res["good"].sum()), "size", int(res["final_size"][0]), "C", len(raw))
prof = res["prof"].astype(np.float64) * 16
names = (["stage window", "position rounds", "full rounds (tokens)", "header + tables", "window records", "-", "TOTAL", "-"] if which == 0 else
         ["token replay", "far copy", "near resolve", "near: group set-up", "near: dependency search", "near: one-at-a-time copies (inside rounds)", "TOTAL", "near: rounds"])
tot = prof[:, 6].mean()
print(("scan" if which==0 else "LZ77"), "blocks", res["n_blocks"][0], "windows", res["n_windows"][0])
for i, nm in enumerate(names):
    if nm == "-" or i == 6: continue
    print(f"  {nm:22s} {prof[:, i].mean():12.0f} cyc/stream  {100*prof[:, i].mean()/tot:5.1f} %%")
print(f"  TOTAL {tot:12.0f} cyc/stream = {tot/int(res['final_size'][0]):.1f} cyc/byte")
'''
from debigulator_amd.build import build
LIBDIR = os.path.join(ROOT, "debigulator_amd", "lib")
libs = [os.path.join(LIBDIR, "libdebigulator_hip_prof%d.so" % w) for w in (0, 1)]
if os.environ.get("PROF_BUILD") or not all(os.path.exists(l) for l in libs):  # (built here, the libraries travel to the GPU box)
    libs = [build(extra_defs=("DEBIG_PROFILE", "DEBIG_PROFILE_LZ=%d" % w), out="libdebigulator_hip_prof%d.so" % w) for w in (0, 1)]
if os.environ.get("PROF_BUILD"):
    sys.exit(0)
for f in sys.argv[1:]:
    for which in (0,1):
        env = dict(os.environ, DEBIG_LIB=libs[which])
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "which": which, "f": f}], env=env, capture_output=True, text=True)
        print(p.stdout[-1500:], p.stderr[-600:] if p.returncode else "")
