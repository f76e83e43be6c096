#!/usr/bin/env python3
"""Secondary measurement: CRC-32 / Adler-32 kernel over 4096 x 64 KiB spans resident in HBM
(HBM-bound: reads every byte once)."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd.checksum import DeviceChecksums, CRC32, ADLER32

n, size = 4096, 65536
if len(sys.argv) > 2:
    n, size = int(sys.argv[1]), int(sys.argv[2])
arena = torch.randint(0, 256, (n * size + 64,), dtype=torch.uint8, device="cuda")
host = arena.cpu().numpy()
for kind, name, fn in ((CRC32, "crc32", zlib.crc32), (ADLER32, "adler32", zlib.adler32)):
    ck = DeviceChecksums(arena, [(i * size + (i % 7), size - 13) for i in range(n)], kind)
    for _ in range(3): ck.launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ck.launch(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    res = ck.results()
    ok = all(res[i] == fn(host[i * size + (i % 7): i * size + (i % 7) + size - 13].tobytes()) for i in range(0, n, 97))
    ms = float(np.median(ts))
    print(f"{name:8s}: {n} spans x {size-13} B, {ms:.3f} ms, {ck.bytes/ms/1e6:8.1f} GB/s read = {ck.bytes/ms/1e6/8000*100:5.1f} % of 8 TB/s, exact={ok}")
