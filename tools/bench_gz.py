#!/usr/bin/env python3
"""Secondary measurement (not bench.py's headline): BASELINE config 5's per-GPU shard --
N gzip members of 1 MiB (text-like payload, dynamic Huffman, ratio about 3:1), resident in
HBM: member payloads are inflated in one launch and every member's CRC-32 trailer is checked
by the checksum kernel.  Usage: bench_gz.py [members=8192] [size=1048576] [width=0 (the library's choice) | 0x10 | 0x12]"""
import os, struct, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import workload
from debigulator_amd.batch import DeviceBatch
from debigulator_amd.checksum import DeviceChecksums, CRC32

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
width = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
uniq = min(n, 32)
pairs = workload.make_streams("dynamic", uniq, size)
members = [workload.gzip_member(r, p) for r, p in pairs]
# host side of decode_gz: fixed 10-byte header (no optional fields in these members), 8-byte trailer
raws = [members[i % uniq][10:-8] for i in range(n)]
want_crc = [struct.unpack("<I", members[i % uniq][-8:-4])[0] for i in range(n)]
assert want_crc[0] == zlib.crc32(pairs[0][1].tobytes())
caps = [max(size + 1, len(r)) for r in raws]
b = DeviceBatch.from_streams(raws, caps)
spans = [(int(b.streams_host[i]["out_off"]), size) for i in range(n)]
ck = DeviceChecksums(b.d_out, spans, CRC32)
comp = sum(len(r) for r in raws)


def step():
    b.launch(waves_per_stream=width)
    ck.launch()


for _ in range(2):
    step()
torch.cuda.synchronize()
ts, ti = [], []
for _ in range(5):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); b.launch(waves_per_stream=width); e1.record(); ck.launch(); e2.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e2)); ti.append(e0.elapsed_time(e1))
res = b.results()
ok = bool((res["good"] == 1).all() and (res["final_size"] == size).all())
crcs = ck.results()
ok_crc = bool((np.asarray(crcs, dtype=np.uint32) == np.asarray(want_crc, dtype=np.uint32)).all())
ok_bytes = all(b.output(i, res) == pairs[i % uniq][1].tobytes() for i in (0, n // 2, n - 1))
ms, mi = float(np.median(ts)), float(np.median(ti))
print(f"cfg5 shard (width {width:#x}): {n} gzip members x {size} B (compressed {comp/1e6:.1f} MB, ratio {n*size/comp:.2f})")
print(f"  inflate          {mi:9.3f} ms  {n*size/mi/1e6:8.1f} GB/s decompressed")
print(f"  inflate + CRC-32 {ms:9.3f} ms  {n*size/ms/1e6:8.1f} GB/s decompressed   sizes/good={ok} crc={ok_crc} bytes={ok_bytes}")
