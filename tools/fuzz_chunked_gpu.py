#!/usr/bin/env python3
"""Stress: large random-structure streams through the chunk-parallel path on the GPU, every output
compared with zlib's (valid streams, recipients of the exact size).  The payloads carry what the
block finder can stumble over: compressed data embedded as literals (real dynamic block headers at
byte-aligned and, after a stored-block header, arbitrary positions), long runs, incompressible noise.
Every stream is then damaged (flipped bits, cut short) and the chunk path compared with a whole
workgroup per stream: status, size, CRC of the decoded prefix.
    python tools/fuzz_chunked_gpu.py [SECONDS=240] [SEED=1]"""
import os, random, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debigulator_amd import _native as N
from debigulator_amd.batch import DeviceBatch

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def text(rng, n):
    words = [bytes(rng.getrandbits(8) & 0x7f | 0x20 for _ in range(rng.randint(2, 10))) for _ in range(300)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words) + b" "
    return bytes(out[:n])


def make(rng, nprng):
    parts = []
    for _ in range(rng.randint(3, 10)):
        n = rng.randint(20000, 1500000)
        k = rng.random()
        if k < 0.35:
            parts.append(text(rng, n))
        elif k < 0.5:
            parts.append(nprng.integers(0, 256, n, dtype=np.uint8).tobytes())
        elif k < 0.6:
            parts.append(bytes([rng.randrange(256)]) * n)
        elif k < 0.75:  # compressed data as payload: real block headers inside what becomes stored blocks
            parts.append(zlib.compress(text(rng, n), rng.choice([1, 6, 9]))[2:-4])
        else:
            rec = bytearray(rng.getrandbits(8) for _ in range(rng.randint(3, 40)))
            b = bytearray()
            while len(b) < n:
                rec[rng.randrange(len(rec))] = rng.getrandbits(8)
                b += rec
            parts.append(bytes(b[:n]))
    c = zlib.compressobj(rng.choice([1, 3, 6, 9]), zlib.DEFLATED, -15, rng.choice([8, 9]),
                         rng.choice([zlib.Z_DEFAULT_STRATEGY] * 5 + [zlib.Z_FILTERED, zlib.Z_RLE]))
    raw = b""
    for p in parts:
        raw += c.compress(p)
        if rng.random() < 0.25:
            raw += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
    raw += c.flush()
    return raw, b"".join(parts)


t0, rounds, streams, nbytes = time.time(), 0, 0, 0
seed = seed0
while time.time() - t0 < budget:
    rng, nprng = random.Random(seed), np.random.default_rng(seed)
    pairs = [make(rng, nprng) for _ in range(24)]
    raws = [p[0] for p in pairs]
    caps = [max(len(p[1]), len(p[0])) for p in pairs]  # the reference refuses a recipient smaller than the input
    for chunk in (None, "32768", "8192"):
        if chunk is None:
            os.environ.pop("DEBIG_CHUNK_BYTES", None)
        else:
            os.environ["DEBIG_CHUNK_BYTES"] = chunk
        b = DeviceBatch.from_streams(raws, caps, out_skew=seed % 13)
        b.launch(waves_per_stream=N.WAVES_CHUNKED)
        res = b.results()
        for i, (raw, plain) in enumerate(pairs):
            assert res[i]["good"] == 1 and int(res[i]["final_size"]) == len(plain), (seed, chunk, i, res[i])
            assert b.output(i, res) == plain, (seed, chunk, i)
        del b
    # damaged copies (a flipped bit, a cut): the chunk path must return what a whole workgroup per
    # stream returns -- status, size and every decoded byte
    os.environ.pop("DEBIG_CHUNK_BYTES", None)
    bad, bcaps = [], []
    for raw, plain in pairs:
        b2 = bytearray(raw)
        if rng.random() < 0.7:
            for _ in range(rng.randint(1, 2)):
                b2[rng.randrange(len(b2))] ^= 1 << rng.randrange(8)
        else:
            b2 = b2[:rng.randint(len(b2) // 4, len(b2) - 1)]
        bad.append(bytes(b2))
        bcaps.append(max(len(plain) + 4096, len(b2)))
    got = {}
    for width in (8, N.WAVES_CHUNKED):
        b = DeviceBatch.from_streams(bad, bcaps, out_skew=3)
        b.launch(waves_per_stream=width)
        res = b.results()
        host = b.outputs_host()
        rows = []
        for i in range(len(bad)):
            off = int(b.streams_host[i]["out_off"])
            n = int(res[i]["final_size"]) if res[i]["final_set"] else 0
            rows.append((int(res[i]["good"]), int(res[i]["status"]), int(res[i]["final_set"]), n,
                         zlib.crc32(host[off:off + n].tobytes())))
        got[width] = rows
        del b
    for i, (a, c) in enumerate(zip(got[8], got[N.WAVES_CHUNKED])):
        assert a == c, (seed, "damaged", i, a, c)
    rounds += 1
    streams += 3 * len(pairs)
    nbytes += 3 * sum(caps)
    seed += 1
    print(f"seed {seed - 1}: ok  ({streams} streams, {nbytes / 1e9:.2f} GB decoded, {time.time() - t0:.0f} s)", flush=True)
print(f"fuzz ok: {rounds} rounds, {streams} streams, {nbytes / 1e9:.2f} GB, seeds {seed0}..{seed - 1}")
