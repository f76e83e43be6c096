#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the COMPILED, UNMODIFIED reference (oracle/_ref).

Runs only where /root/reference exists (the build container).  The fixtures are DATA:
inputs (the reference's own sample files copied byte-for-byte into resources/, small
synthetic streams as hex) and the reference's outputs (sha256 digests, small literal
vectors).  The GPU box, which never sees the reference, checks the product and the
oracle against these files.

    python tests/golden/make_golden.py
"""
import glob
import hashlib
import json
import os
import random
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import binding  # noqa: E402
from debigulator_amd import workload  # noqa: E402


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def corrupt_corpus():
    """6. damaged raw streams through reference build B in a child process (tests/ref_worker.py):
    how the reference FAILS (src/inflate.c:1427-1434 bad code-length code, :1809 distance symbol,
    :1843-1852 too-far with the partial final size, :465-473 no code, :949 stored LEN/NLEN).  Cases on
    which the reference is in undefined behaviour (oracle ub_flags) or dies are left out."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ref_worker import RefWorker, corrupt_cases

    orc = binding.Oracle()
    w = RefWorker()
    keep, quota = [], {"good": 60, "fail_partial": 130, "fail_empty": 20}
    for raw, cap in corrupt_cases(20261004, 3000):
        g, f, o, st = orc.inflate(raw, cap, want_stats=True)
        if st.ub_flags:
            continue
        r = w.inflate("B", raw, cap)
        if r is None:
            continue
        good, final, digest = r
        cls = "good" if good else ("fail_partial" if final else "fail_empty")
        if quota[cls] == 0:
            continue
        quota[cls] -= 1
        # a recipient just large enough: nothing the reference produced is cut off by it
        small = max((final or 0) + 1, len(raw)) + 16
        assert w.inflate("B", raw, small) == r
        keep.append({"raw_hex": raw.hex(), "recipient_size": small, "good": good, "final": final,
                     "out_sha256": digest, "oracle": "B", "class": cls})
    w.close()
    assert sum(quota.values()) == 0, quota
    json.dump(keep, open(os.path.join(HERE, "corpus_corrupt.json"), "w"))
    print("corpus_corrupt.json:", len(keep), "cases; reference deaths:", w.crashes)


def main():
    binding.build(ref=True)
    if "--only-corrupt" in sys.argv:
        return corrupt_corpus()
    A = binding.Reference("A")  # silent, asserts on (canonical)
    B = binding.Reference("B")  # silent, asserts off (inputs on which A aborts)

    # ---- 1. the reference's own sample files (BASELINE configs 1 and 3)
    res = {"png": {}, "gz": {}}
    for f in sorted(glob.glob(os.path.join(HERE, "resources", "*.png"))):
        d = open(f, "rb").read()
        w, h, g = A.png_wh(d)
        good, out = A.decode_png(d, 120_000_000, tid=1)
        res["png"][os.path.basename(f)] = {"input_sha256": sha(d), "width": w, "height": h, "good": int(good),
                                           "rgba_sha256": sha(out.tobytes())}
    d = open(os.path.join(HERE, "resources", "gzipsample.gz"), "rb").read()
    good, out = A.decode_gz(d, 561872)
    res["gz"]["gzipsample.gz"] = {"input_sha256": sha(d), "good": int(good), "size": 561872, "sha256": sha(out)}
    json.dump(res, open(os.path.join(HERE, "resources.json"), "w"), indent=1, sort_keys=True)

    # ---- 2. known-answer streams through raw inflate()
    kat = []
    k1 = bytes.fromhex("0de10190244992244902") + b"\0" * 48 + b"\x32" + b"\0" * 78
    k2 = bytes.fromhex("0de00190244992244902") + b"\0" * 48 + b"\x32" + b"\0" * 78
    cases = {
        "K1": (k1 + bytes.fromhex("1023ba05"), 256), "K2": (k2 + bytes.fromhex("10a35b"), 256),
        "K3": (k2 + bytes.fromhex("108309"), 256), "K4": (k1 + bytes.fromhex("1023fa05"), 256),
        "K5": (bytes.fromhex("010500faff7878787878"), 64), "K6": (bytes.fromhex("01050000007878787878"), 64),
        "K7": (bytes.fromhex("4b4c4a4e842100"), 64), "K8": (b"\x07" + b"\0" * 8, 64),
        "K10": (zlib.compress(b"a")[2:-4], 64), "K11": (zlib.compress(os.urandom(0) + bytes(range(256)) * 8, 0)[2:-4], 100),
    }
    for name, (raw, cap) in cases.items():
        ref = B if name in ("K3", "K8") else A  # A aborts on these two (inflate.c:702 / :997)
        good, final, out = ref.inflate(raw, cap)
        kat.append({"name": name, "raw_hex": raw.hex(), "recipient_size": cap, "good": int(good),
                    "final": final, "out_hex": out.hex(), "oracle": "B" if ref is B else "A"})
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    # ---- 3. differential corpus: small zlib-made raw streams (bytes stored, so the local
    #         zlib version never matters again) + what the reference returns for them
    rng = random.Random(20261003)
    corpus = []

    def payload(n, kind):
        if kind == 0:
            return bytes(rng.getrandbits(8) for _ in range(n))
        if kind == 1:
            words = [bytes(rng.getrandbits(8) for _ in range(rng.randint(3, 9))) for _ in range(60)]
            b = bytearray()
            while len(b) < n:
                b += rng.choice(words) + b" "
            return bytes(b[:n])
        if kind == 2:
            return bytes([rng.choice(b"ab")]) * n
        return bytes(rng.choice(b"abcdefgh") for _ in range(n))

    while len(corpus) < 240:
        data = payload(rng.randint(1, 1500), rng.randint(0, 3))
        strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
        c = zlib.compressobj(rng.choice([0, 1, 6, 9]), zlib.DEFLATED, -15, rng.choice([1, 9]), strat)
        raw = c.compress(data[: len(data) // 2]) + (c.flush(zlib.Z_FULL_FLUSH) if rng.random() < 0.5 else b"")
        raw += c.compress(data[len(data) // 2:]) + c.flush()
        cap = max(len(data) + 1, len(raw))
        good, final, out = A.inflate(raw, cap)
        corpus.append({"raw_hex": raw.hex(), "recipient_size": cap, "good": int(good), "final": final,
                       "out_sha256": sha(out), "truncated_by_tail_rule": int(out != data)})
    json.dump(corpus, open(os.path.join(HERE, "corpus_zlib.json"), "w"))

    # ---- 4. the synthetic generator (BASELINE config 2 shapes): pin generator determinism and
    #         the reference's answer for the first streams of every kind
    gen = []
    for kind in ("stored", "fixed", "dynamic"):
        for i in range(8):
            raw, plain = workload.make_stream(kind, i, 65536)
            cap = max(65537, len(raw))
            good, final, out = A.inflate(raw, cap)
            gen.append({"kind": kind, "index": i, "size": 65536, "raw_len": len(raw), "raw_sha256": sha(raw),
                        "plain_sha256": sha(plain.tobytes()), "good": int(good), "final": final,
                        "out_sha256": sha(out)})
    json.dump(gen, open(os.path.join(HERE, "generator.json"), "w"), indent=1)

    # ---- 5. synthetic PNGs through the reference's decode_png (all filter types, ct 6 / 3)
    pngs = []
    for j, (w, h, ct, ft, enc) in enumerate([(64, 48, 6, 5, "dynamic"), (200, 130, 6, 4, "dynamic"),
                                             (97, 61, 6, 3, "fixed"), (33, 7, 6, 1, "stored"),
                                             (120, 80, 3, 5, "dynamic"), (1, 1, 6, 0, "dynamic"),
                                             (301, 3, 6, 2, "dynamic"), (257, 200, 6, 5, "dynamic")]):
        png, pix = workload.make_png(1000 + j, w, h, ct=ct, ftype=ft, noise=6, enc=enc, idat_chunk=4096)
        good, out = A.decode_png(png, 120_000_000, tid=1)
        pngs.append({"seed": 1000 + j, "w": w, "h": h, "ct": ct, "ftype": ft, "enc": enc, "png_hex": png.hex(),
                     "good": int(good), "rgba_sha256": sha(out.tobytes())})
    json.dump(pngs, open(os.path.join(HERE, "png_synth.json"), "w"))
    corrupt_corpus()
    print("golden fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
