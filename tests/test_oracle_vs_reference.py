"""CPU, build container only: the oracle against the COMPILED, UNMODIFIED reference
(oracle/_ref, built in place from /root/reference by oracle/Makefile).  Skipped where the
reference library is absent and cannot be built."""
import glob
import os
import random
import zlib

import numpy as np
import pytest

from oracle import binding

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref():
    if not binding.ref_available("A") and os.path.isdir("/root/reference/src"):
        binding.build(ref=True)
    if not binding.ref_available("A"):
        pytest.skip("reference library not built (no /root/reference here)")
    return binding.Reference("A")


def test_raw_inflate_fuzz(oracle, ref):
    rng = random.Random(77)
    truncated = 0
    for it in range(1500):
        n = rng.randint(1, 6000)
        kind = rng.randint(0, 3)
        if kind == 0:
            data = bytes(rng.getrandbits(8) for _ in range(n))
        elif kind == 1:
            words = [bytes(rng.getrandbits(8) for _ in range(rng.randint(3, 9))) for _ in range(100)]
            data = (b" ".join(rng.choice(words) for _ in range(n // 5 + 1)))[:n]
        elif kind == 2:
            data = bytes([rng.choice(b"ab")]) * n
        else:
            data = bytes(rng.choice(b"abcdefgh") for _ in range(n))
        c = zlib.compressobj(rng.choice([0, 1, 6, 9]), zlib.DEFLATED, -15, 9,
                             rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        raw = c.compress(data) + c.flush()
        cap = max(len(data) + 1, len(raw))
        a = oracle.inflate(raw, cap)
        b = ref.inflate(raw, cap)
        assert a == b, it
        truncated += a[2] != data
    assert truncated > 10  # the tail rule Q2 really fires and both sides agree on it


def test_flush_heavy_streams(oracle, ref):
    """Encoders that flush every few bytes (PNG writers flushing per row): hundreds of blocks,
    empty stored blocks in between, fixed and dynamic codes.  The GPU kernels treat this pattern
    specially (short-block probe), so the oracle is pinned to the reference on it as well."""
    rng = random.Random(78)
    for it in range(250):
        data = bytes(rng.choice(b"abcdefgh \n") for _ in range(rng.randint(20, 5000)))
        c = zlib.compressobj(rng.choice([0, 1, 6, 9]), zlib.DEFLATED, -15, 9,
                             rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE]))
        raw, i, maxchunk = b"", 0, rng.choice([3, 20, 60, 200])
        while i < len(data):
            n = rng.randint(1, maxchunk)
            raw += c.compress(data[i:i + n])
            i += n
            raw += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK]))
        raw += c.flush()
        cap = max(len(data) + 1, len(raw))
        assert oracle.inflate(raw, cap) == ref.inflate(raw, cap), it


def test_gates(oracle, ref):
    raw = zlib.compress(b"hello world " * 40)[2:-4]
    # (recipient_size between C and D is excluded: the asserts-on reference aborts there, Q12)
    for cap in (len(raw) - 1, 4096):
        assert oracle.inflate(raw, cap)[:2] == ref.inflate(raw, cap)[:2]
    for short in (b"\x03\x00", b"\x4b\x04\x00", b"\x4b\x04\x00\x00"):
        assert oracle.inflate(short, 64)[:2] == ref.inflate(short, 64)[:2] == (0, None)


def test_resources(oracle, ref):
    for f in sorted(glob.glob(os.path.join(GOLD, "resources", "*.png"))):
        d = open(f, "rb").read()
        g1, o1 = oracle.decode_png(d)
        g2, o2 = ref.decode_png(d, tid=2)
        assert g1 == g2 and np.array_equal(o1, o2), os.path.basename(f)
    d = open(os.path.join(GOLD, "resources", "gzipsample.gz"), "rb").read()
    g1, o1, n = oracle.decode_gz(d, 600000)
    g2, o2 = ref.decode_gz(d, n)
    assert g1 == g2 == 1 and o1 == o2


def _safe_offsets(png):
    """byte offsets whose corruption keeps the reference inside defined behaviour: chunk
    types, CRCs, IHDR flags, the zlib header -- never a chunk length or IDAT payload (the
    reference computes the CRC over `length` bytes before it bounds-checks the length)."""
    offs = list(range(0, 8))
    at = 8
    first_idat = True
    while at + 8 <= len(png):
        ln = int.from_bytes(png[at:at + 4], "big")
        typ = bytes(png[at + 4:at + 8])
        offs += list(range(at + 4, at + 8))
        if typ == b"IHDR":
            offs += list(range(at + 8 + 8, at + 8 + 13))
        if typ == b"IDAT" and first_idat:
            offs += [at + 8, at + 9]
            first_idat = False
        offs += list(range(at + 8 + ln, at + 12 + ln))
        at += 12 + ln
    return offs


def test_png_rejects_agree(oracle, ref):
    base = bytearray(open(os.path.join(GOLD, "resources", "structuredart1.png"), "rb").read())
    offs = _safe_offsets(base)
    rng = random.Random(5)
    rejected = 0
    for it in range(200):
        d = bytearray(base)
        for _ in range(rng.randint(1, 2)):
            d[rng.choice(offs)] ^= 1 << rng.randrange(8)
        g1, o1 = oracle.decode_png(bytes(d), rgba_size=400)
        g2, o2 = ref.decode_png(bytes(d), tid=2, rgba_size=400)
        assert g1 == g2, it
        rejected += g1 == 0
        if g1:
            assert np.array_equal(o1, o2), it
    assert rejected > 50


from ref_worker import RefWorker, corrupt_cases  # noqa: E402


def test_corrupted_raw_streams_fail_like_the_reference(oracle, ref):
    """VERDICT r1 missing-5: HOW the reference fails on damaged raw streams -- bad code-length code
    (src/inflate.c:1427-1434), distance symbol > 29 (:1809), distance beyond the output with the
    partial final size (:1843-1852), undecodable bit patterns (:465-473), stored LEN/NLEN (:949) --
    pinned on the same mutators the GPU suite leans on.  Reference build B (asserts off: build A
    aborts where B reports), in a child process; cases on which the oracle notes undefined
    behaviour of the reference (ub_flags) or the reference dies are excluded."""
    import hashlib

    w = RefWorker()
    compared = failed = partial = 0
    try:
        for it, (raw, cap) in enumerate(corrupt_cases(4242, 1500)):
            g, f, o, st = oracle.inflate(raw, cap, want_stats=True)
            if st.ub_flags:
                continue
            r = w.inflate("B", raw, cap)
            if r is None:
                continue
            assert (g, f, hashlib.sha256(o).hexdigest()) == r, it
            compared += 1
            failed += g == 0
            partial += g == 0 and bool(f)
    finally:
        w.close()
    assert compared > 800 and failed > 150 and partial > 100, (compared, failed, partial, w.crashes)
    assert w.crashes < 20, w.crashes
