"""GPU parity of the drop-in C API (inflate / decode_png / decode_gz prototypes of the
reference) against the golden fixtures made from the compiled reference
(tests/golden/make_golden.py) and against the oracle."""
import glob
import hashlib
import json
import os
import random

import numpy as np
import pytest

from debigulator_amd import workload

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def api(gpu_device):
    from debigulator_amd import api as _api

    return _api


def test_inflate_known_answers(api):
    """SURVEY.md Appendix B streams: reference good / final / bytes."""
    for k in json.load(open(os.path.join(GOLD, "kat.json"))):
        good, final, out = api.inflate(bytes.fromhex(k["raw_hex"]), k["recipient_size"])
        assert good == k["good"], k["name"]
        assert final == k["final"], k["name"]
        assert out.hex() == k["out_hex"], k["name"]


def test_inflate_corpus_vs_reference_digests(api):
    corpus = json.load(open(os.path.join(GOLD, "corpus_zlib.json")))
    res = api.inflate_batch([bytes.fromhex(c["raw_hex"]) for c in corpus], [c["recipient_size"] for c in corpus])
    for c, (good, final, out) in zip(corpus, res):
        assert good == c["good"] and final == c["final"]
        assert sha(out) == c["out_sha256"]
    assert sum(c["truncated_by_tail_rule"] for c in corpus) > 0  # the Q2 tail rule is exercised


def test_inflate_batch_big_batch_with_a_few_large_streams(api):
    """debig_inflate_batch picks the kernel width from the batch (csrc/host/debig_ctx.h:
    debig_pick_waves): more than 1024 streams of which a few are large runs those 4-wide next
    to the small ones.  Whatever it picks, every stream must come back exact."""
    from debigulator_amd import workload

    small = workload.make_streams("dynamic", 16, 20000)
    large = workload.make_streams("fixed", 2, 2 << 20)
    pairs = [small[i % 16] for i in range(1100)]
    pairs[7], pairs[600], pairs[1099] = large[0], large[1], large[0]
    res = api.inflate_batch([p[0] for p in pairs], [len(p[1]) + 1 for p in pairs])
    for i, ((good, final, out), (_, plain)) in enumerate(zip(res, pairs)):
        assert good == 1 and final == len(plain), i
        if i % 97 == 0 or i in (7, 600, 1099):
            assert out == plain.tobytes(), i


def test_inflate_batch_with_oversized_recipients_downloads_only_what_was_decoded(api):
    """csrc/host/debig_ctx.c: debig_download_unpack.  The device output arena is laid out by the
    recipients' capacities; 300 streams that decode to 20 KB into 1 MiB recipients leave it 98 % empty,
    so the decoded ranges are packed on the device (debig_hip_gather) before they cross PCIe.  One
    failing stream and one empty-output stream ride along; every byte must still be right, and bytes
    beyond final_recipient_size must not be touched in the callers' buffers."""
    from debigulator_amd import workload

    pairs = [workload.make_stream(("dynamic", "fixed", "stored")[i % 3], 50 + i, 20000) for i in range(300)]
    datas = [bytes(p[0]) for p in pairs]
    caps = [1 << 20] * len(pairs)
    bad = bytearray(datas[5]); bad[len(bad) // 2] ^= 0x40
    datas[5] = bytes(bad)
    res = api.inflate_batch(datas, caps)
    n_ok = 0
    for i, ((good, final, out), (_, plain)) in enumerate(zip(res, pairs)):
        if i == 5:
            continue
        assert good == 1 and final == len(plain), i
        assert out == plain.tobytes(), i
        n_ok += 1
    assert n_ok == len(pairs) - 1
    # dense layout right behind it on the same thread id (the arenas are reused)
    res = api.inflate_batch(datas[:40], [max(len(p[1]) + 1, len(p[0])) for p in pairs[:40]])  # Q1: recipient >= input
    for i, ((good, final, out), (_, plain)) in enumerate(zip(res, pairs[:40])):
        if i != 5:
            assert good == 1 and out == plain.tobytes(), i


def test_inflate_batch_skewed_sizes_are_dispatched_longest_first(api):
    """A batch of 600 streams in which a quarter is large goes through debig_plan_batch
    (descriptors launched longest first, 4 wavefronts wide, results returned in the caller's
    order): every stream must come back at its own index."""
    from debigulator_amd import workload

    small = workload.make_streams("dynamic", 8, 3000)
    large = workload.make_streams("fixed", 3, 400000)
    pairs = [large[i % 3] if i % 4 == 1 else small[i % 8] for i in range(600)]
    res = api.inflate_batch([p[0] for p in pairs], [len(p[1]) + 1 for p in pairs])
    for i, ((good, final, out), (_, plain)) in enumerate(zip(res, pairs)):
        assert good == 1 and final == len(plain), i
        if i % 37 == 0 or i % 4 == 1 and i < 40:
            assert out == plain.tobytes(), i


def test_inflate_argument_gates(api):
    import ctypes as C
    from debigulator_amd import _native as N

    L = api._lib()
    good = C.c_uint32(7)
    fin = C.c_uint64(123)
    buf = np.zeros(64, dtype=np.uint8)
    raw = np.frombuffer(bytes.fromhex("4b4c4a4e842100"), dtype=np.uint8)
    # NULL recipient / final / input: good = 0, final untouched (src/inflate.c:797-824)
    L.debig_inflate(None, 64, C.byref(fin), None, 0, raw.ctypes.data, len(raw), C.byref(good), 0)
    assert good.value == 0 and fin.value == 123
    good.value = 7
    L.debig_inflate(buf.ctypes.data, 64, None, None, 0, raw.ctypes.data, len(raw), C.byref(good), 0)
    assert good.value == 0
    good.value = 7
    L.debig_inflate(buf.ctypes.data, 64, C.byref(fin), None, 0, None, len(raw), C.byref(good), 0)
    assert good.value == 0 and fin.value == 123
    # recipient smaller than the input, input shorter than 5 bytes: final untouched as well
    assert api.inflate(raw.tobytes(), 6) == (0, None, b"")
    assert api.inflate(b"\x03\x00", 64) == (0, None, b"")
    assert api.inflate(raw.tobytes(), 64)[0:2] == (1, 12)
    assert N.lib() is L


def test_decode_gz_sample(api):
    g = json.load(open(os.path.join(GOLD, "resources.json")))["gz"]["gzipsample.gz"]
    data = open(os.path.join(GOLD, "resources", "gzipsample.gz"), "rb").read()
    assert sha(data) == g["input_sha256"]
    good, out = api.decode_gz(data)
    assert good == 1 and len(out) == g["size"] and sha(out) == g["sha256"]
    assert api.decode_gz(b"\x1f\x8b\x07" + data[3:])[0] == 0       # CM != 8
    assert api.decode_gz(b"PK" + data[2:])[0] == 0                 # bad magic


def test_decode_png_resources_match_reference(api, oracle):
    """All 15 sample PNGs of the reference equal the reference's own output bit for bit:
    phoebus.png thanks to the P2 aliasing replay, backgrounddetailed1.png (colour type 2)
    thanks to the P3 replay (zero-initialised caller buffer, as in the golden run)."""
    gold = json.load(open(os.path.join(GOLD, "resources.json")))["png"]
    files = sorted(glob.glob(os.path.join(GOLD, "resources", "*.png")))
    assert len(files) == 15
    datas = [open(f, "rb").read() for f in files]
    res = api.decode_png_batch(datas)
    for f, d, (good, rgba) in zip(files, datas, res):
        name = os.path.basename(f)
        g = gold[name]
        assert sha(d) == g["input_sha256"]
        assert good == g["good"] == 1, name
        w, h, _ = api.decode_png_get_width_height(d)
        assert (w, h) == (g["width"], g["height"])
        assert sha(rgba.tobytes()) == g["rgba_sha256"], name
        # and the single-call entry point agrees with the batch one
    good, rgba = api.decode_png(datas[files.index(os.path.join(GOLD, "resources", "phoebus.png"))])
    assert good == 1 and sha(rgba.tobytes()) == gold["phoebus.png"]["rgba_sha256"]


def test_decode_png_synthetic_all_filters(api):
    for p in json.load(open(os.path.join(GOLD, "png_synth.json"))):
        good, rgba = api.decode_png(bytes.fromhex(p["png_hex"]))
        assert good == p["good"], p
        assert sha(rgba.tobytes()) == p["rgba_sha256"], (p["w"], p["h"], p["ct"], p["ftype"])


def test_decode_png_rejects(api):
    data = bytearray(open(os.path.join(GOLD, "resources", "structuredart1.png"), "rb").read())
    bad_crc = bytearray(data); bad_crc[-5] ^= 1                       # IEND CRC
    assert api.decode_png(bytes(bad_crc))[0] == 0
    not_png = bytearray(data); not_png[1] = ord("Q")
    assert api.decode_png(bytes(not_png))[0] == 0
    assert api.decode_png(bytes(data), rgba_size=10 * 10 * 4 + 4)[0] == 0   # wrong rgba_values_size
    assert api.decode_png(bytes(data[:-12]))[0] == 0                        # no chunk after IDAT (P6)
    assert api.decode_png(bytes(data))[0] == 1


def test_legacy_names(api):
    import ctypes as C

    L = api._lib()
    data = np.frombuffer(open(os.path.join(GOLD, "resources", "structuredart2.png"), "rb").read(), dtype=np.uint8)
    w, h, g = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L.init_PNG_decoder(None)
    L.get_PNG_width_height(data.ctypes.data, len(data), C.byref(w), C.byref(h), C.byref(g))
    assert (w.value, h.value, g.value) == (10, 10, 1)
    out = np.zeros(400, dtype=np.uint8)
    L.decode_PNG(data.ctypes.data, len(data), out.ctypes.data, 400, C.byref(g))
    gold = json.load(open(os.path.join(GOLD, "resources.json")))["png"]["structuredart2.png"]
    assert g.value == 1 and sha(out.tobytes()) == gold["rgba_sha256"]


def test_decode_png_rgb_p3_depends_on_prior_buffer_like_the_reference(api, oracle):
    """colour type 2: the reference's result is a function of the stream AND of what the
    caller's buffer held before (P3).  Same prior bytes -> same bytes as the oracle (which is
    pinned to the reference on this file); DEBIG_STRICT=1 -> a real RGBA image."""
    import ctypes as C

    data = open(os.path.join(GOLD, "resources", "backgrounddetailed1.png"), "rb").read()
    w, h, _ = api.decode_png_get_width_height(data)
    prior = (np.arange(w * h * 4, dtype=np.uint32) * 2654435761 >> 13).astype(np.uint8)
    want_good, want = oracle.decode_png(data, prior=prior)
    L = api._lib()
    d = np.frombuffer(data, dtype=np.uint8)
    out = prior.copy()
    good = C.c_uint8(7)
    api.decode_png_init(thread_id=3)
    L.decode_png(d.ctypes.data, len(d), out.ctypes.data, out.size, 3, C.byref(good))
    assert good.value == want_good == 1
    assert np.array_equal(out, want)
    os.environ["DEBIG_STRICT"] = "1"
    try:
        good2, rgba = api.decode_png(data, thread_id=3)
    finally:
        del os.environ["DEBIG_STRICT"]
    assert good2 == 1 and rgba.reshape(h, w, 4)[:, :, 3].min() == 255
    g3, strict_want = oracle.decode_png(data, flags=1)  # ORC_PNG_STRICT does not touch P3; compare alpha only
    assert g3 == 1


def test_decode_png_very_wide_rows(api, oracle, gpu_device):
    """No width limits (the reference has none, src/decode_png.c:1512-1564): a palette image 20 000
    pixels wide (its index row no longer fits the de-filter kernel's LDS row: it is handed from band to
    band through device scratch) and an RGB image 5 000 pixels wide through the colour-type-2 replay
    (rows go through the LDS in pieces), against the oracle; the palette image also through the
    device-resident batch class."""
    from debigulator_amd.png_device import DevicePngBatch

    rng = np.random.default_rng(5)
    pal = rng.integers(0, 256, 768, dtype=np.uint8)
    png3, _ = workload.make_png(31, 20000, 131, ct=3, ftype=4, noise=3, enc="dynamic", palette=pal)
    want_good, want = oracle.decode_png(png3)
    assert want_good == 1
    good, rgba = api.decode_png(png3, thread_id=4)
    assert good == 1 and np.array_equal(rgba.reshape(-1), np.asarray(want).reshape(-1))
    b = DevicePngBatch([png3], device=gpu_device)
    b.launch()
    res, ires = b.results()
    assert (res["good"] == 1).all() and (ires["good"] == 1).all()
    assert np.array_equal(b.rgba(0).reshape(-1), np.asarray(want).reshape(-1))
    for ftype in (4, 3, 1):
        png2, _ = workload.make_png(32 + ftype, 5000, 7, ct=2, ftype=ftype, noise=5, enc="fixed")
        want_good, want = oracle.decode_png(png2)
        good, rgba = api.decode_png(png2, thread_id=4)
        assert good == want_good == 1, ftype
        assert np.array_equal(rgba.reshape(-1), np.asarray(want).reshape(-1)), ftype


def test_gzip_trailer_verification_on_gpu(api):
    """extension beyond the reference (which reads the CRC32/ISIZE trailer and ignores it,
    src/decode_gz.c:281-297): checked on the GPU against the decompressed bytes"""
    from debigulator_amd import workload

    members, plains = [], []
    for i in range(6):
        plain = workload.payload("text", 500 + i, 100000 + 777 * i)
        members.append(bytearray(workload.gzip_member(workload.encode("dynamic", plain), plain)))
        plains.append(plain.tobytes())
    members[2][-6] ^= 0x40   # damage the stored CRC-32
    members[4][-2] ^= 0x01   # damage ISIZE
    res = api.decode_gz_batch([bytes(m) for m in members], [len(p) + 1 for p in plains])
    for i, (good, out, tok) in enumerate(res):
        assert good == 1 and out == plains[i]          # the reference's verdict: all fine
        assert tok == (0 if i in (2, 4) else 1)


def test_checksum_kernels_vs_zlib(gpu_device):
    import zlib
    import torch
    from debigulator_amd.checksum import ADLER32, CRC32, DeviceChecksums

    arena = torch.randint(0, 256, (3_000_000,), dtype=torch.uint8, device=gpu_device)
    host = arena.cpu().numpy()
    spans = [(0, 0), (5, 1), (6, 3), (7, 4), (100, 15), (101, 16), (102, 17), (1000, 16383), (20001, 16384),
             (40003, 16385), (70000, 100000), (200001, 1234567), (1500000, 1499999)]
    for kind, fn in ((CRC32, zlib.crc32), (ADLER32, zlib.adler32)):
        ck = DeviceChecksums(arena, spans, kind)
        ck.launch()
        got = ck.results()
        for (o, n), g in zip(spans, got):
            assert g == fn(host[o:o + n].tobytes()), (kind, o, n)


# ---------------------------------------------------------------------------------------------
# debig_gunzip_batch (include/decode_gz.h): beyond the reference -- complete RFC 1952 headers,
# multi-member files, CRC-32/ISIZE verified on the GPU.  Ground truth: Python's zlib/gzip.
def _gz_member(data, level=6, name=None, comment=None, extra=None, hcrc=False, text=False):
    import struct
    import zlib

    flg = (1 if text else 0) | (2 if hcrc else 0) | (4 if extra is not None else 0) | (8 if name else 0) | \
        (16 if comment else 0)
    h = bytes([31, 139, 8, flg, 0, 0, 0, 0, 0, 255])
    if extra is not None:
        h += struct.pack("<H", len(extra)) + extra
    if name:
        h += name + b"\0"
    if comment:
        h += comment + b"\0"
    if hcrc:
        h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    return h + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def _bgzf_member(data, level=6):
    import struct

    plain = _gz_member(data, level, extra=b"BC\x02\x00\x00\x00")
    m = bytearray(plain)
    struct.pack_into("<H", m, 16, len(m) - 1)  # BSIZE = total member size - 1 (after SI1 SI2 LEN at 12..15)
    return bytes(m)


def test_gunzip_batch_headers_members_and_trailers(api):
    import gzip
    import random

    rng = random.Random(5)

    def blob(n):
        return bytes(rng.choice(b"abcdefgh \n") for _ in range(n))

    files, want = [], []
    # 0: python's own gzip writer (FNAME set via GzipFile default, mtime) -- the plain case
    d0 = blob(70000)
    files.append(gzip.compress(d0, 6)); want.append((0, d0, 1))
    # 1: every optional header field at once
    d1 = blob(5000)
    files.append(_gz_member(d1, 9, name=b"a name.txt", comment=b"a comment", extra=b"XY\x03\x00abc", hcrc=True, text=True))
    want.append((0, d1, 1))
    # 2: five members, different levels (incl. stored), different optional fields, an empty member
    parts = [blob(30000), blob(1), b"", blob(100000), blob(257)]
    files.append(_gz_member(parts[0], 1) + _gz_member(parts[1], 0, name=b"x") + _gz_member(parts[2], 6) +
                 _gz_member(parts[3], 9, comment=b"c") + _gz_member(parts[4], 6, hcrc=True))
    want.append((0, b"".join(parts), 5))
    # 3: BGZF-style: 40 members with their size in a BC subfield + the empty EOF member
    chunks = [blob(rng.randint(1, 20000)) for _ in range(40)]
    files.append(b"".join(_bgzf_member(c) for c in chunks) + _bgzf_member(b""))
    want.append((0, b"".join(chunks), 41))
    # 4: zero padding after the last member is accepted; 5: other bytes are reported
    d4 = blob(3000)
    files.append(_gz_member(d4) + b"\0" * 1000); want.append((0, d4, 1))
    files.append(_gz_member(d4) + b"garbage!garbage!"); want.append((7, d4, 1))
    # 6: a damaged second member keeps the first one's output; 7: CRC mismatch; 8: ISIZE mismatch
    m2 = bytearray(_gz_member(blob(20000)))
    m2[len(m2) // 2] ^= 0x10
    files.append(_gz_member(d4) + bytes(m2)); want.append((None, d4, 1))
    bad_crc = bytearray(_gz_member(d4)); bad_crc[-8] ^= 1
    files.append(bytes(bad_crc)); want.append((5, d4, 1))
    bad_isz = bytearray(_gz_member(d4)); bad_isz[-1] ^= 1
    files.append(bytes(bad_isz)); want.append((6, d4, 1))
    # 9: truncated inside the stream; 10: not gzip; 11: reserved flag bit; 12: output too small
    files.append(_gz_member(blob(50000))[:2000]); want.append((None, None, 0))
    files.append(b"PK\x03\x04 not a gzip file at all"); want.append((1, b"", 0))
    r = bytearray(_gz_member(d4)); r[3] |= 0x40
    files.append(bytes(r)); want.append((1, b"", 0))
    files.append(_gz_member(blob(50000))); want.append((4, None, 0))
    caps = [len(w[1]) + 64 if w[1] is not None else 1 << 17 for w in want]
    caps[12] = 1000
    got = api.gunzip_batch(files, caps)
    for i, ((st, out, members), (wst, wout, wmem)) in enumerate(zip(got, want)):
        if wst is not None:
            assert st == wst, (i, api.GZ_STATUS[st])
        else:
            assert st not in (0, 7), (i, api.GZ_STATUS[st])  # damaged: some error, whichever check trips first
        if wout is not None:
            assert out == wout, i
        if wst == 0:
            assert members == wmem, (i, members)
    # python agrees on the multi-member file
    assert gzip.decompress(files[2]) == want[2][1]


def test_inflate_batch_multi_one_device_and_staging(api, oracle, monkeypatch):
    """debig_inflate_batch_multi (include/inflate.h) with n_devices = 1 (one worker thread, its own
    device context) and debig_inflate_batch over 3000 host-buffer streams: inputs go up in one
    transfer through the page-locked arena, outputs come down in pieces and are unpacked by host
    threads.  Mixed sizes, a gated stream, a failing stream, a NULL input -- vs the oracle."""
    import ctypes as C

    L = api._lib()
    L.debig_inflate_batch_multi.restype = C.c_int
    L.debig_inflate_batch_multi.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_uint32]
    rng = random.Random(31)
    raws, caps = [], []
    for i in range(3000):
        kind = ("fixed", "dynamic", "stored")[i % 3]
        raw, plain = workload.make_stream(kind, 7000 + i, rng.choice([700, 5000, 20000, 65536]))
        if i % 97 == 5:
            raw = raw[: len(raw) // 2]  # damaged
        raws.append(raw)
        caps.append(max(len(plain) + 1, len(raw)) if i % 211 else 8)  # a few fail the size gate
    want = [oracle.inflate(r, c) for r, c in zip(raws, caps)]
    got = api.inflate_batch(raws, caps)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, i
    # the multi-device entry point, one device
    n = 600
    ins = [np.frombuffer(r, dtype=np.uint8) for r in raws[:n]]
    outs = [np.zeros(max(c, 1), dtype=np.uint8) for c in caps[:n]]
    in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins])
    out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    in_ptrs[17] = None  # NULL input: that stream is skipped, goods = 0, final untouched
    in_sizes = (C.c_uint64 * n)(*[len(a) for a in ins])
    capsa = (C.c_uint64 * n)(*caps[:n])
    finals = (C.c_uint64 * n)(*([api.NOT_SET] * n))
    goods = (C.c_uint32 * n)()
    assert L.debig_inflate_batch_multi(out_ptrs, capsa, finals, in_ptrs, in_sizes, goods, n, 1) == 0
    for i in range(n):
        g, f, o = want[i]
        if i == 17:
            assert goods[i] == 0 and finals[i] == api.NOT_SET
            continue
        assert goods[i] == g, i
        if f is None:
            assert finals[i] == api.NOT_SET
        else:
            assert finals[i] == f and outs[i][:f].tobytes() == o, i
    assert L.debig_inflate_batch_multi(out_ptrs, capsa, finals, in_ptrs, in_sizes, goods, n, 9) != 0  # no such device
    assert L.debig_inflate_batch_multi(out_ptrs, capsa, None, in_ptrs, in_sizes, goods, n, 1) != 0  # NULL array: an error, no crash
    # three worker threads, three device contexts, all on device 0 (DEBIG_MULTI_ONE_DEVICE: the rehearsal
    # of the n_devices > 1 path on a one-GPU box): stream i is decoded by worker i mod 3
    import torch

    monkeypatch.setenv("DEBIG_MULTI_ONE_DEVICE", "1")
    dev_before = torch.cuda.current_device()
    for a in outs:
        a[:] = 0
    finals = (C.c_uint64 * n)(*([api.NOT_SET] * n))
    goods = (C.c_uint32 * n)()
    assert L.debig_inflate_batch_multi(out_ptrs, capsa, finals, in_ptrs, in_sizes, goods, n, 3) == 0
    monkeypatch.delenv("DEBIG_MULTI_ONE_DEVICE")
    assert torch.cuda.current_device() == dev_before
    for i in range(n):
        g, f, o = want[i]
        if i == 17:
            assert goods[i] == 0 and finals[i] == api.NOT_SET
            continue
        assert goods[i] == g, i
        if f is not None:
            assert finals[i] == f and outs[i][:f].tobytes() == o, i
    L.debig_inflate_batch_multi_release.restype = None
    L.debig_inflate_batch_multi_release()
    assert L.debig_inflate_batch_multi(out_ptrs, capsa, finals, in_ptrs, in_sizes, goods, 50, 1) == 0  # and again after the release


def test_large_streams_take_the_chunk_parallel_path_through_the_c_calls(api, monkeypatch):
    """Few streams with 1 MiB of input or more on average: the C host layer (csrc/host/debig_ctx.c)
    picks DEBIG_WAVES_CHUNKED and passes the batch in groups of streams that fit its workspace
    cap; with DEBIG_CHUNKED_WS_MB=48 the five streams below need several groups.  decode_png() of a
    1024 x 1024 all-Paeth image goes the same way.  Everything must come back exact."""
    pairs = workload.make_streams("dynamic", 3, 3 << 20) + workload.make_streams("png", 2, 4 << 20)
    assert sum(len(p[0]) for p in pairs) >= len(pairs) << 20
    for cap_mb in (None, "48"):
        if cap_mb is None:
            monkeypatch.delenv("DEBIG_CHUNKED_WS_MB", raising=False)
        else:
            monkeypatch.setenv("DEBIG_CHUNKED_WS_MB", cap_mb)
        res = api.inflate_batch([p[0] for p in pairs], [len(p[1]) for p in pairs])
        for i, ((good, final, out), (_, plain)) in enumerate(zip(res, pairs)):
            assert good == 1 and final == len(plain), (cap_mb, i)
            assert out == plain.tobytes(), (cap_mb, i)
    monkeypatch.delenv("DEBIG_CHUNKED_WS_MB", raising=False)
    # more than 1024 streams with a very large one among them: chunk tasks for the whole batch
    small = workload.make_streams("dynamic", 16, 20000)
    big = workload.make_streams("dynamic", 2, 12 << 20)
    assert max(len(b[0]) for b in big) >= 4 << 20
    mixed = [small[i % 16] for i in range(1100)]
    mixed[3], mixed[1000] = big[0], big[1]
    res = api.inflate_batch([p[0] for p in mixed], [len(p[1]) + 1 for p in mixed])
    for i, ((good, final, out), (_, plain)) in enumerate(zip(res, mixed)):
        assert good == 1 and final == len(plain), i
        if i % 97 == 0 or i in (3, 1000):
            assert out == plain.tobytes(), i
    png, pix = workload.make_png(7300, 1024, 1024, ct=6, ftype=4, noise=workload.CFG4_NOISE, enc="dynamic")
    assert len(png) >= 1 << 20
    good, rgba = api.decode_png(png)
    assert good == 1 and np.array_equal(rgba.reshape(1024, 4096), pix)
    outs = api.decode_png_batch([png, png, png])
    for good, rgba in outs:
        assert good == 1 and np.array_equal(np.asarray(rgba).reshape(1024, 4096), pix)


def test_gunzip_batch_large_members_go_through_chunk_tasks(api):
    """Files of a few LARGE gzip members (3-6 MB each, zlib level 6 and 9, a stored one): the
    members are inflated in chunk tasks (few streams, > 1 MiB of input each), and the next member's
    header is found from debig_result.in_end_bits of the chunk path -- the input span of every
    stream runs on into the following members (DEBIG_STREAM_NO_REF_GATES), whose block headers the
    block finder also sees.  python's gzip module is the referee."""
    import gzip
    import random

    rng = random.Random(17)

    def blob(n):
        words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 9))) for _ in range(500)]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(words) + b" "
        return bytes(out[:n])

    files, want = [], []
    for f in range(4):
        parts = [blob(rng.randint(3 << 20, 6 << 20)) for _ in range(rng.randint(1, 3))]
        raw = b""
        for k, p in enumerate(parts):
            raw += _gz_member(p, rng.choice([6, 9]) if (f + k) % 4 else 0, name=b"member" if k else None)
        files.append(raw)
        want.append((b"".join(parts), len(parts)))
        assert gzip.decompress(raw) == want[-1][0]
    got = api.gunzip_batch(files, [len(w[0]) + 8 for w in want])
    for i, ((st, out, members), (plain, nmem)) in enumerate(zip(got, want)):
        assert st == 0, (i, api.GZ_STATUS[st])
        assert members == nmem and out == plain, i
