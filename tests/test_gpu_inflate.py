"""GPU parity: the HIP inflate path (through the C-ABI) vs the oracle, bit-exact.

Reference behaviour under test: src/inflate.c:786-1965 (inflate()), quirks Q1-Q15
of SURVEY.md 8a.  Inputs are seeded; sizes are chosen so the oracle finishes in seconds.
"""
import hashlib
import random
import zlib

import numpy as np
import pytest

from debigulator_amd import workload
from debigulator_amd.batch import DeviceBatch

pytestmark = pytest.mark.gpu

# oracle.ub_flags for which the reference has no defined answer: over-subscribed code
# lengths (assert only), CL repeat at position 0 (reads table[-1])
UB_EXCLUDED = 0x10 | 0x02


def _zlib_raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=9, flushes=0, rng=None):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    raw = b""
    step = max(1, len(data) // (flushes + 1))
    for i in range(0, len(data), step):
        raw += c.compress(data[i:i + step])
        if flushes and rng.random() < 0.7:
            raw += c.flush(zlib.Z_FULL_FLUSH)
    return raw + c.flush()


def _payload(rng, n, kind):
    if kind == 0:
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == 1:
        words = [bytes(rng.getrandbits(8) for _ in range(rng.randint(3, 9))) for _ in range(200)]
        b = bytearray()
        while len(b) < n:
            b += rng.choice(words) + b" "
        return bytes(b[:n])
    if kind == 2:
        return bytes([rng.choice(b"ab")]) * n
    if kind == 3:
        return bytes(rng.choice(b"abcdefgh") for _ in range(n))
    return (bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 40))) * (n // 2 + 1))[:n]


# wavefronts per stream: debig_inflate_kernel, debig_inflate_mw_kernel<2>, <4>, <8>, and the two
# mixed modes (large streams 4-wide beside small ones 1- / 2-wide, include/debig_hip.h)
# 0x10 = DEBIG_WAVES_SPLIT: the scan + LZ77 kernel pair (what the library picks for n > 1024)
# 0x20 = DEBIG_WAVES_CHUNKED: large streams cut into chunk tasks at block headers
# 0x11 = DEBIG_WAVES_SPLIT_QUEUED: the pair behind persistent workgroups and a work queue
# 0x12 = DEBIG_WAVES_STRAND: the long-segment scan (csrc/inflate_strand_kernel.inc) in front of the same LZ77 half
# 0x13 = DEBIG_WAVES_STRAND_PIPE: the same with the scan and the LZ77 half on two wavefronts of a workgroup, side by side
WIDTHS = (1, 2, 4, 8, 0x41, 0x42, 0x10, 0x11, 0x12, 0x13, 0x20)


def _check(oracle, gpu_device, raws, caps, widths=WIDTHS, **kw):
    """Every kernel width must give the oracle's answer (include/debig_hip.h:
    debig_hip_inflate_batch_ex -- results are identical for every choice)."""
    exp = [oracle.inflate(r, c, want_stats=True) for r, c in zip(raws, caps)]
    b = DeviceBatch.from_streams(raws, caps, device=gpu_device, **kw)
    for width in widths:
        b.d_out.zero_()
        b.d_results.zero_()
        b.launch(waves_per_stream=width)
        res = b.results()
        host = b.outputs_host()
        for i, (g, f, o, st) in enumerate(exp):
            cap = int(b.streams_host[i]["out_cap"])
            off = int(b.streams_host[i]["out_off"])
            # nothing beyond recipient_size may be touched, whatever the input
            assert not host[off + cap:off + cap + 32].any(), f"width {width} stream {i}: wrote past recipient_size"
            if st.ub_flags & UB_EXCLUDED:
                continue  # the reference itself is in undefined behaviour here
            assert res[i]["good"] == g, (width, i, res[i], f)
            if f is None:
                assert res[i]["final_set"] == 0
                continue
            assert res[i]["final_set"] == 1
            assert int(res[i]["final_size"]) == f, (width, i, res[i], f)
            assert host[off:off + f].tobytes() == o, f"width {width} stream {i}: bytes differ"


def test_zlib_streams_all_strategies(oracle, gpu_device):
    rng = random.Random(11)
    raws, caps = [], []
    for it in range(300):
        data = _payload(rng, rng.randint(1, 40000), rng.randint(0, 4))
        strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
        raw = _zlib_raw(data, rng.choice([0, 1, 6, 9]), strat, rng.choice([1, 8, 9]), rng.randint(0, 3), rng)
        raws.append(raw)
        caps.append(max(len(data) + 1, len(raw)))
    _check(oracle, gpu_device, raws, caps)


@pytest.mark.parametrize("in_skew,out_skew", [(1, 0), (0, 3), (7, 13), (15, 15)])
def test_unaligned_buffers(oracle, gpu_device, in_skew, out_skew):
    rng = random.Random(5 + in_skew + out_skew)
    raws, caps = [], []
    for it in range(40):
        data = _payload(rng, rng.randint(1, 30000), rng.randint(0, 4))
        raw = _zlib_raw(data, rng.choice([0, 6]), zlib.Z_DEFAULT_STRATEGY, 9, rng.randint(0, 2), rng)
        raws.append(raw)
        caps.append(max(len(data) + 1, len(raw)))
    _check(oracle, gpu_device, raws, caps, in_skew=in_skew, out_skew=out_skew)


@pytest.mark.parametrize("kind", ["stored", "fixed", "dynamic"])
def test_cfg2_synthetic_64k(oracle, gpu_device, kind):
    """BASELINE config 2 shape (64 KiB streams), 64 of them checked byte for byte."""
    pairs = workload.make_streams(kind, 64, 65536)
    raws = [p[0] for p in pairs]
    caps = [max(65536 + 1, len(r)) for r in raws]
    _check(oracle, gpu_device, raws, caps)
    # the generator's own plain text is what must come out (Q2 cannot truncate these)
    b = DeviceBatch.from_streams(raws, caps, device=gpu_device)
    b.launch()
    res = b.results()
    for i, (_, plain) in enumerate(pairs):
        assert b.output(i, res) == plain.tobytes()


def test_high_ratio_and_long_matches(oracle, gpu_device):
    raws, caps = [], []
    for n in (1, 2, 3, 257, 258, 259, 8191, 8192, 8193, 100000, 1 << 20):
        for byte in (b"\0", b"ab", b"abc", bytes(range(7))):
            data = (byte * (n // len(byte) + 1))[:n]
            raw = _zlib_raw(data, 9)
            raws.append(raw)
            caps.append(max(len(data) + 1, len(raw)))
    _check(oracle, gpu_device, raws, caps)


def test_gates_and_errors(oracle, gpu_device):
    """Q1 gates, Q4 stored NLEN, Q10 distance too far, truncated and corrupted streams."""
    rng = random.Random(3)
    raws, caps = [], []
    good_raw = _zlib_raw(b"hello world " * 50, 6)
    raws += [good_raw, good_raw, b"\x03\x00", b"\x4b\x04\x00"]
    caps += [len(good_raw) - 1, 4096, 4096, 4096]  # recipient too small; ok; too short x2
    raws.append(bytes.fromhex("010500faff7878787878")); caps.append(64)      # K5
    raws.append(bytes.fromhex("01050000007878787878")); caps.append(64)      # K6 (Q4)
    raws.append(bytes.fromhex("4b4c4a4e842100")); caps.append(64)            # K7
    k1 = bytes.fromhex("0de10190244992244902") + b"\0" * 48 + b"\x32" + b"\0" * 78
    raws.append(k1 + bytes.fromhex("1023ba05")); caps.append(256)            # K1
    raws.append(k1 + bytes.fromhex("1023fa05")); caps.append(256)            # K4 (Q10)
    k2 = bytes.fromhex("0de00190244992244902") + b"\0" * 48 + b"\x32" + b"\0" * 78
    raws.append(k2 + bytes.fromhex("10a35b")); caps.append(256)              # K2 (Q6)
    raws.append(k2 + bytes.fromhex("108309")); caps.append(256)              # K3 (Q2)
    for it in range(200):  # corrupted / truncated: must agree with the oracle and stay in bounds
        data = _payload(rng, rng.randint(50, 4000), rng.randint(0, 4))
        raw = bytearray(_zlib_raw(data, rng.choice([1, 6, 9]),
                                  rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY])))
        if rng.random() < 0.5:
            raw = raw[: rng.randint(5, len(raw))]
        else:
            for _ in range(rng.randint(1, 3)):
                raw[rng.randrange(len(raw))] ^= 1 << rng.randrange(8)
        raws.append(bytes(raw))
        caps.append(max(len(data) * 4 + 64, len(raw)))
    _check(oracle, gpu_device, raws, caps)


def test_roundtrip_full_size_property(gpu_device):
    """BASELINE config 2 at FULL size (4096 x 64 KiB, fixed-Huffman): size-independent
    property -- every stream inflates to exactly the generator's plain text."""
    pairs = workload.make_streams("fixed", 4096, 65536)
    raws = [p[0] for p in pairs]
    caps = [max(65536 + 1, len(r)) for r in raws]
    b = DeviceBatch.from_streams(raws, caps, device=gpu_device)
    b.launch()
    res = b.results()
    assert (res["good"] == 1).all()
    assert (res["final_size"] == 65536).all()
    host = b.outputs_host()
    for i, (_, plain) in enumerate(pairs):
        off = int(b.streams_host[i]["out_off"])
        assert np.array_equal(host[off:off + 65536], plain), i


def test_streams_of_tiny_blocks(oracle, gpu_device):
    """Encoders that flush every few bytes (PNG writers flushing per row): hundreds of blocks of
    a few bytes each, empty stored blocks in between, clean / truncated / bit-flipped.  These
    run the kernels' short-block probe instead of the speculative rounds."""
    rng = random.Random(77)
    raws, caps = [], []
    for it in range(300):
        data = _payload(rng, rng.randint(20, 20000), rng.choice([1, 3, 4]))
        strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])
        c = zlib.compressobj(rng.choice([0, 1, 6, 9]), zlib.DEFLATED, -15, 9, strat)
        raw, i, maxchunk = b"", 0, rng.choice([3, 20, 60, 200, 1000])
        while i < len(data):
            n = rng.randint(1, maxchunk)
            raw += c.compress(data[i:i + n])
            i += n
            raw += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK]))
        raw = bytearray(raw + c.flush())
        mode = it % 3
        if mode == 1:
            raw = raw[: rng.randint(5, len(raw))]
        elif mode == 2:
            for _ in range(rng.randint(1, 3)):
                raw[rng.randrange(len(raw))] ^= 1 << rng.randrange(8)
        raws.append(bytes(raw))
        caps.append(max(len(data) * 3 + 64, len(raw)))
    _check(oracle, gpu_device, raws, caps, in_skew=3, out_skew=9)


def test_large_streams_every_width(gpu_device):
    """Few large streams (the decode_png config 4 regime): 4 MiB each, text-like fixed and
    dynamic Huffman and Paeth-filtered image rows; every kernel width must return the
    generator's plain bytes (size-independent round-trip property; the oracle is not
    needed at this size)."""
    size = 4 << 20
    pairs = []
    for kind in ("fixed", "dynamic", "png"):
        pairs += workload.make_streams(kind, 2, size)
    pairs += workload.make_streams("dynamic", 6, 40000)  # the small class of the mixed modes
    raws = [p[0] for p in pairs]
    caps = [len(p[1]) + 1 for p in pairs]
    b = DeviceBatch.from_streams(raws, caps, device=gpu_device, out_skew=5)
    for width in WIDTHS:
        b.d_out.zero_()
        b.d_results.zero_()
        b.launch(waves_per_stream=width)
        res = b.results()
        assert (res["good"] == 1).all(), (width, res)
        for i, (_, plain) in enumerate(pairs):
            assert b.output(i, res) == plain.tobytes(), (width, i)


def test_p2_aliasing_replay_every_width(gpu_device):
    """decode_png's buffer-aliasing replay (parameters p2_s0 / p2_est of debig_stream) is
    part of every inflate kernel.  The reference's sample files (phoebus.png is the one whose last 756 bytes the
    replay changes) go through inflate + de-filter with EVERY width and must come out with the REFERENCE's own
    digest (tests/golden/resources.json, made by the compiled reference) -- not merely the same for all widths."""
    import glob
    import hashlib
    import json
    import os

    from debigulator_amd.png_device import DevicePngBatch

    gdir = os.path.join(os.path.dirname(__file__), "golden")
    gold = json.load(open(os.path.join(gdir, "resources.json")))["png"]
    files = [f for f in sorted(glob.glob(os.path.join(gdir, "resources", "*.png")))
             if not f.endswith("backgrounddetailed1.png")]  # colour type 2: the P3 replay is another kernel's test
    assert len(files) == 14 and any(f.endswith("phoebus.png") for f in files)
    b = DevicePngBatch([open(f, "rb").read() for f in files], device=gpu_device)
    for width in WIDTHS:
        b.inflate.d_out[:b.pal_base].fill_(0xA5)  # (the palettes live behind the streams)
        b.inflate.d_results.zero_()
        b.d_rgba.zero_()
        b.launch(waves_per_stream=width)
        res, ires = b.results()
        assert (res["good"] == 1).all() and (ires["good"] == 1).all(), width
        for k, f in enumerate(files):
            name = os.path.basename(f)
            assert hashlib.sha256(b.rgba(k).tobytes()).hexdigest() == gold[name]["rgba_sha256"], (hex(width), name)


def test_chunked_path_against_the_oracle(oracle, gpu_device):
    """DEBIG_WAVES_CHUNKED on streams of many chunk tasks (8 MiB image rows and text, the library's
    own chunk size), intact and damaged (a flipped bit early, in the middle, near the end; a
    truncated stream; a recipient one byte short): good, size and every byte must be the ORACLE's,
    for the chunk-parallel path and for debig_inflate_mw_kernel<8> (the kernel it hands damaged
    streams to)."""
    rng = random.Random(77)
    pairs = workload.make_streams("png", 3, 8 << 20) + workload.make_streams("dynamic", 3, 8 << 20)
    raws, caps = [], []
    for raw, plain in pairs:
        raw = bytes(raw)
        raws.append(raw); caps.append(len(plain) + 1)
        for pos in (rng.randrange(50, 2000), len(raw) // 2 + rng.randrange(1000), len(raw) - rng.randrange(100, 3000)):
            bad = bytearray(raw)
            bad[pos] ^= 1 << rng.randrange(8)
            raws.append(bytes(bad)); caps.append(len(plain) + 4096)
    raws.append(raws[0][:len(raws[0]) * 2 // 3]); caps.append(caps[0])
    raws.append(raws[0]); caps.append(caps[0] - 2)  # one byte short of the output
    _check(oracle, gpu_device, raws, caps, widths=(8, 0x20), out_skew=3)
    # the intact ones really decode (the comparison above is not between two failures)
    for i in (0, 4):
        g, f, o, _ = oracle.inflate(raws[i], caps[i], want_stats=True)
        assert g == 1 and f == caps[i] - 1


def test_offsets_beyond_4_gib(oracle, gpu_device):
    """64-bit offset arithmetic in every kernel: the streams sit BEHIND the 4 GiB mark of both arenas
    (in_off, out_off > 2^32; sparse arenas: only the far end is ever touched).  Small streams of
    every block type, a damaged one and one large enough for several chunk tasks; every width must
    give the oracle's answer."""
    import torch

    from debigulator_amd.batch import DeviceBatch as DB, pack_streams

    rng = random.Random(4096)
    raws, caps = [], []
    for kind in ("stored", "fixed", "dynamic"):
        raw, plain = workload.make_stream(kind, 7, 40000)
        raws.append(bytes(raw)); caps.append(max(len(plain) + 1, len(raw)))
    bad = bytearray(raws[2]); bad[len(bad) // 2] ^= 0x10
    raws.append(bytes(bad)); caps.append(caps[2] + 4096)
    raw, plain = workload.make_stream("dynamic", 9, 3 << 20)
    raws.append(bytes(raw)); caps.append(len(plain) + 1)
    data = _payload(rng, 30000, 1)
    raws.append(_zlib_raw(data, 6, zlib.Z_DEFAULT_STRATEGY, 9, 2, rng)); caps.append(len(data) + 1)
    exp = [oracle.inflate(r, c, want_stats=True) for r, c in zip(raws, caps)]
    in_arena, streams, out_bytes = pack_streams(raws, caps, in_skew=5, out_skew=3)
    base_in, base_out = (1 << 32) + 4096 + 16, (1 << 32) + (1 << 20) + 32
    streams = streams.copy()
    streams["in_off"] += np.uint64(base_in)
    streams["out_off"] += np.uint64(base_out)
    b = DB.__new__(DB)
    b.torch, b.device, b.n, b.streams_host = torch, torch.device(gpu_device), len(raws), streams
    b.d_in = torch.empty(base_in + len(in_arena), dtype=torch.uint8, device=gpu_device)
    b.d_in[base_in:] = torch.from_numpy(in_arena).to(gpu_device)
    b.d_out = torch.empty(base_out + out_bytes, dtype=torch.uint8, device=gpu_device)
    b.order, b.planned_waves, b.d_ws, b.chunk_groups = None, 0, None, None
    b.dev_streams_host = streams
    b.d_streams = torch.from_numpy(streams.view(np.uint8).reshape(-1)).to(gpu_device)
    from debigulator_amd.batch import RESULT_DTYPE
    from debigulator_amd import _native

    b.d_results = torch.zeros(b.n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=gpu_device)
    b.lib = _native.lib()
    for width in WIDTHS:
        b.d_out[base_out:].zero_()
        b.d_results.zero_()
        b.launch(waves_per_stream=width)
        res = b.results()
        host = b.d_out[base_out:].cpu().numpy()
        for i, (g, f, o, st) in enumerate(exp):
            off = int(streams[i]["out_off"]) - base_out
            cap = int(streams[i]["out_cap"])
            assert not host[off + cap:off + cap + 32].any(), (width, i)
            if st.ub_flags & UB_EXCLUDED:
                continue
            assert res[i]["good"] == g, (width, i, res[i])
            if f is None:
                assert res[i]["final_set"] == 0
                continue
            assert int(res[i]["final_size"]) == f, (width, i, res[i], f)
            assert host[off:off + f].tobytes() == o, f"width {width:#x} stream {i}: bytes differ"
    assert exp[0][0] == 1 and exp[4][0] == 1 and exp[3][0] in (0, 1)


def test_chunked_path_many_small_tasks_agree_with_oracle(oracle, gpu_device, monkeypatch):
    """DEBIG_CHUNK_BYTES = 1024 / 3072 cuts ordinary test streams into dozens of chunk tasks each, so the
    block finder, the repair and chain kernels, the two-plane replay, the window walk and the translate
    kernel run on thousands of task boundaries on the real hardware (kernel-to-kernel visibility
    across XCDs is something the CPU emulator cannot show).  400 streams of mixed structure, a third
    damaged; every answer must be the oracle's."""
    import random

    rng = random.Random(4242)
    raws, caps = [], []
    for it in range(400):
        parts = []
        for _ in range(rng.randint(1, 4)):
            n = rng.randint(500, 40000)
            parts.append(_payload(rng, n, rng.randrange(5)))
        strat = rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
        c = zlib.compressobj(rng.choice([1, 4, 6, 9]), zlib.DEFLATED, -15, rng.choice([8, 9]), strat)
        raw = b""
        for p in parts:
            raw += c.compress(p)
            if rng.random() < 0.3:
                raw += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
        raw += c.flush()
        plain_len = sum(len(p) for p in parts)
        r = rng.random()
        if r < 0.2 and len(raw) > 100:
            bad = bytearray(raw)
            bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
            raw = bytes(bad)
        elif r < 0.33 and len(raw) > 100:
            raw = raw[:rng.randint(50, len(raw) - 1)]
        raws.append(raw)
        caps.append(max(plain_len + rng.choice([0, 1, 64]), len(raw)))
    for chunk in ("1024", "3072"):
        monkeypatch.setenv("DEBIG_CHUNK_BYTES", chunk)
        _check(oracle, gpu_device, raws, caps, widths=(0x20,), in_skew=int(chunk) % 7, out_skew=3)
    monkeypatch.delenv("DEBIG_CHUNK_BYTES", raising=False)


def test_batch_call_with_caller_workspace_is_graph_capturable(gpu_device):
    """include/debig_hip.h: debig_hip_inflate_batch_ws allocates nothing and never synchronises, so the
    launches of a step can be captured into a hipGraph and replayed (both the scan / LZ77 pair and
    the chunk-parallel path; the workspace is the caller's)."""
    import torch

    from debigulator_amd import _native as N

    for width, pairs in ((0x10, workload.make_streams("dynamic", 1500, 20000)),
                         (0x20, workload.make_streams("dynamic", 6, 3 << 20))):
        raws = [p[0] for p in pairs]
        b = DeviceBatch.from_streams(raws, [len(p[1]) for p in pairs], device=gpu_device)
        assert b.lib.debig_hip_init(None) == 0  # the table images: nothing is left to allocate inside a call
        b.launch(waves_per_stream=width)  # allocates the batch's workspace, outside the capture
        torch.cuda.synchronize()
        s = torch.cuda.Stream(device=gpu_device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            b.launch(waves_per_stream=width)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            b.launch(waves_per_stream=width)
        for _ in range(2):
            b.d_out.zero_()
            b.d_results.zero_()
            g.replay()
            res = b.results()
            assert (res["good"] == 1).all(), width
            for i in (0, len(pairs) // 2, len(pairs) - 1):
                assert b.output(i, res) == pairs[i][1].tobytes(), (width, i)


def test_default_workspace_is_safe_for_concurrent_callers(gpu_device):
    """include/debig_hip.h: callers that bring no workspace share one cached buffer per device; their
    groups of launches (plan, scan, lz, retry) are serialised on it.  Two host threads on two HIP
    streams hammer debig_hip_inflate_batch_ex (scan / LZ77 pair, internal workspace) with different
    batches at the same time: every stream of every round must still come out right."""
    import ctypes as C
    import threading

    import torch

    batches, plains = [], []
    for kind, seed0 in (("fixed", 100), ("dynamic", 300)):
        pairs = [workload.make_stream(kind, seed0 + i, 20000) for i in range(1200)]
        batches.append(DeviceBatch.from_streams([p[0] for p in pairs], [len(p[1]) + 1 for p in pairs], device=gpu_device))
        plains.append([p[1].tobytes() for p in pairs])
    lib = batches[0].lib
    assert lib.debig_hip_init(None) == 0
    streams = [torch.cuda.Stream(device=gpu_device) for _ in batches]
    errors = []

    def hammer(k):
        b, s = batches[k], streams[k]
        try:
            for _ in range(12):
                rc = lib.debig_hip_inflate_batch_ex(b.d_in.data_ptr(), b.d_out.data_ptr(), b.d_streams.data_ptr(),
                                                    b.d_results.data_ptr(), b.n, 0x10, C.c_void_p(s.cuda_stream))
                if rc:
                    errors.append((k, rc))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    for rnd in range(3):
        for b in batches:
            b.d_out.zero_()
            b.d_results.zero_()
        torch.cuda.synchronize()
        ts = [threading.Thread(target=hammer, args=(k,)) for k in range(len(batches))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        assert not errors, errors
        for b, want in zip(batches, plains):
            res = b.results()
            assert (res["good"] == 1).all(), rnd
            for i in range(0, b.n, 37):
                assert b.output(i, res) == want[i], (rnd, i)


def test_plan_once_execute_many(oracle, gpu_device):
    """include/debig_hip.h: debig_hip_inflate_plan_ws carves the workspace for a set of descriptors,
    debig_hip_inflate_planned_ws runs scan + LZ77 over it any number of times.  A roomy workspace and
    one so small that most streams are handed to the one-kernel path; mixed block types, a damaged
    stream; outputs cleared between the runs; every run must give the oracle's answers."""
    import ctypes as C

    import torch

    rng = random.Random(2024)
    raws, caps = [], []
    for i in range(1500):
        data = _payload(rng, rng.randint(100, 30000), rng.randint(0, 4))
        raw = _zlib_raw(data, rng.choice([0, 1, 6]), rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED]), 9, rng.randint(0, 2), rng)
        if i % 301 == 7:
            raw = raw[: len(raw) // 2] + bytes(8)
        raws.append(raw); caps.append(max(len(data) + 1, len(raw)))
    exp = [oracle.inflate(r, c, want_stats=True) for r, c in zip(raws, caps)]
    b = DeviceBatch.from_streams(raws, caps, device=gpu_device)
    lib = b.lib
    s = torch.cuda.current_stream()
    total_in = sum(len(r) for r in raws)
    for ws_bytes in (int(lib.debig_hip_inflate_workspace_bytes(total_in, b.n)), 3 << 20):
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=gpu_device)
        assert lib.debig_hip_inflate_plan_ws(b.d_streams.data_ptr(), b.n, ws.data_ptr(), ws_bytes, C.c_void_p(s.cuda_stream)) == 0
        for rnd in range(3):
            b.d_out.zero_()
            b.d_results.zero_()
            assert lib.debig_hip_inflate_planned_ws(b.d_in.data_ptr(), b.d_out.data_ptr(), b.d_streams.data_ptr(),
                                                    b.d_results.data_ptr(), b.n, ws.data_ptr(), ws_bytes,
                                                    C.c_void_p(s.cuda_stream)) == 0
            res = b.results()
            host = b.outputs_host()
            for i, (g, f, o, st) in enumerate(exp):
                if st.ub_flags & UB_EXCLUDED:
                    continue
                assert res[i]["good"] == g, (ws_bytes, rnd, i)
                if f is not None:
                    off = int(b.streams_host[i]["out_off"])
                    assert int(res[i]["final_size"]) == f and host[off:off + f].tobytes() == o, (ws_bytes, rnd, i)
    assert lib.debig_hip_inflate_plan_ws(b.d_streams.data_ptr(), 20000, ws.data_ptr(), ws_bytes, None) != 0  # more than one group


def test_invalid_width_is_rejected(gpu_device):
    pairs = workload.make_streams("fixed", 1, 4096)
    b = DeviceBatch.from_streams([pairs[0][0]], [8192], device=gpu_device)
    with pytest.raises(Exception):
        b.launch(waves_per_stream=3)


def test_mass_corruption_stays_in_bounds_and_agrees(oracle, gpu_device):
    """20 000 damaged streams in one launch: every one must agree with the oracle (good flag,
    final size, bytes) and nothing may be written past recipient_size -- the kernel's loops
    are data dependent, so this is also the no-hang / no-fault test."""
    rng = random.Random(2026)
    bases = []
    for k in range(40):
        data = _payload(rng, rng.randint(200, 6000), rng.randint(0, 4))
        strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])
        bases.append((_zlib_raw(data, rng.choice([1, 6, 9]), strat, 9, rng.randint(0, 2), rng), len(data)))
    raws, caps = [], []
    for it in range(20000):
        raw, n = bases[it % len(bases)]
        raw = bytearray(raw)
        mode = rng.randrange(4)
        if mode == 0:
            raw = raw[: rng.randint(5, len(raw))]
        elif mode == 1:
            for _ in range(rng.randint(1, 4)):
                raw[rng.randrange(len(raw))] ^= 1 << rng.randrange(8)
        elif mode == 2:
            i = rng.randrange(len(raw))
            raw[i:i + rng.randint(1, 8)] = bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 8)))
        else:
            raw = raw + bytes(rng.getrandbits(8) for _ in range(rng.randint(1, 64)))
        raws.append(bytes(raw))
        caps.append(max(n * 3 + 64, len(raw)))
    _check(oracle, gpu_device, raws, caps)


def test_golden_corpora_reference_made(gpu_device):
    """Every kernel width against REFERENCE-made vectors (tests/golden, no oracle in between):
    the known-answer streams, the zlib corpus (tail rule Q2 cases included) and the damaged
    streams of corpus_corrupt.json (how the reference fails: src/inflate.c:1427-1434, :1809,
    :1843-1852 with the partial final size)."""
    import hashlib
    import json
    import os

    gold = os.path.join(os.path.dirname(__file__), "golden")
    items = []
    for name in ("kat.json", "corpus_zlib.json", "corpus_corrupt.json"):
        items += json.load(open(os.path.join(gold, name)))
    raws = [bytes.fromhex(k["raw_hex"]) for k in items]
    caps = [k["recipient_size"] for k in items]
    b = DeviceBatch.from_streams(raws, caps, device=gpu_device, out_skew=3)
    for width in WIDTHS:
        b.d_out.zero_()
        b.d_results.zero_()
        b.launch(waves_per_stream=width)
        res = b.results()
        host = b.outputs_host()
        for i, k in enumerate(items):
            off = int(b.streams_host[i]["out_off"])
            assert res[i]["good"] == k["good"], (width, i)
            if k["final"] is None:
                assert res[i]["final_set"] == 0
                continue
            assert int(res[i]["final_size"]) == k["final"], (width, i)
            got = host[off:off + k["final"]].tobytes()
            want = k.get("out_sha256") or hashlib.sha256(bytes.fromhex(k["out_hex"])).hexdigest()
            assert hashlib.sha256(got).hexdigest() == want, (width, i)
            assert not host[off + caps[i]:off + caps[i] + 32].any()


def test_randomised_dynamic_headers(oracle, gpu_device):
    """tests/header_fuzz.py: random prefix codes for the three alphabets and a randomised run-length
    coding of the code-length sequence (runs across the alphabets, runs that reach behind the last
    length, damaged headers) -- the header decoder that looks 64 bit positions up at a time
    (decode_code_lengths, inflate_kernel.inc) against the oracle's serial loop (src/inflate.c:1416-1520),
    every width."""
    import header_fuzz as hf
    cs = hf.cases(600)
    raws = [c[0] + bytes(8) for c in cs]
    caps = [max(2048, len(r) + 1) for r in raws]
    _check(oracle, gpu_device, raws, caps)


def test_queued_dispatch_few_workgroups_many_launches(oracle, gpu_device, monkeypatch):
    """DEBIG_WAVES_SPLIT_QUEUED with far fewer resident workgroups than streams (DEBIG_SPLIT_WORKGROUPS = 5
    for 300 streams of very different cost, alternating) and the same plan executed several times: the
    queue counter is never reset, every launch must hand every stream out exactly once."""
    monkeypatch.setenv("DEBIG_SPLIT_WORKGROUPS", "5")
    rng = random.Random(77)
    raws, plains = [], []
    for i in range(300):
        p = _payload(rng, 200 if i % 2 else rng.randint(20000, 60000), i % 5)
        plains.append(p)
        raws.append(_zlib_raw(p, level=rng.choice([0, 1, 6, 9])) + bytes(8))
    caps = [max(len(p) + 16, len(r) + 1) for p, r in zip(plains, raws)]
    b = DeviceBatch.from_streams(raws, caps, device=gpu_device)
    for _ in range(4):
        b.d_out.zero_()
        b.d_results.zero_()
        b.launch(waves_per_stream=0x11)
        res = b.results()
        assert (res["good"] == 1).all()
        for i in (0, 1, 2, 150, 298, 299):
            assert b.output(i, res) == plains[i]
        want = [oracle.inflate(r, c)[:2] for r, c in zip(raws[:40], caps[:40])]
        assert [(int(res[i]["good"]), int(res[i]["final_size"])) for i in range(40)] == want
