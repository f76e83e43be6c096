"""CPU: the C-ABI shared library loads and exports every symbol include/*.h declares
(no compute call is made: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header_text):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    names = set()
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text):
        name = m.group(1)
        before = text[: m.start()].rstrip()
        # a function declaration's name is preceded by a type token or '*', not by '(' or ','
        if name in ("defined", "sizeof", "__attribute__", "visibility") or before.endswith(("(", ",", "#define")):
            continue
        if re.search(r"(\*|\b[A-Za-z_][A-Za-z0-9_]*)\s*$", before) and not before.endswith(("return", "typedef")):
            names.add(name)
    return names


def test_library_exports_every_declared_symbol(native_lib):
    declared = set()
    for h in ("debig_hip.h", "inflate.h", "decode_png.h", "decode_gz.h"):
        declared |= _declared_functions(open(os.path.join(ROOT, "include", h)).read())
    # `inflate` is a macro alias for debig_inflate (zlib owns the plain symbol); function-pointer
    # parameter names are not functions
    declared -= {"inflate", "malloc_funcptr", "arg_memset_func", "arg_memcpy_func", "arg_free_funcptr",
                 "arg_memset_funcptr", "free_funcptr"}
    assert {"debig_hip_inflate_batch", "debig_hip_inflate_batch_ex", "debig_hip_png_defilter_batch", "debig_inflate", "inflate_init",
            "inflate_destroy", "decode_png", "decode_png_init", "decode_png_deinit",
            "decode_png_get_width_height", "decode_gz", "init_decode_gz", "decode_PNG", "init_PNG_decoder",
            "get_PNG_width_height", "debig_inflate_batch", "debig_decode_png_batch",
            "debig_decode_gz_batch", "debig_gunzip_batch"} <= declared
    missing = [n for n in sorted(declared) if not hasattr(native_lib, n)]
    assert not missing, missing


def test_plain_inflate_symbol_is_not_exported(native_lib):
    """zlib exports `inflate`; libamdhip64/librccl/python load zlib.  Ours must not shadow it."""
    from debigulator_amd import _native

    out = os.popen(f"nm -D --defined-only {_native.LIB_PATH}").read()
    syms = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert "inflate" not in syms and "debig_inflate" in syms


def test_struct_layouts_match_header():
    from debigulator_amd import _native as N
    from debigulator_amd.batch import RESULT_DTYPE, STREAM_DTYPE

    assert ctypes.sizeof(N.DebigStream) == 56 == STREAM_DTYPE.itemsize
    assert ctypes.sizeof(N.DebigResult) == 72 == RESULT_DTYPE.itemsize
    assert ctypes.sizeof(N.DebigPngImage) == 56


def test_gzip_header_parser_host_only(native_lib):
    """debig_gz_parse_header (include/decode_gz.h): RFC 1952 member headers with every optional
    field, the BGZF size subfield, truncation and rejects -- pure host code, no GPU needed."""
    import ctypes as C
    import struct
    import zlib

    f = native_lib.debig_gz_parse_header
    f.restype = C.c_uint32
    f.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]

    def parse(b, avail=None):
        hl, ms = C.c_uint64(77), C.c_uint64(77)
        st = f(b, len(b) if avail is None else avail, C.byref(hl), C.byref(ms))
        return st, hl.value, ms.value

    base = bytes([31, 139, 8, 0, 0, 0, 0, 0, 0, 255])
    assert parse(base + b"xx") == (0, 10, 0)
    # FEXTRA (two subfields, the second is BGZF's BC with BSIZE = 0x1234) + FNAME + FCOMMENT + FHCRC
    extra = b"XY\x03\x00abc" + b"BC\x02\x00\x34\x12"
    h = bytes([31, 139, 8, 2 | 4 | 8 | 16, 0, 0, 0, 0, 0, 255]) + struct.pack("<H", len(extra)) + extra + b"name\0" + b"a comment\0"
    h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
    assert parse(h + b"payload") == (0, len(h), 0x1234 + 1)
    # every truncation point of that header is reported as truncated, never read past
    for cut in range(10, len(h)):
        assert parse(h, cut)[0] == 2, cut
    assert parse(h[:9], 9)[0] == 2 and parse(b"", 0)[0] == 2
    # rejects: wrong magic, wrong method, reserved flag bits
    assert parse(b"PK" + base[2:])[0] == 1
    assert parse(bytes([31, 139, 7]) + base[3:])[0] == 1
    assert parse(bytes([31, 139, 8, 0x20]) + base[4:])[0] == 1


def test_dispatch_plan_for_skewed_batches():
    """debigulator_amd.batch.plan_batch (same rule as csrc/host/debig_ctx.h: debig_plan_batch):
    only batches of 513..1024 streams whose largest quarter holds half of the input are
    reordered longest first, and results come back in the caller's order."""
    import numpy as np
    from debigulator_amd.batch import STREAM_DTYPE, plan_batch

    def mk(lens):
        s = np.zeros(len(lens), dtype=STREAM_DTYPE)
        s["in_len"] = lens
        return s

    assert plan_batch(mk([1000] * 1024)) == (None, 0)                 # uniform: own order, library default
    assert plan_batch(mk([10**6] * 100 + [1000] * 100))[0] is None      # too few streams
    order, waves = plan_batch(mk([1000, 10**6, 1000, 1000] * 200))     # 800 streams, a quarter of them long
    assert waves == 4 and list(order[:3]) == [1, 5, 9] and sorted(order) == list(range(800))
    assert plan_batch(mk([1000, 10**6] * 400)) == (None, 0)             # half long: not skewed enough
    assert plan_batch(mk([1000] * 2000))[0] is None                     # beyond the range


def _png_file(w, h, ct, idat_payload, extra=b""):
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    ihdr = struct.pack(">IIBBBBB", w, h, 8, ct, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + extra + chunk(b"IDAT", idat_payload) + chunk(b"IEND", b"")


def test_png_probe_rejects_wrapped_dimensions_host_only(native_lib, oracle):
    """w*h*4 is uint32 arithmetic in the reference (src/decode_png.c:965-985), so
    w=32769, h=32768 'needs' a 128 KiB buffer.  The reference ends with out_good = 0 (its de-filter
    loop runs into the end of the buffer, :1459); the product must reject the file on the host --
    its kernels iterate the real w and h.  Pure host code: no GPU needed."""
    import ctypes as C
    import zlib

    f = native_lib.debig_png_probe
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]

    def probe(png, rgba_size, wm=120_000_000):
        w, h, est, z = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        ok = f(png, len(png), rgba_size, wm, C.byref(w), C.byref(h), C.byref(est), C.byref(z))
        return ok, w.value, h.value, est.value, z.value

    # a well-formed small file is accepted, with the sizes the reference computes
    filt = b"".join(b"\x00" + bytes(range(16)) for _ in range(4))
    zl = zlib.compress(filt, 6)
    good_png = _png_file(4, 4, 6, zl)
    ok, w, h, est, z = probe(good_png, 64)
    assert (ok, w, h, est, z) == (1, 4, 4, 4 * 4 * 4 + 4 + 1, len(zl) - 2 - 4)
    assert oracle.decode_png(good_png, rgba_size=64)[0] == 1
    assert probe(good_png, 63)[0] == 0  # rgba_values_size must be exactly 4wh

    # wrapped: (32769 * 32768 * 4) mod 2^32 == 131072
    assert (32769 * 32768 * 4) % (1 << 32) == 131072
    for (ww, hh, size) in ((32769, 32768, 131072), (65536, 65536, 0), (16385, 65536, 262144)):
        bad = _png_file(ww, hh, 6, zl)
        assert probe(bad, size)[0] == 0, (ww, hh)
        if size:  # the oracle restates the reference: same verdict
            assert oracle.decode_png(bad, rgba_size=size)[0] == 0

    # an IDAT payload too short for inflate()'s gates (zsize would wrap to ~4 GiB) is refused
    # before any arena is sized from it
    for payload in (zl[:2], zl[:3], zl[:5], zl[:6]):
        tiny = _png_file(4, 4, 6, payload)
        assert probe(tiny, 64)[0] == 0
        assert oracle.decode_png(tiny, rgba_size=64)[0] == 0


def test_multi_device_shard_partition_host_only(native_lib):
    """debig_shard_round_robin (include/inflate.h): stream i -> GPU i mod n (BASELINE config 5); the
    shares of all devices partition the batch.  Pure host code."""
    import ctypes as C

    f = native_lib.debig_shard_round_robin
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    for n, nd in ((0, 4), (1, 8), (7, 2), (65536, 8), (1000, 3), (5, 16)):
        seen = []
        for d in range(nd):
            cnt = f(n, nd, d, None)
            buf = (C.c_uint32 * max(cnt, 1))()
            assert f(n, nd, d, buf) == cnt
            got = list(buf[:cnt])
            assert got == list(range(d, n, nd))
            seen += got
        assert sorted(seen) == list(range(n))
    assert f(10, 0, 0, None) == 0 and f(10, 4, 4, None) == 0


def test_compat_archive_has_the_literal_inflate_symbol(native_lib):
    """SURVEY 8b rule 3: the plain `inflate` symbol exists ONLY in the optional static archive
    (debigulator_amd/lib/libdebig_compat.a), as a forwarder to debig_inflate."""
    from debigulator_amd import _native

    ar = os.path.join(os.path.dirname(_native.LIB_PATH), "libdebig_compat.a")
    assert os.path.exists(ar), "python -m debigulator_amd.build makes it"
    out = os.popen(f"nm {ar}").read()
    assert " T inflate" in out and " U debig_inflate" in out


def test_dispatch_plan_rules():
    """debigulator_amd.batch.plan_batch mirrors csrc/host/debig_ctx.h (debig_pick_waves /
    debig_plan_batch): few streams that are large on average, or thousands with a very large one
    among them, go through chunk tasks; a skewed batch of 513..1024 is launched 4-wide, most bytes touched first."""
    import numpy as np

    from debigulator_amd import _native as N
    from debigulator_amd.batch import STREAM_DTYPE, plan_batch

    def streams(lens):
        s = np.zeros(len(lens), dtype=STREAM_DTYPE)
        s["in_len"] = lens
        return s

    assert plan_batch(streams([2 << 20] * 64)) == (None, N.WAVES_CHUNKED)
    assert plan_batch(streams([400 << 10] * 256)) == (None, 0)                  # 0.4 MiB on average: workgroups
    # ... unless the caller says the streams are filtered image rows (DEBIG_STREAM_IMAGE_ROWS; decode_png does) and
    # EVERY stream is long: chunk tasks from 256 KiB on, up to 512 streams
    rows = streams([400 << 10] * 256)
    rows["flags"] = N.STREAM_IMAGE_ROWS
    assert plan_batch(rows) == (None, N.WAVES_CHUNKED)
    rows["in_len"][7] = 30000
    assert plan_batch(rows) == (None, 0)
    one = streams([971497])
    one["flags"] = N.STREAM_IMAGE_ROWS
    assert plan_batch(one) == (None, N.WAVES_CHUNKED)
    assert plan_batch(streams([20000] * 2000 + [8 << 20])) == (None, N.WAVES_CHUNKED)
    assert plan_batch(streams([20000] * 2000 + [1 << 20])) == (None, 0x41)    # a few large among thousands: those 4-wide
    assert plan_batch(streams([20000] * 2000 + [100000])) == (None, 0)
    # 513..1024 skewed streams that are also >= 1 MiB on average: chunk tasks win, no reordering (as in C)
    assert plan_batch(streams([200 << 10] * 600 + [6 << 20] * 200)) == (None, N.WAVES_CHUNKED)
    assert plan_batch(streams([65536] * 8192)) == (None, 0)
    order, waves = plan_batch(streams([1000] * 600 + [500000] * 200))
    assert waves == 4 and list(order[:3]) == [600, 601, 602]
    # ... and the order goes by the bytes a stream touches (in_len + out_cap): a small input that
    # decodes to megabytes starts with the long ones
    s = streams([1000] * 600 + [500000] * 200)
    s["out_cap"] = s["in_len"] * 3
    s["out_cap"][5] = 4 << 20
    order, waves = plan_batch(s)
    assert waves == 4 and list(order[:3]) == [5, 600, 601]
