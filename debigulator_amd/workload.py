"""Deterministic synthetic workloads (BASELINE.json configs).

Thin ctypes wrapper over tools/streamgen.c: own PRNG + own DEFLATE encoders, so the
same bytes regenerate on every machine.  Workload tooling for tests/ and bench.py.
"""
import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "streamgen.c")
LIB = os.path.join(ROOT, "tools", "libstreamgen.so")
SEED0 = 0xDEB16

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", LIB, SRC])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        u64, u32, vp = C.c_uint64, C.c_uint32, C.c_void_p
        L.sg_payload_text.argtypes = [u64, vp, u64]
        L.sg_payload_random.argtypes = [u64, vp, u64]
        L.sg_payload_image.argtypes = [u64, u32, u32, u32, u32, vp]
        L.sg_enc_stored.restype = u64
        L.sg_enc_stored.argtypes = [vp, u64, vp, u64, u32]
        L.sg_enc_fixed.restype = u64
        L.sg_enc_fixed.argtypes = [vp, u64, vp, u64, u32, u32]
        L.sg_enc_dynamic.restype = u64
        L.sg_enc_dynamic.argtypes = [vp, u64, vp, u64, u32, u32, u32]
        L.sg_wrap_gzip.restype = u64
        L.sg_wrap_gzip.argtypes = [vp, u64, vp, u64, vp, u64]
        L.sg_wrap_png.restype = u64
        L.sg_wrap_png.argtypes = [vp, u64, vp, u64, u32, u32, u32, vp, u32, u32, vp, u64]
        L.sg_payload_image_mix.argtypes = [u64, u32, u32, u32, u32, u32, u32, vp]
        L.sg_png_filter.argtypes = [vp, u32, u32, u32, u32, vp]
        L.sg_crc32.restype = u32
        L.sg_crc32.argtypes = [u32, vp, u64]
        _lib = L
    return _lib


def payload(kind, seed, n):
    out = np.zeros(n, dtype=np.uint8)
    if kind == "text":
        lib().sg_payload_text(seed, out.ctypes.data, n)
    elif kind == "random":
        lib().sg_payload_random(seed, out.ctypes.data, n)
    else:
        raise ValueError(kind)
    return out


def encode(kind, plain, tokens_per_block=0, depth=8, eob_min_bits=8, stored_block=65535):
    """plain: uint8 array -> raw DEFLATE bytes"""
    L = lib()
    n = len(plain)
    dst = np.zeros(n + n // 4 + 1024, dtype=np.uint8)
    if kind == "stored":
        c = L.sg_enc_stored(plain.ctypes.data, n, dst.ctypes.data, dst.size, stored_block)
    elif kind == "fixed":
        c = L.sg_enc_fixed(plain.ctypes.data, n, dst.ctypes.data, dst.size, tokens_per_block, depth)
    elif kind == "dynamic":
        c = L.sg_enc_dynamic(plain.ctypes.data, n, dst.ctypes.data, dst.size, tokens_per_block, depth,
                             eob_min_bits)
    else:
        raise ValueError(kind)
    if c == 0:
        raise RuntimeError("encoder overflow")
    return dst[:c].tobytes()


def make_stream(kind, index, size=65536):
    """cfg2 stream `index`: (raw deflate bytes, plain bytes).  stored -> random payload,
    fixed/dynamic -> text-like payload."""
    seed = SEED0 + index
    if kind == "png":  # Paeth-filtered scanlines of a noisy RGBA image, dynamic Huffman (config 4's data)
        w = 128
        h = size // (4 * w + 1)
        pix = np.zeros(w * h * 4, dtype=np.uint8)
        lib().sg_payload_image(seed, w, h, 4, 24, pix.ctypes.data)
        plain = np.zeros(size, dtype=np.uint8)
        lib().sg_png_filter(pix.ctypes.data, w, h, 4, 4, plain.ctypes.data)
        return encode("dynamic", plain), plain
    plain = payload("random" if kind == "stored" else "text", seed, size)
    return encode(kind, plain), plain


def make_streams(kind, count, size=65536, first=0, threads=8):
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda i: make_stream(kind, i, size), range(first, first + count)))


def gzip_member(raw, plain):
    L = lib()
    dst = np.zeros(len(raw) + 32, dtype=np.uint8)
    r = np.frombuffer(raw, dtype=np.uint8)
    p = np.ascontiguousarray(plain)
    n = L.sg_wrap_gzip(r.ctypes.data, len(raw), p.ctypes.data, len(p), dst.ctypes.data, dst.size)
    return dst[:n].tobytes()


CFG4_NOISE = (1, 2, 48)  # config 4: amplitude 1, amplitude 2 for HI/256 of the samples -> ratio about 3:1


def make_png(seed, w, h, ct=6, ftype=4, noise=8, enc="dynamic", idat_chunk=65536, palette=None):
    """Synthetic PNG: smooth image + noise, forward-filtered, own DEFLATE encoder.
    noise: amplitude, or a (lo, hi, hi_per_256) mix (CFG4_NOISE lands at S/C of about 3).
    Returns (png bytes, defiltered pixel bytes [h, w*bpp])."""
    L = lib()
    bpp = {6: 4, 2: 3, 3: 1}[ct]
    pix = np.zeros(w * h * bpp, dtype=np.uint8)
    if isinstance(noise, tuple):
        L.sg_payload_image_mix(seed, w, h, bpp, noise[0], noise[1], noise[2], pix.ctypes.data)
    else:
        L.sg_payload_image(seed, w, h, bpp, noise, pix.ctypes.data)
    filt = np.zeros(h * (w * bpp + 1), dtype=np.uint8)
    L.sg_png_filter(pix.ctypes.data, w, h, bpp, ftype, filt.ctypes.data)
    raw = encode(enc, filt)
    r = np.frombuffer(raw, dtype=np.uint8)
    dst = np.zeros(len(raw) + 4096 + 12 * (len(raw) // max(idat_chunk, 1) + 4), dtype=np.uint8)
    pal_ptr, npal = None, 0
    if ct == 3:
        if palette is None:
            palette = np.arange(768, dtype=np.uint32).astype(np.uint8)
        palette = np.ascontiguousarray(palette, dtype=np.uint8)
        pal_ptr, npal = palette.ctypes.data, len(palette) // 3
    n = L.sg_wrap_png(r.ctypes.data, len(raw), filt.ctypes.data, len(filt), w, h, ct, pal_ptr, npal,
                      idat_chunk, dst.ctypes.data, dst.size)
    if n == 0:
        raise RuntimeError("png wrap overflow")
    return dst[:n].tobytes(), pix.reshape(h, w * bpp)
