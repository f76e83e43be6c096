"""Host-side mirror of the reference's interface, over the C drop-in layer.

Same names, argument meaning and error behaviour as the reference headers
(src/inflate.h:51-60, src/decode_png.h:69-103, src/decode_gz.h:23-38); every call goes
through the C-ABI of libdebigulator_hip.so (include/inflate.h, decode_png.h, decode_gz.h)
and runs on the GPU.  No CPU fallback exists: without the library or a GPU these raise.
"""
import ctypes as C

import numpy as np

from . import _native as N

_configured = False
NOT_SET = 0xFFFFFFFFFFFFFFFF


class DecodedData(C.Structure):
    _fields_ = [("data", C.c_void_p), ("data_size", C.c_uint32), ("good", C.c_uint32)]


def _lib():
    global _configured
    L = N.lib()
    if not _configured:
        vp, u8p, u32, u64 = C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint64
        L.debig_inflate.restype = None
        L.debig_inflate.argtypes = [vp, u64, C.POINTER(u64), vp, u64, vp, u64, C.POINTER(u32), u32]
        L.debig_inflate_batch.restype = C.c_int
        L.debig_inflate_batch.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32]
        L.decode_png_init.restype = None
        L.decode_png_init.argtypes = [vp, vp, vp, vp, u32, u32]
        L.decode_png_deinit.argtypes = [u32]
        L.decode_png_get_width_height.argtypes = [vp, u64, C.POINTER(u32), C.POINTER(u32), u8p]
        L.decode_png.restype = None
        L.decode_png.argtypes = [vp, u64, vp, u64, u32, u8p]
        L.debig_decode_png_batch.restype = C.c_int
        L.debig_decode_png_batch.argtypes = [vp, vp, vp, vp, vp, u32, u32]
        L.init_decode_gz.argtypes = [vp, vp, vp]
        L.decode_gz.restype = C.POINTER(DecodedData)
        L.decode_gz.argtypes = [vp, u32]
        L.init_PNG_decoder.argtypes = [vp]
        L.get_PNG_width_height.argtypes = [vp, u64, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
        L.decode_PNG.argtypes = [vp, u64, vp, u64, C.POINTER(u32)]
        _configured = True
    return L


_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]
_libc.free.argtypes = [C.c_void_p]


def _fn(name):
    return C.cast(getattr(_libc, name), C.c_void_p)


def _u8(b):
    return np.ascontiguousarray(np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b)


def inflate(data, recipient_size, thread_id=0):
    """reference inflate(): -> (good, final_recipient_size or None if untouched, bytes)"""
    L = _lib()
    d = _u8(data)
    out = np.zeros(max(recipient_size, 1), dtype=np.uint8)
    fin = C.c_uint64(NOT_SET)
    good = C.c_uint32(7)
    L.debig_inflate(out.ctypes.data, recipient_size, C.byref(fin), None, 0, d.ctypes.data, len(d), C.byref(good), thread_id)
    final = None if fin.value == NOT_SET else fin.value
    return good.value, final, out[: min(final or 0, recipient_size)].tobytes()


def inflate_batch(datas, recipient_sizes, thread_id=0):
    L = _lib()
    n = len(datas)
    ins = [_u8(d) for d in datas]
    outs = [np.zeros(max(c, 1), dtype=np.uint8) for c in recipient_sizes]
    in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins])
    out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    in_sizes = (C.c_uint64 * n)(*[len(a) for a in ins])
    caps = (C.c_uint64 * n)(*recipient_sizes)
    finals = (C.c_uint64 * n)(*([NOT_SET] * n))
    goods = (C.c_uint32 * n)()
    rc = L.debig_inflate_batch(out_ptrs, caps, finals, in_ptrs, in_sizes, goods, n, thread_id)
    N.check(rc, "debig_inflate_batch")
    res = []
    for i in range(n):
        final = None if finals[i] == NOT_SET else finals[i]
        res.append((goods[i], final, outs[i][: min(final or 0, recipient_sizes[i])].tobytes()))
    return res


_png_inited = set()


def decode_png_init(working_memory_size=120_000_000, thread_id=0):
    _lib().decode_png_init(_fn("malloc"), _fn("free"), _fn("memset"), _fn("memcpy"), working_memory_size, thread_id)
    _png_inited.add(thread_id)


def decode_png_get_width_height(data):
    d = _u8(data)
    w, h, g = C.c_uint32(), C.c_uint32(), C.c_uint8()
    _lib().decode_png_get_width_height(d.ctypes.data, len(d), C.byref(w), C.byref(h), C.byref(g))
    return w.value, h.value, g.value


def decode_png(data, thread_id=0, rgba_size=None):
    """reference decode_png(): -> (good, RGBA uint8 array of 4*w*h)"""
    if thread_id not in _png_inited:
        decode_png_init(thread_id=thread_id)
    d = _u8(data)
    w, h, _ = decode_png_get_width_height(d)
    n = w * h * 4 if rgba_size is None else rgba_size
    out = np.zeros(max(n, 1), dtype=np.uint8)
    good = C.c_uint8(7)
    _lib().decode_png(d.ctypes.data, len(d), out.ctypes.data, n, thread_id, C.byref(good))
    return good.value, out[:n]


def decode_png_batch(datas, thread_id=0):
    if thread_id not in _png_inited:
        decode_png_init(thread_id=thread_id)
    L = _lib()
    n = len(datas)
    ins = [_u8(d) for d in datas]
    sizes = []
    for a in ins:
        w, h, _ = decode_png_get_width_height(a)
        sizes.append(w * h * 4)
    outs = [np.zeros(max(s, 1), dtype=np.uint8) for s in sizes]
    in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins])
    in_sizes = (C.c_uint64 * n)(*[len(a) for a in ins])
    out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    out_sizes = (C.c_uint64 * n)(*sizes)
    goods = (C.c_uint8 * n)()
    rc = L.debig_decode_png_batch(in_ptrs, in_sizes, out_ptrs, out_sizes, goods, n, thread_id)
    N.check(rc, "debig_decode_png_batch")
    return [(goods[i], outs[i][: sizes[i]]) for i in range(n)]


_gz_inited = False


def decode_gz(data):
    """reference decode_gz(): -> (good, bytes) ; None if the library returned NULL"""
    global _gz_inited
    L = _lib()
    if not _gz_inited:
        L.init_decode_gz(_fn("malloc"), _fn("memset"), _fn("memcpy"))
        _gz_inited = True
    d = _u8(data).copy()
    p = L.decode_gz(d.ctypes.data, len(d))
    if not p:
        return None
    dd = p.contents
    good, size = dd.good, dd.data_size
    out = C.string_at(dd.data, size) if (good and dd.data) else b""
    if dd.data:
        _libc.free(dd.data)
    _libc.free(C.cast(p, C.c_void_p))
    return good, out


def decode_gz_batch(datas, out_caps, verify_trailer=True):
    """n gzip members in one launch -> [(good, bytes, trailer_ok)]; trailer_ok = CRC-32 and ISIZE
    of the member match the decompressed bytes (checked on the GPU; the reference never checks)."""
    L = _lib()
    L.debig_decode_gz_batch_ex.restype = C.c_int
    L.debig_decode_gz_batch_ex.argtypes = [C.c_void_p] * 7 + [C.c_uint32]
    n = len(datas)
    ins = [_u8(d) for d in datas]
    outs = [np.zeros(max(c, 1), dtype=np.uint8) for c in out_caps]
    in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins])
    in_sizes = (C.c_uint32 * n)(*[len(a) for a in ins])
    out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    caps = (C.c_uint64 * n)(*out_caps)
    sizes = (C.c_uint64 * n)()
    goods = (C.c_uint32 * n)()
    tok = (C.c_uint32 * n)()
    rc = L.debig_decode_gz_batch_ex(in_ptrs, in_sizes, out_ptrs, caps, sizes, goods, tok if verify_trailer else None, n)
    N.check(rc, "debig_decode_gz_batch_ex")
    return [(goods[i], outs[i][: sizes[i]].tobytes(), tok[i]) for i in range(n)]


GZ_STATUS = {0: "ok", 1: "header", 2: "truncated", 3: "inflate", 4: "output_full", 5: "crc", 6: "isize", 7: "trailing"}


def gunzip_batch(datas, out_caps):
    """RFC 1952-complete gunzip of n files (include/decode_gz.h: debig_gunzip_batch): every
    member, every optional header field, CRC-32/ISIZE verified on the GPU.
    -> [(status, bytes, n_members)], status as in GZ_STATUS."""
    L = _lib()
    L.debig_gunzip_batch.restype = C.c_int
    L.debig_gunzip_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32]
    n = len(datas)
    ins = [_u8(d) for d in datas]
    outs = [np.zeros(max(c, 1), dtype=np.uint8) for c in out_caps]
    in_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in ins])
    in_sizes = (C.c_uint64 * n)(*[len(d) for d in datas])
    out_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    caps = (C.c_uint64 * n)(*out_caps)
    sizes = (C.c_uint64 * n)()
    status = (C.c_uint32 * n)()
    members = (C.c_uint32 * n)()
    rc = L.debig_gunzip_batch(in_ptrs, in_sizes, out_ptrs, caps, sizes, status, members, n)
    N.check(rc, "debig_gunzip_batch")
    return [(status[i], outs[i][: sizes[i]].tobytes(), members[i]) for i in range(n)]
