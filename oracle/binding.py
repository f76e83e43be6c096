"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the parity oracle.

Loads ``oracle/liboracle.so`` (our CPU restatement, ``debig_oracle.c``) and, when
it has been built, ``oracle/_ref/libdebig_ref_{A,B}.so`` (the UNMODIFIED
reference compiled in place by ``oracle/Makefile``; see ``ref_harness.c``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``debigulator_amd``) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcStats(C.Structure):
    _fields_ = [
        ("ub_flags", C.c_uint32),
        ("n_blocks", C.c_uint32),
        ("n_stored", C.c_uint32),
        ("n_fixed", C.c_uint32),
        ("n_dynamic", C.c_uint32),
        ("n_symbols", C.c_uint64),
        ("n_matches", C.c_uint64),
        ("bits_consumed", C.c_uint64),
        ("tail_gate_fired", C.c_uint32),
    ]


PNG_STRICT, PNG_NO_CRC, PNG_ASSERTS_OFF = 1, 2, 4
NOT_SET = 0xFFFFFFFFFFFFFFFF


def build(ref=True):
    """(re)build liboracle.so and, if the reference tree exists, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    return np.ascontiguousarray(a)


class Oracle:
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.L = C.CDLL(path)
        L.orc_inflate_ex.restype = C.c_uint32
        L.orc_inflate_ex.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                     C.c_void_p, C.c_void_p, C.POINTER(OrcStats)]
        L.orc_decode_png.restype = C.c_uint32
        L.orc_decode_png.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.c_uint32, C.c_uint32]
        L.orc_png_get_width_height.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32),
                                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        L.orc_decode_gz.restype = C.c_uint32
        L.orc_decode_gz.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                    C.POINTER(C.c_uint64)]
        L.orc_gz_locate.restype = C.c_uint32
        L.orc_gz_locate.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32)]
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]

    def inflate(self, data, recipient_size=None, want_stats=False):
        """-> (good, final or None if untouched, bytes produced[, stats])"""
        d = _u8(data)
        if recipient_size is None:
            recipient_size = max(len(d) * 40 + 4096, 1 << 16)
        out = np.zeros(recipient_size + 8, dtype=np.uint8)
        fin = C.c_uint64(NOT_SET)
        fs = C.c_uint32(0)
        st = OrcStats()
        good = self.L.orc_inflate_ex(d.ctypes.data, len(d), out.ctypes.data, recipient_size,
                                     C.byref(fin), C.byref(fs), None, None, C.byref(st))
        final = fin.value if fs.value else None
        produced = out[: (final or 0)].tobytes()
        return (good, final, produced, st) if want_stats else (good, final, produced)

    def png_wh(self, data):
        d = _u8(data)
        w, h, g = C.c_uint32(), C.c_uint32(), C.c_uint8()
        self.L.orc_png_get_width_height(d.ctypes.data, len(d), C.byref(w), C.byref(h), C.byref(g))
        return w.value, h.value, g.value

    def decode_png(self, data, wm_size=120_000_000, flags=0, prior=None, rgba_size=None):
        d = _u8(data)
        w, h, g = self.png_wh(d)
        n = (w * h * 4) if rgba_size is None else rgba_size
        out = np.zeros(max(n, 1), dtype=np.uint8) if prior is None else np.array(prior, dtype=np.uint8)
        good = self.L.orc_decode_png(d.ctypes.data, len(d), out.ctypes.data, n, wm_size, flags)
        return good, out[:n]

    def decode_gz(self, data, out_cap=None):
        d = _u8(data)
        if out_cap is None:
            out_cap = len(d) * 40 + (1 << 20)
        out = np.zeros(out_cap, dtype=np.uint8)
        n = C.c_uint64(0)
        good = self.L.orc_decode_gz(d.ctypes.data, len(d), out.ctypes.data, out_cap, C.byref(n))
        return good, out[: min(n.value, out_cap)].tobytes(), n.value

    def gz_locate(self, data):
        d = _u8(data)
        off, ln = C.c_uint32(), C.c_uint32()
        ok = self.L.orc_gz_locate(d.ctypes.data, len(d), C.byref(off), C.byref(ln))
        return ok, off.value, ln.value

    def crc32(self, data, crc=0xFFFFFFFF):
        d = _u8(data)
        return self.L.orc_crc32(crc, d.ctypes.data, len(d))


def ref_available(variant="A"):
    return os.path.exists(os.path.join(_HERE, "_ref", f"libdebig_ref_{variant}.so"))


class Reference:
    """The compiled reference (A = silent, asserts on; B = silent, asserts off)."""

    def __init__(self, variant="A"):
        self.variant = variant
        L = self.L = C.CDLL(os.path.join(_HERE, "_ref", f"libdebig_ref_{variant}.so"))
        L.refh_inflate.restype = C.c_uint32
        L.refh_inflate.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                   C.POINTER(C.c_uint64), C.c_uint64]
        L.refh_decode_png.restype = C.c_uint32
        L.refh_decode_png.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_uint32, C.c_uint32]
        L.refh_png_wh.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        L.refh_decode_gz.restype = C.c_uint32
        L.refh_decode_gz.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]

    def inflate(self, data, recipient_size=None, scratch=0):
        d = _u8(data)
        if recipient_size is None:
            recipient_size = max(len(d) * 40 + 4096, 1 << 16)
        out = np.zeros(recipient_size + 8, dtype=np.uint8)
        fin = C.c_uint64(0)
        good = self.L.refh_inflate(d.ctypes.data, len(d), out.ctypes.data, recipient_size,
                                   C.byref(fin), scratch)
        final = None if fin.value == NOT_SET else fin.value
        return good, final, out[: (final or 0)].tobytes()

    def png_wh(self, data):
        d = _u8(data)
        w, h, g = C.c_uint32(), C.c_uint32(), C.c_uint8()
        self.L.refh_png_wh(d.ctypes.data, len(d), C.byref(w), C.byref(h), C.byref(g))
        return w.value, h.value, g.value

    def decode_png(self, data, wm_size=120_000_000, tid=1, prior=None, rgba_size=None):
        d = _u8(data)
        w, h, g = self.png_wh(d)
        n = (w * h * 4) if rgba_size is None else rgba_size
        out = np.zeros(max(n, 1), dtype=np.uint8) if prior is None else np.array(prior, dtype=np.uint8)
        good = self.L.refh_decode_png(d.ctypes.data, len(d), out.ctypes.data, n, wm_size, tid)
        return good, out[:n]

    def decode_gz(self, data, expect_size):
        d = _u8(data)
        out = np.zeros(expect_size + 8, dtype=np.uint8)
        good = self.L.refh_decode_gz(d.ctypes.data, len(d), out.ctypes.data, expect_size)
        return good, out[:expect_size].tobytes()
