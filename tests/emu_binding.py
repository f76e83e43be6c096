"""TEST TOOLING: run the product's HIP kernels on the CPU lock-step emulator
(tools/simt_emu).  Same structs as include/debig_hip.h."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tools", "simt_emu")


class DebigStream(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint64), ("out_off", C.c_uint64),
                ("out_cap", C.c_uint64), ("p2_s0", C.c_int64), ("p2_est", C.c_uint64),
                ("p2_on", C.c_uint32), ("flags", C.c_uint32)]


class DebigResult(C.Structure):
    _fields_ = [("final_size", C.c_uint64), ("good", C.c_uint32), ("status", C.c_uint32),
                ("final_set", C.c_uint32), ("n_blocks", C.c_uint32), ("n_windows", C.c_uint32),
                ("n_rounds", C.c_uint32), ("prof", C.c_uint32 * 8),
                ("in_end_bits", C.c_uint64)]


def load_emu(asan=False, variant=None):
    name = "libdebig_emu_asan.so" if asan else "libdebig_emu.so"
    if variant:  # e.g. "sweep": tools/simt_emu/Makefile
        name = "libdebig_emu_%s.so" % variant
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, name])
    # DEBIG_EMU_LIB: a variant build of the same sources (e.g. -DDEBIG_NEAR_SWEEP_MIN=1: every span through the sweep)
    L = C.CDLL(os.environ.get("DEBIG_EMU_LIB") or os.path.join(EMU_DIR, name))
    L.emu_inflate_batch.restype = C.c_int
    L.emu_inflate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.emu_inflate_batch_cls.restype = C.c_int
    L.emu_inflate_batch_cls.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_uint32]
    L.emu_inflate_mw_batch.restype = C.c_int
    L.emu_inflate_mw_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                       C.c_uint32]
    L.emu_inflate_split_batch.restype = C.c_int
    L.emu_inflate_split_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                          C.POINTER(C.c_uint32)]
    L.emu_inflate_chunked_batch.restype = C.c_int
    L.emu_inflate_chunked_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                            C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    return L


CHUNKED = 0x20  # nw code of the chunk-parallel path (include/debig_hip.h: DEBIG_WAVES_CHUNKED)
SPLIT = 0x10  # nw code of the scan / LZ77 kernel pair (include/debig_hip.h: DEBIG_WAVES_SPLIT)
SPLIT_QUEUED = 0x11  # the same behind persistent workgroups and a work queue (DEBIG_WAVES_SPLIT_QUEUED): 3 workgroups, run twice
STRAND = 0x12  # the long-segment scan behind the same LZ77 half (DEBIG_WAVES_STRAND)
STRAND_PIPE = 0x13  # the same two halves on two wavefronts of a workgroup, record by record (DEBIG_WAVES_STRAND_PIPE)
STRAND_PIPE_BIG = 0x113  # (test code only) the same with the 12 KB LZ77 tile the shim uses while the device holds every stream
last_split_retried = 0  # streams the pair handed to the one-kernel path in the last SPLIT call


def layout_batch(raws, caps, in_misalign=0, out_misalign=0, p2=None, flags=0):
    """Pack streams into arenas.  Returns (in_arena, out_arena, streams[], results[])."""
    n = len(raws)
    streams = (DebigStream * n)()
    results = (DebigResult * n)()
    in_off = 64 + in_misalign
    out_off = 64 + out_misalign
    offs = []
    for i, (r, cap) in enumerate(zip(raws, caps)):
        streams[i].in_off = in_off
        streams[i].in_len = len(r)
        streams[i].out_off = out_off
        streams[i].out_cap = cap
        streams[i].flags = flags
        if p2 is not None and p2[i] is not None:
            streams[i].p2_on = 1
            streams[i].p2_s0 = p2[i][0]
            streams[i].p2_est = p2[i][1]
        offs.append((in_off, out_off))
        in_off += (len(r) + 63 + 16) // 16 * 16 + in_misalign
        out_off += (cap + 1024 + 15) // 16 * 16 + out_misalign
    in_arena = np.zeros(in_off + 64, dtype=np.uint8)
    out_arena = np.full(out_off + 64, 0xA5, dtype=np.uint8)
    for (io, _), r in zip(offs, raws):
        in_arena[io:io + len(r)] = np.frombuffer(r, dtype=np.uint8)
        # poison what follows the stream: the kernel must read it as zero
        in_arena[io + len(r):io + len(r) + 8] = 0xFF
    return in_arena, out_arena, streams, results, offs


def emu_inflate(L, raws, caps, grid=0, nw=1, classes=None, ws_bytes=None, chunk_bytes=4096, retry_width=1, **kw):
    """nw = 1: debig_inflate_kernel; nw = 2 / 4: debig_inflate_mw_kernel<nw> (one stream per
    workgroup of nw wavefronts).  classes = [(nw, cls), ...]: one launch per entry, each
    restricted to a stream class (1 small, 2 large), like the shim's mixed-width mode."""
    global last_split_retried
    in_arena, out_arena, streams, results, offs = layout_batch(raws, caps, **kw)
    if classes is not None:
        rc = 0
        for w, cls in classes:
            rc |= L.emu_inflate_batch_cls(in_arena.ctypes.data, out_arena.ctypes.data, streams, results,
                                          len(raws), grid, w, cls)
    elif nw in (SPLIT, SPLIT_QUEUED, STRAND, STRAND_PIPE, STRAND_PIPE_BIG):
        if ws_bytes is None:
            ws_bytes = len(raws) * (32 + 24576) + 9 * sum(len(r) for r in raws)
        nr = C.c_uint32(0)
        if nw == SPLIT_QUEUED:
            os.environ["DEBIG_EMU_SPLIT_QUEUED"] = "1"
        if nw == STRAND:
            os.environ["DEBIG_EMU_STRAND"] = "1"
        if nw in (STRAND_PIPE, STRAND_PIPE_BIG):
            os.environ["DEBIG_EMU_STRAND_PIPE"] = "1"
        if nw == STRAND_PIPE_BIG:
            os.environ["DEBIG_EMU_PIPE_BIG_TILE"] = "1"
        try:
            rc = L.emu_inflate_split_batch(in_arena.ctypes.data, out_arena.ctypes.data, streams, results, len(raws),
                                           ws_bytes, C.byref(nr))
        finally:
            os.environ.pop("DEBIG_EMU_SPLIT_QUEUED", None)
            os.environ.pop("DEBIG_EMU_STRAND", None)
            os.environ.pop("DEBIG_EMU_STRAND_PIPE", None)
            os.environ.pop("DEBIG_EMU_PIPE_BIG_TILE", None)
        last_split_retried = nr.value
    elif nw == CHUNKED:
        if ws_bytes is None:
            tasks = sum(len(r) // chunk_bytes + 1 for r in raws)
            ws_bytes = (len(raws) * 8192 + tasks * (32768 * 4 + 24576) + 12 * sum(len(r) for r in raws) +
                        3 * sum(min(c, 1032 * len(r)) for r, c in zip(raws, caps)))
        nr = C.c_uint32(0)
        rc = L.emu_inflate_chunked_batch(in_arena.ctypes.data, out_arena.ctypes.data, streams, results, len(raws),
                                         ws_bytes, chunk_bytes, retry_width, C.byref(nr))
        last_split_retried = nr.value
    elif nw == 1:
        rc = L.emu_inflate_batch(in_arena.ctypes.data, out_arena.ctypes.data, streams, results, len(raws), grid)
    else:
        rc = L.emu_inflate_mw_batch(in_arena.ctypes.data, out_arena.ctypes.data, streams, results, len(raws),
                                    grid, nw)
    assert rc == 0
    outs = []
    for i, (_, oo) in enumerate(offs):
        r = results[i]
        final = r.final_size if r.final_set else None
        outs.append((r.good, final, out_arena[oo:oo + (final or 0)].tobytes(), r))
    return outs, out_arena, offs
