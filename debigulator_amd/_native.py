"""ctypes view of libdebigulator_hip.so (the C-ABI in include/debig_hip.h).

The product path has NO CPU fallback: if the library (or a GPU) is missing the
calls raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libdebigulator_hip.so")


class DebigStream(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint64), ("out_off", C.c_uint64),
                ("out_cap", C.c_uint64), ("p2_s0", C.c_int64), ("p2_est", C.c_uint64),
                ("p2_on", C.c_uint32), ("flags", C.c_uint32)]


class DebigResult(C.Structure):
    _fields_ = [("final_size", C.c_uint64), ("good", C.c_uint32), ("status", C.c_uint32),
                ("final_set", C.c_uint32), ("n_blocks", C.c_uint32), ("n_windows", C.c_uint32),
                ("n_rounds", C.c_uint32), ("prof", C.c_uint32 * 8),
                ("in_end_bits", C.c_uint64)]


class DebigPngImage(C.Structure):
    _fields_ = [("stream_off", C.c_uint64), ("rgba_off", C.c_uint64), ("pal_off", C.c_uint64),
                ("width", C.c_uint32), ("height", C.c_uint32), ("color_type", C.c_uint32),
                ("asserts_off", C.c_uint32), ("tmp_off", C.c_uint64), ("replay_p3", C.c_uint32),
                ("reserved", C.c_uint32)]


class DebigPngResult(C.Structure):
    _fields_ = [("good", C.c_uint32), ("bad_row", C.c_uint32)]


WAVES_SPLIT = 0x10  # include/debig_hip.h: DEBIG_WAVES_SPLIT
WAVES_SPLIT_QUEUED = 0x11  # DEBIG_WAVES_SPLIT_QUEUED: persistent workgroups + work queue
WAVES_STRAND = 0x12  # DEBIG_WAVES_STRAND: the long-segment scan in front of the same LZ77 half
WAVES_STRAND_PIPE = 0x13  # DEBIG_WAVES_STRAND_PIPE: scan and LZ77 wavefronts side by side in one workgroup
STRAND_PIPE_MAX_STREAMS, STRAND_PIPE_MEAN_IN_BYTES = 2048, 128 << 10  # DEBIG_STRAND_PIPE_MAX_STREAMS / _MEAN_IN_BYTES
STRAND_MIN_STREAMS, STRAND_MAX_STREAMS = 768, 3072  # DEBIG_STRAND_MIN_STREAMS / _MAX_STREAMS: what width 0 picks
WAVES_CHUNKED = 0x20  # include/debig_hip.h: DEBIG_WAVES_CHUNKED
STREAM_IMAGE_ROWS = 2  # DEBIG_STREAM_IMAGE_ROWS: hint for the choice of path (filtered image rows)

_lib = None


def lib():
    """Load the native library; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # In a python process torch must come first: it ships its own libamdhip64/libhsa-runtime64
    # (same soname as /opt/rocm's).  Loaded second, our library binds to torch's copy and both
    # share one HIP runtime (streams, device pointers); loaded first, it would pull in a second
    # runtime and torch.cuda would then see no device.  C callers just link /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("DEBIG_LIB", LIB_PATH)  # diagnostic builds (tools/prof_phases.py)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: run `python -m debigulator_amd.build` (there is no CPU fallback)")
    L = C.CDLL(path)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.debig_hip_inflate_batch.restype = C.c_int
    L.debig_hip_inflate_batch.argtypes = [vp, vp, vp, vp, u32, vp]
    L.debig_hip_inflate_batch_ex.restype = C.c_int
    L.debig_hip_inflate_batch_ex.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    L.debig_hip_inflate_plan_ws.restype = C.c_int
    L.debig_hip_inflate_plan_ws.argtypes = [vp, u32, vp, u64, vp]
    L.debig_hip_inflate_planned_ws.restype = C.c_int
    L.debig_hip_inflate_planned_ws.argtypes = [vp, vp, vp, vp, u32, vp, u64, vp]
    L.debig_hip_mem_free.restype = C.c_uint64
    L.debig_hip_mem_free.argtypes = []
    L.debig_hip_inflate_planned_ws_ex.restype = C.c_int
    L.debig_hip_inflate_planned_ws_ex.argtypes = [vp, vp, vp, vp, u32, u32, vp, u64, vp]
    L.debig_hip_init.restype = C.c_int
    L.debig_hip_init.argtypes = [vp]
    L.debig_hip_inflate_batch_ws.restype = C.c_int
    L.debig_hip_inflate_batch_ws.argtypes = [vp, vp, vp, vp, u32, u32, vp, u64, vp]
    L.debig_hip_inflate_workspace_bytes.restype = u64
    L.debig_hip_inflate_workspace_bytes.argtypes = [u64, u32]
    L.debig_hip_inflate_workspace_bytes_io.restype = u64
    L.debig_hip_inflate_workspace_bytes_io.argtypes = [u64, u64, u32]
    L.debig_hip_inflate_chunked_workspace_bytes.restype = u64
    L.debig_hip_inflate_chunked_workspace_bytes.argtypes = [u64, u64, u32]
    L.debig_hip_png_defilter_batch.restype = C.c_int
    L.debig_hip_png_defilter_batch.argtypes = [vp, vp, vp, vp, u32, vp]
    L.debig_hip_png_decode_fused_batch.restype = C.c_int
    L.debig_hip_png_decode_fused_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, vp, u64, vp]
    L.debig_hip_device_count.restype = C.c_int
    L.debig_hip_set_device.restype = C.c_int
    L.debig_hip_set_device.argtypes = [C.c_int]
    L.debig_hip_malloc.restype = vp
    L.debig_hip_malloc.argtypes = [u64]
    L.debig_hip_free.argtypes = [vp]
    for name in ("debig_hip_memcpy_h2d", "debig_hip_memcpy_d2h"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, vp, u64, vp]
    L.debig_hip_memset.restype = C.c_int
    L.debig_hip_memset.argtypes = [vp, C.c_int, u64, vp]
    L.debig_hip_stream_sync.restype = C.c_int
    L.debig_hip_stream_sync.argtypes = [vp]
    L.debig_hip_error_string.restype = C.c_char_p
    L.debig_hip_error_string.argtypes = [C.c_int]
    L.debig_hip_event_create.restype = vp
    L.debig_hip_event_record.restype = C.c_int
    L.debig_hip_event_record.argtypes = [vp, vp]
    L.debig_hip_event_elapsed_ms.restype = C.c_float
    L.debig_hip_event_elapsed_ms.argtypes = [vp, vp]
    L.debig_hip_event_destroy.argtypes = [vp]
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().debig_hip_error_string(rc)
        raise RuntimeError(f"HIP error {rc} in {what}: {msg.decode() if msg else '?'}")
