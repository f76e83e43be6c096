"""CRC-32 / Adler-32 of byte spans resident in HBM (debig_hip_checksum_batch, include/debig_hip.h).

Mirrors what the reference does with update_crc over PNG chunks (src/decode_png.c:313-333)
and adds the gzip / zlib trailer checks the reference skips."""
import ctypes as C

import numpy as np

from . import _native as N

SPAN_DTYPE = np.dtype([("off", "<u8"), ("len", "<u8")])
CRC32, ADLER32 = 0, 1


def _lib():
    L = N.lib()
    L.debig_hip_checksum_batch.restype = C.c_int
    L.debig_hip_checksum_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    return L


class DeviceChecksums:
    """spans of one device arena (a torch uint8 tensor) -> uint32 results, any number of launches"""

    def __init__(self, d_arena, spans, kind=CRC32):
        import torch

        self.torch = torch
        self.d_arena = d_arena
        self.n = len(spans)
        self.kind = kind
        sp = np.zeros(self.n, dtype=SPAN_DTYPE)
        sp["off"] = [s[0] for s in spans]
        sp["len"] = [s[1] for s in spans]
        self.d_spans = torch.from_numpy(sp.view(np.uint8).reshape(-1)).to(d_arena.device)
        self.d_out = torch.zeros(self.n, dtype=torch.int32, device=d_arena.device)
        self.lib = _lib()
        self.bytes = int(sp["len"].sum())

    def launch(self, stream=None):
        torch = self.torch
        if stream is None:
            stream = torch.cuda.current_stream(self.d_arena.device)
        rc = self.lib.debig_hip_checksum_batch(self.d_arena.data_ptr(), self.d_spans.data_ptr(),
                                               self.d_out.data_ptr(), self.n, self.kind,
                                               C.c_void_p(stream.cuda_stream))
        N.check(rc, "debig_hip_checksum_batch")

    def results(self):
        self.torch.cuda.synchronize()
        return self.d_out.cpu().numpy().view(np.uint32)
