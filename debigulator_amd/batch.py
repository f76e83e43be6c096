"""Batched inflate on the GPU: arenas in HBM + one kernel launch.

torch is used for what it is good at here -- device memory and streams; the
compute is the hand-written HIP kernel behind `debig_hip_inflate_batch`
(include/debig_hip.h).  Mirrors N calls of the reference's inflate()
(src/inflate.h:51-60): per stream an input slice, a recipient slice with
`recipient_size`, and back come `final_recipient_size` and `good`.
"""
import ctypes as C
import os

import numpy as np

from . import _native as N

STREAM_DTYPE = np.dtype([("in_off", "<u8"), ("in_len", "<u8"), ("out_off", "<u8"), ("out_cap", "<u8"),
                         ("p2_s0", "<i8"), ("p2_est", "<u8"), ("p2_on", "<u4"), ("flags", "<u4")])
RESULT_DTYPE = np.dtype([("final_size", "<u8"), ("good", "<u4"), ("status", "<u4"), ("final_set", "<u4"),
                         ("n_blocks", "<u4"), ("n_windows", "<u4"), ("n_rounds", "<u4"), ("prof", "<u4", (8,)),
                         ("in_end_bits", "<u8")])
assert STREAM_DTYPE.itemsize == C.sizeof(N.DebigStream)
assert RESULT_DTYPE.itemsize == C.sizeof(N.DebigResult)

OUT_SLACK = 64  # bytes between recipients (never written; lets tests verify that)


def _align(x, a):
    return (x + a - 1) // a * a


def pack_streams(raws, caps, in_align=16, out_align=16, in_skew=0, out_skew=0, p2=None, flags=0):
    """Lay n compressed streams and their recipients out in two arenas (host side).

    Returns (in_arena uint8[], streams structured[], out_bytes)."""
    n = len(raws)
    streams = np.zeros(n, dtype=STREAM_DTYPE)
    in_off = 0
    out_off = 0
    for i in range(n):
        in_off = _align(in_off, in_align) + in_skew
        out_off = _align(out_off, out_align) + out_skew
        streams[i]["in_off"] = in_off
        streams[i]["in_len"] = len(raws[i])
        streams[i]["out_off"] = out_off
        streams[i]["out_cap"] = caps[i]
        streams[i]["flags"] = flags
        if p2 is not None and p2[i] is not None:
            streams[i]["p2_on"] = 1
            streams[i]["p2_s0"] = p2[i][0]
            streams[i]["p2_est"] = p2[i][1]
        in_off += len(raws[i])
        out_off += caps[i] + OUT_SLACK
    in_arena = np.zeros(_align(in_off, 16) + 64, dtype=np.uint8)
    for i in range(n):
        o = int(streams[i]["in_off"])
        in_arena[o:o + len(raws[i])] = np.frombuffer(raws[i], dtype=np.uint8)
    return in_arena, streams, _align(out_off, 16) + 64


LARGE_IN_BYTES = 256 << 10  # include/debig_hip.h: DEBIG_LARGE_IN_BYTES
LARGE_OUT_BYTES = 1 << 20   # DEBIG_LARGE_OUT_BYTES
WAVES_LARGE4_SMALL1 = 0x41  # DEBIG_WAVES_LARGE4_SMALL1
CHUNKED_ROWS_MIN_IN_BYTES = 256 << 10  # csrc/host/debig_ctx.h: DEBIG_CHUNKED_ROWS_MIN_IN_BYTES


def pick_waves(streams):
    """csrc/host/debig_ctx.h: debig_pick_waves, line for line.  0 = the library's own choice from the
    batch size (8 / 4 / 2 wavefronts per stream up to 256 / 512 / 768 streams, the long-segment scan up to 3072,
    the scan + LZ77 pair beyond), which is what the C rule returns in those cases."""
    n = len(streams)
    lens = streams["in_len"].astype(np.int64)
    if n <= 1024:
        if n and int(lens.sum()) >= n << 20:  # few streams, >= 1 MiB of input each on average
            return N.WAVES_CHUNKED
        if n and n <= 512 and (streams["flags"] & N.STREAM_IMAGE_ROWS).all() and int(lens.min()) >= CHUNKED_ROWS_MIN_IN_BYTES:
            return N.WAVES_CHUNKED  # filtered image rows, every stream long: chunk tasks from 256 KiB on
        if 256 < n <= N.STRAND_MIN_STREAMS and int(lens.sum()) >= n * N.STRAND_PIPE_MEAN_IN_BYTES:
            return N.WAVES_STRAND_PIPE  # long streams: scan and LZ77 half side by side
        return 0  # (8 / 4 / 2 wavefronts per stream up to 256 / 512 / 768 streams, the pipeline up to 1024)
    n_large = int(((lens >= LARGE_IN_BYTES) | (streams["out_cap"].astype(np.int64) >= LARGE_OUT_BYTES)).sum())
    if int(lens.max()) >= 4 << 20 and n <= 16384:  # thousands of streams, a very large one among them
        return N.WAVES_CHUNKED
    return WAVES_LARGE4_SMALL1 if 0 < n_large <= 256 else 0


def plan_batch(streams):
    """Dispatch plan of one launch, the same rule as csrc/host/debig_ctx.h: debig_plan_batch.
    Chunk tasks when pick_waves says so (never reordered); workgroups start in descriptor order, so a
    batch of 513..1024 streams whose largest quarter holds at least half of the input bytes is
    launched 4 wavefronts wide, the streams that touch the most bytes (in_len + out_cap) first.  -> (order or None, waves_per_stream or 0)."""
    n = len(streams)
    waves = pick_waves(streams)
    if waves == N.WAVES_CHUNKED or n <= 512 or n > 1024:
        return None, waves
    lens = streams["in_len"].astype(np.int64)
    order = np.lexsort((np.arange(n), -lens))
    total, top = int(lens.sum()), int(lens[order[: n // 4]].sum())
    if total and top * 2 >= total:
        # the order itself goes by the bytes a stream touches, in_len + out_cap: a small input that decodes
        # to megabytes runs as long as a large one (config 3: 40.3 -> 34.1 ms)
        work = lens + streams["out_cap"].astype(np.int64)
        return np.lexsort((np.arange(n), -work)), 4
    return None, waves


class DeviceBatch:
    """Streams resident in HBM, ready to be inflated any number of times."""

    def __init__(self, in_arena, streams, out_bytes, device="cuda:0", plan=False):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.n = len(streams)
        self.streams_host = streams
        self.d_in = torch.from_numpy(in_arena).to(self.device)
        self.d_out = torch.zeros(out_bytes, dtype=torch.uint8, device=self.device)
        # streams_host stays in the caller's order; the device copy may be in dispatch order
        self.order, self.planned_waves = plan_batch(streams) if plan else (None, 0)
        dev_streams = streams if self.order is None else np.ascontiguousarray(streams[self.order])
        self.d_streams = torch.from_numpy(dev_streams.view(np.uint8).reshape(-1)).to(self.device)
        self.d_results = torch.zeros(self.n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
        self.lib = N.lib()
        self.d_ws = None  # token workspace of the scan / LZ77 kernel pair, allocated on first use
        self.dev_streams_host = dev_streams
        self.chunk_groups = None  # WAVES_CHUNKED: [(first, count)] stream groups that fit the workspace

    @classmethod
    def from_streams(cls, raws, caps, device="cuda:0", plan=False, **kw):
        in_arena, streams, out_bytes = pack_streams(raws, caps, **kw)
        return cls(in_arena, streams, out_bytes, device, plan=plan)

    def launch(self, stream=None, waves_per_stream=0):
        """Asynchronous: one kernel launch on `stream` (default: torch's current stream).
        waves_per_stream: 1 (one wavefront per stream), 2 / 4 / 8 (one stream per workgroup of
        that many wavefronts, for few large streams), _native.WAVES_SPLIT (scan + LZ77 kernel pair,
        the throughput path; WAVES_SPLIT_QUEUED: behind persistent workgroups and a work queue), 0 = library's choice from the batch size."""
        torch = self.torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        if waves_per_stream == 0 and self.planned_waves and not os.environ.get("DEBIG_WAVES_PER_STREAM"):
            waves_per_stream = self.planned_waves
        if waves_per_stream == 0 and int(os.environ.get("DEBIG_WAVES_PER_STREAM", "0"), 0) == N.WAVES_CHUNKED:
            waves_per_stream = N.WAVES_CHUNKED
        if waves_per_stream == N.WAVES_CHUNKED:
            return self._launch_chunked(stream)
        ws_ptr, ws_bytes = None, 0
        if waves_per_stream == 0 and self.n > N.STRAND_MIN_STREAMS and not os.environ.get("DEBIG_WAVES_PER_STREAM"):
            # include/debig_hip.h: what 0 means for this many streams
            waves_per_stream = (N.WAVES_STRAND_PIPE if self.n <= N.STRAND_PIPE_MAX_STREAMS else
                                N.WAVES_STRAND if self.n <= N.STRAND_MAX_STREAMS else N.WAVES_SPLIT)
        if waves_per_stream in (N.WAVES_SPLIT, N.WAVES_SPLIT_QUEUED, N.WAVES_STRAND, N.WAVES_STRAND_PIPE):
            if self.d_ws is None:  # caller-owned workspace: nothing is allocated inside the call
                total_in = int(self.streams_host["in_len"].sum())
                total_out = int(self.streams_host["out_cap"].sum())
                nbytes = int(self.lib.debig_hip_inflate_workspace_bytes_io(total_in, total_out, self.n))
                nbytes = int(nbytes * float(os.environ.get("DEBIG_WS_SCALE", "1")))  # experiments: a larger / smaller token workspace
                self.d_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            ws_ptr, ws_bytes = self.d_ws.data_ptr(), self.d_ws.numel()
            if self.n <= 16384:
                # the descriptors and the workspace are this object's own and never change: the workspace
                # is carved once (debig_hip_inflate_plan_ws), every launch is scan + LZ77 only
                # (the plan kernel runs on ONE stream: a launch on another stream plans again there -- cheaper
                # than an event chain, and the slots are only ever carved to the same values)
                if not getattr(self, "_planned", False) or getattr(self, "_plan_stream", None) != stream.cuda_stream:
                    rc = self.lib.debig_hip_inflate_plan_ws(self.d_streams.data_ptr(), self.n, ws_ptr, ws_bytes,
                                                            C.c_void_p(stream.cuda_stream))
                    self._planned = rc == 0
                    self._plan_stream = stream.cuda_stream
                if self._planned:
                    N.check(self.lib.debig_hip_inflate_planned_ws_ex(self.d_in.data_ptr(), self.d_out.data_ptr(),
                                                                     self.d_streams.data_ptr(), self.d_results.data_ptr(), self.n,
                                                                     waves_per_stream,
                                                                     ws_ptr, ws_bytes, C.c_void_p(stream.cuda_stream)),
                            "debig_hip_inflate_planned_ws_ex")
                    return
                # a workspace too small to plan: debig_hip_inflate_batch_ws below falls back to the one-kernel path
        rc = self.lib.debig_hip_inflate_batch_ws(self.d_in.data_ptr(), self.d_out.data_ptr(),
                                                 self.d_streams.data_ptr(), self.d_results.data_ptr(),
                                                 self.n, waves_per_stream, ws_ptr, ws_bytes,
                                                 C.c_void_p(stream.cuda_stream))
        N.check(rc, "debig_hip_inflate_batch_ws")

    def _launch_chunked(self, stream, after_group=None, max_group=0, lanes=None):
        """The chunk-parallel path for few large streams: the batch goes through in groups of
        streams whose workspace need (debig_hip_inflate_chunked_workspace_bytes) fits
        the free device memory less an eighth (DEBIG_CHUNKED_WS_MB overrides); one workspace, reused group after group.
        after_group(first, count): called when a group's launches are enqueued (dispatch-order positions: what a caller
        chains behind that group's streams, DevicePngBatch's de-filter); max_group: at most that many streams in a group;
        lanes: HIP streams that take the groups in turn, each with a workspace of its own (group k's launches go to
        lanes[k % len(lanes)] and run beside the other lanes' groups; after_group gets the lane as third argument; the
        caller orders `stream` and the lanes)."""
        torch = self.torch
        if self.chunk_groups is not None and getattr(self, "_chunk_max_group", 0) != max_group:
            self.chunk_groups = None
        if self.chunk_groups is None:
            self._chunk_max_group = max_group
            # few, large groups (every group pays the window kernel's serial walk and a dozen launch tails:
            # profiles/r03_chunk_workspace_groups.txt): a group may take what the device has free, less an
            # eighth (at least 40 GiB asked for); DEBIG_CHUNKED_WS_MB overrides.  Same rule as csrc/host/debig_ctx.c
            if os.environ.get("DEBIG_CHUNKED_WS_MB"):
                cap = int(os.environ["DEBIG_CHUNKED_WS_MB"]) << 20
            else:
                free = int(torch.cuda.mem_get_info(self.device)[0])
                cap = max(40960 << 20, free - free // 8)
            need = lambda a, b, c: int(self.lib.debig_hip_inflate_chunked_workspace_bytes(int(a), int(b), int(c)))
            ds = self.dev_streams_host

            def carve(per_group):
                groups, first, tin, tout, biggest = [], 0, 0, 0, 0
                for i in range(self.n):
                    a, b = int(ds[i]["in_len"]), int(ds[i]["out_cap"])
                    if i > first and (need(tin + a, tout + b, i - first + 1) > cap or i - first >= per_group):
                        groups.append((first, i - first))
                        biggest = max(biggest, need(tin, tout, i - first))
                        first, tin, tout = i, 0, 0
                    tin += a
                    tout += b
                groups.append((first, self.n - first))
                return groups, max(biggest, need(tin, tout, self.n - first))

            groups, biggest = carve(max_group if max_group else self.n)
            groups, biggest = carve(-(-self.n // len(groups)))  # evened out: as many streams in each as the fullest needs
            self.chunk_groups = groups
            self.d_ws_lanes = []
            self.d_ws_chunked = None
            try:
                self.d_ws_chunked = torch.empty(biggest, dtype=torch.uint8, device=self.device)
            except RuntimeError:  # out of device memory: the library falls back to whole workgroups per stream
                self.d_ws_chunked = None
            self.d_ws_lanes = [self.d_ws_chunked]
        nl = min(len(lanes), len(self.chunk_groups)) if lanes else 1
        while len(self.d_ws_lanes) < nl and self.d_ws_chunked is not None:
            try:
                self.d_ws_lanes.append(torch.empty_like(self.d_ws_chunked))
            except RuntimeError:  # no room for another workspace: fewer lanes
                break
        nl = min(nl, len(self.d_ws_lanes))
        ssz, rsz = STREAM_DTYPE.itemsize, RESULT_DTYPE.itemsize
        for k, (first, count) in enumerate(self.chunk_groups):
            ws = self.d_ws_lanes[k % nl]
            on = lanes[k % nl] if lanes else stream
            rc = self.lib.debig_hip_inflate_batch_ws(self.d_in.data_ptr(), self.d_out.data_ptr(),
                                                     self.d_streams.data_ptr() + first * ssz,
                                                     self.d_results.data_ptr() + first * rsz, count, N.WAVES_CHUNKED,
                                                     ws.data_ptr() if ws is not None else None,
                                                     ws.numel() if ws is not None else 0,
                                                     C.c_void_p(on.cuda_stream))
            N.check(rc, "debig_hip_inflate_batch_ws")
            if after_group is not None:
                after_group(first, count, on)

    def results(self):
        """results in the caller's stream order"""
        self.torch.cuda.synchronize(self.device)
        raw = self.d_results.cpu().numpy().view(RESULT_DTYPE)
        if self.order is None:
            return raw
        out = np.empty_like(raw)
        out[self.order] = raw
        return out

    def output(self, i, res=None):
        res = self.results() if res is None else res
        o = int(self.streams_host[i]["out_off"])
        n = int(res[i]["final_size"])
        return self.d_out[o:o + n].cpu().numpy().tobytes()

    def outputs_host(self):
        self.torch.cuda.synchronize(self.device)
        return self.d_out.cpu().numpy()
