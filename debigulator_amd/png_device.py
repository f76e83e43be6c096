"""PNG decode with everything resident in HBM (BASELINE configs 3 and 4).

Host side only splits the container (pure python here; the C drop-in layer does the same in
csrc/host/debig_png.c): IDAT payloads -> input arena; then ONE inflate launch
(debig_hip_inflate_batch, with the decode_png aliasing-replay parameters) and ONE de-filter
launch (debig_hip_png_defilter_batch).  Used by the GPU tests and tools/bench_png.py.
"""
import ctypes as C
import os

import numpy as np

from . import _native as N
from .batch import DeviceBatch, pack_streams


def split_png(data):
    """-> dict(w, h, ct, raw (DEFLATE payload handed to inflate), palette[768] or None)"""
    d = bytes(data)
    assert d[1:4] == b"PNG"
    at, z, pal, w, h, ct = 8, [], None, 0, 0, 0
    while at + 8 <= len(d):
        ln = int.from_bytes(d[at:at + 4], "big")
        typ = d[at + 4:at + 8]
        body = d[at + 8:at + 8 + ln]
        if typ == b"IHDR":
            w, h, ct = int.from_bytes(body[0:4], "big"), int.from_bytes(body[4:8], "big"), body[9]
        elif typ == b"PLTE":
            p = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
            pal = np.zeros(768, dtype=np.uint8)
            pal[0:len(p)], pal[256:256 + len(p)], pal[512:512 + len(p)] = p[:, 0], p[:, 1], p[:, 2]
        elif typ == b"IDAT":
            z.append(body)
        at += 12 + ln
    zz = b"".join(z)
    return {"w": w, "h": h, "ct": ct, "raw": zz[2:-4], "palette": pal}


# where the fused kernel (one workgroup per image) beats the pair of launches: profiles/r04_fused_vs_pair.txt
FUSED_MIN_IMAGES, FUSED_MAX_IMAGES, FUSED_MAX_STREAM = 513, 1536, 64 << 20
# A batch of mid-size images of two kinds (config 3: photo-like files of about 1 MB of IDAT in some forty DEFLATE blocks among
# small and highly compressible ones) is decoded as TWO batches side by side on two HIP streams: the long streams as chunk
# tasks (DEBIG_WAVES_CHUNKED: a stream is cut at its block headers and spread over many workgroups -- 365 copies of such a
# file: 12 .. 19 ms against 21 on a wavefront pair per stream) + one de-filter launch, the rest through the fused kernel
# (few blocks: nothing to cut; profiles/r04_hybrid_cfg3.txt).
# Streams of a few KB go a third way: one workgroup per stream + a de-filter launch on another side stream (such files tend to
# be a block every few hundred bytes -- 800 blocks in 5 KB --, which the fused kernel's scan hands back: a launch behind it).
HYBRID_LONG_IN_BYTES, HYBRID_MIN_LONG, HYBRID_MIN_IMAGES, HYBRID_MAX_IMAGES = 256 << 10, 1, 8, 4096
HYBRID_TINY_IN_BYTES = 16 << 10
# Few large images whose streams go as chunk tasks (config 4): the de-filter of one group of images beside the inflate of the
# next (launch_pipelined); groups of PIPE_GROUP_IMAGES images (the workspace may allow more per group).
PIPE_MIN_IMAGES, PIPE_GROUP_IMAGES, PIPE_LANES = 4, 8, 1


class DevicePngBatch:
    def __init__(self, pngs, device="cuda:0", strict=False):
        import torch

        self.torch = torch
        self.items = [split_png(p) for p in pngs]
        n = self.n = len(self.items)
        raws = [it["raw"] for it in self.items]
        ests = [4 * it["w"] * it["h"] + it["h"] + 1 for it in self.items]
        p2 = None if strict else [(e - 772 + ((16 - (e & 15)) & 15), e) for e in ests]
        in_arena, streams, out_bytes = pack_streams(raws, ests, p2=p2, flags=N.STREAM_IMAGE_ROWS)
        # palettes live behind the stream arena
        pal_base = self.pal_base = out_bytes  # d_out[:pal_base]: the scanline streams
        self.inflate = DeviceBatch(in_arena, streams, out_bytes + 768 * n + 64, device, plan=True)
        img = (N.DebigPngImage * n)()
        off = 0
        self.rgba_off = []
        for i, it in enumerate(self.items):
            img[i].stream_off = int(streams[i]["out_off"])
            img[i].rgba_off = off
            img[i].pal_off = pal_base + 768 * i
            img[i].width, img[i].height, img[i].color_type, img[i].asserts_off = it["w"], it["h"], it["ct"], 0
            self.rgba_off.append(off)
            off += (4 * it["w"] * it["h"] + 31) // 16 * 16
            if it["ct"] == 3 and it["w"] > 16384:  # include/debig_hip.h: index-row scratch of wide palette images
                img[i].tmp_off = off
                off += (it["w"] + 47) // 16 * 16
            if it["palette"] is not None:
                self.inflate.d_out[pal_base + 768 * i: pal_base + 768 * (i + 1)] = torch.from_numpy(it["palette"]).to(device)
        self.rgba_bytes = sum(4 * it["w"] * it["h"] for it in self.items)
        self.c_bytes = sum(len(r) for r in raws)
        self.s_bytes = sum(e - 1 for e in ests)
        self.max_stream = max(ests) if ests else 0
        self.d_rgba = torch.zeros(off + 64, dtype=torch.uint8, device=device)
        self.d_img = torch.frombuffer(bytearray(bytes(img)), dtype=torch.uint8).to(device)
        self.img_host = np.frombuffer(bytes(img), dtype=np.uint8).reshape(n, C.sizeof(N.DebigPngImage)).copy()
        self.fused = None  # launch_fused: descriptors in its own dispatch order, made on first use
        self.last_fused = False
        self.device = device
        self.pngs, self.strict = pngs, strict
        self.piped, self.last_piped = None, False  # launch_pipelined: image descriptors in dispatch order, side stream
        self.hybrid = None  # launch_hybrid: {"long": (indices, sub-batch), "rest": (indices, sub-batch), side stream}, made on first use
        self.last_hybrid = False
        lens = np.array([len(r) for r in raws], dtype=np.int64)
        self.n_long = int((lens >= HYBRID_LONG_IN_BYTES).sum())
        self.d_ires = torch.zeros(n * C.sizeof(N.DebigPngResult), dtype=torch.uint8, device=device)
        self.lib = N.lib()

    def launch(self, stream=None, waves_per_stream=0, fused=None, hybrid=None):
        """waves_per_stream: inflate width (include/debig_hip.h: debig_hip_inflate_batch_ex), 0 = the batch's own plan.
        fused: True = one kernel per batch (launch_fused), False = inflate launch + de-filter launch, None = the faster of the
        two for this batch by the measured rule (FUSED_MIN_IMAGES .. FUSED_MAX_IMAGES images, none above FUSED_MAX_STREAM
        bytes of scanline stream), unless a width is asked for"""
        auto = fused is None and waves_per_stream == 0 and not os.environ.get("DEBIG_WAVES_PER_STREAM")
        if hybrid is None:  # (a batch of long streams only is one part: chunk tasks + a de-filter launch)
            hybrid = (auto and HYBRID_MIN_IMAGES <= self.n <= HYBRID_MAX_IMAGES and self.max_stream <= FUSED_MAX_STREAM and
                      HYBRID_MIN_LONG <= self.n_long)
        if hybrid:
            return self.launch_hybrid(stream)
        if fused is None:
            fused = auto and FUSED_MIN_IMAGES <= self.n <= FUSED_MAX_IMAGES and self.max_stream <= FUSED_MAX_STREAM
        if fused:
            return self.launch_fused(stream)
        self.last_hybrid = False
        torch = self.torch
        if stream is None:
            stream = torch.cuda.current_stream(self.inflate.device)
        self.last_fused = self.last_piped = False
        if (auto and self.inflate.planned_waves == N.WAVES_CHUNKED and self.n >= PIPE_MIN_IMAGES and
                os.environ.get("DEBIG_PNG_PIPELINE", "0") == "1"):
            return self.launch_pipelined(stream)
        self.inflate.launch(stream, waves_per_stream=waves_per_stream)
        rc = self.lib.debig_hip_png_defilter_batch(self.inflate.d_out.data_ptr(), self.d_rgba.data_ptr(),
                                                   self.d_img.data_ptr(), self.d_ires.data_ptr(), self.n,
                                                   C.c_void_p(stream.cuda_stream))
        N.check(rc, "debig_hip_png_defilter_batch")

    def launch_pipelined(self, stream=None, group_images=None, lanes=None):
        """Few large images (config 4: chunk tasks): the inflate goes through in groups of streams anyway (one workspace,
        reused); the de-filter of a group's images runs on a side stream while the next group inflates -- it fills the slots
        the inflate's serial and narrow launches (window walk, chain, repair, carve) leave idle.  Same results as launch()."""
        torch = self.torch
        inf = self.inflate
        dev = inf.device
        if stream is None:
            stream = torch.cuda.current_stream(dev)
        if group_images is None:
            group_images = int(os.environ.get("DEBIG_PNG_PIPE_GROUP", "0")) or PIPE_GROUP_IMAGES
        if lanes is None:
            lanes = int(os.environ.get("DEBIG_PNG_PIPE_LANES", "0")) or PIPE_LANES
        if self.piped is None:
            order = inf.order if inf.order is not None else np.arange(self.n)
            self.piped = {"order": np.asarray(order),
                          "d_img": torch.from_numpy(np.ascontiguousarray(self.img_host[order]).reshape(-1)).to(dev),
                          "d_ires": torch.zeros_like(self.d_ires),
                          "side": torch.cuda.Stream(device=dev, priority=-1), "lanes": []}
        p = self.piped
        side = p["side"]
        while len(p["lanes"]) < lanes:
            p["lanes"].append(torch.cuda.Stream(device=dev))
        isz, rsz = C.sizeof(N.DebigPngImage), C.sizeof(N.DebigPngResult)
        ev0 = torch.cuda.Event()
        ev0.record(stream)
        for ln in p["lanes"][:lanes]:
            ln.wait_event(ev0)

        def after_group(first, count, on):
            # the de-filter launches all go to ONE side stream (they share the library's progress counters; a group's
            # de-filter is short beside its inflate), behind the event of their group's inflate
            ev = torch.cuda.Event()
            ev.record(on)
            side.wait_event(ev)
            rc = self.lib.debig_hip_png_defilter_batch(inf.d_out.data_ptr(), self.d_rgba.data_ptr(),
                                                       p["d_img"].data_ptr() + first * isz, p["d_ires"].data_ptr() + first * rsz,
                                                       count, C.c_void_p(side.cuda_stream))
            N.check(rc, "debig_hip_png_defilter_batch")

        inf._launch_chunked(stream, after_group=after_group, max_group=group_images, lanes=p["lanes"][:lanes])
        for ln in p["lanes"][:lanes]:
            stream.wait_stream(ln)
        stream.wait_stream(side)
        self.last_piped, self.last_fused, self.last_hybrid = True, False, False

    def launch_hybrid(self, stream=None):
        """the long streams as chunk tasks + a de-filter launch on a side stream, the tiny ones as a workgroup per stream + a
        de-filter launch on another, the others through the fused kernel on `stream`; `stream` continues when all are through.
        Same results as launch(fused=False)."""
        torch = self.torch
        dev = self.inflate.device
        if stream is None:
            stream = torch.cuda.current_stream(dev)
        if self.hybrid is None:
            lens = np.array([len(it["raw"]) for it in self.items], dtype=np.int64)
            cls = {"long": np.nonzero(lens >= HYBRID_LONG_IN_BYTES)[0],
                   "tiny": np.nonzero(lens < HYBRID_TINY_IN_BYTES)[0],
                   "rest": np.nonzero((lens < HYBRID_LONG_IN_BYTES) & (lens >= HYBRID_TINY_IN_BYTES))[0]}
            self.hybrid = {"parts": []}
            for key in ("long", "rest", "tiny"):
                idx = cls[key]
                if len(idx):
                    sub = DevicePngBatch([self.pngs[i] for i in idx], device=self.device, strict=self.strict)
                    # every part on a stream of its own; the fused kernel's on a HIGH-priority one: its few, latency-bound
                    # workgroups (50 KB of LDS each) must get their slots while the chunk tasks of the long streams fill the chip
                    self.hybrid["parts"].append((key, idx, sub, torch.cuda.Stream(device=dev, priority=-1 if key == "rest" else 0)))
        ev0 = torch.cuda.Event()
        ev0.record(stream)
        order = sorted(self.hybrid["parts"], key=lambda p: {"rest": 0, "long": 1, "tiny": 2}[p[0]])
        for key, idx, sub, side in order:
            side.wait_event(ev0)
            if key == "long":
                sub.launch(side, waves_per_stream=N.WAVES_CHUNKED, fused=False, hybrid=False)
            elif key == "rest" and sub.n >= 16:
                sub.launch_fused(side)
            else:
                sub.launch(side, fused=False, hybrid=False)
        for key, idx, sub, side in order:
            stream.wait_stream(side)
        self.last_hybrid, self.last_fused, self.last_piped = True, False, False

    def launch_fused(self, stream=None, workspace_bytes=None):
        """inflate and de-filter in ONE kernel (debig_hip_png_decode_fused_batch): a workgroup per
        image, the de-filter wavefronts a tile behind the LZ77 wavefront.  Same results as launch()."""
        torch = self.torch
        inf = self.inflate
        if stream is None:
            stream = torch.cuda.current_stream(inf.device)
        if self.fused is None:
            st = inf.streams_host
            # the longest streams first (the batch ends when its slowest image does); stream i and image i stay together
            order = np.argsort(-(st["in_len"].astype(np.int64) + st["out_cap"].astype(np.int64)), kind="stable")
            f = {"order": order}
            f["d_streams"] = torch.from_numpy(np.ascontiguousarray(st[order]).view(np.uint8).reshape(-1)).to(inf.device)
            f["d_img"] = torch.from_numpy(np.ascontiguousarray(self.img_host[order]).reshape(-1)).to(inf.device)
            f["d_res"] = torch.zeros_like(inf.d_results)
            f["d_ires"] = torch.zeros_like(self.d_ires)
            total_in, total_out = int(st["in_len"].sum()), int(st["out_cap"].sum())
            nbytes = int(self.lib.debig_hip_inflate_workspace_bytes_io(total_in, total_out, self.n))
            f["ws_bytes"] = nbytes
            f["d_ws"] = torch.empty(nbytes, dtype=torch.uint8, device=inf.device)
            self.fused = f
        f = self.fused
        wsb = f["ws_bytes"] if workspace_bytes is None else min(int(workspace_bytes), f["ws_bytes"])
        rc = self.lib.debig_hip_png_decode_fused_batch(inf.d_in.data_ptr(), inf.d_out.data_ptr(), f["d_streams"].data_ptr(),
                                                       f["d_res"].data_ptr(), self.d_rgba.data_ptr(), f["d_img"].data_ptr(),
                                                       f["d_ires"].data_ptr(), self.n, f["d_ws"].data_ptr(), wsb,
                                                       C.c_void_p(stream.cuda_stream))
        N.check(rc, "debig_hip_png_decode_fused_batch")
        self.last_fused, self.last_hybrid, self.last_piped = True, False, False

    def launch_inflate_only(self, stream=None):
        self.last_fused = self.last_hybrid = self.last_piped = False  # (the object's own arenas: what results() / rgba() read from now on)
        self.inflate.launch(stream)

    def launch_defilter_only(self, stream=None):
        """De-filter the streams a previous launch_inflate_only() / launch(fused=False) left in HBM (timing of that kernel alone)."""
        self.last_fused = self.last_hybrid = self.last_piped = False
        torch = self.torch
        if stream is None:
            stream = torch.cuda.current_stream(self.inflate.device)
        rc = self.lib.debig_hip_png_defilter_batch(self.inflate.d_out.data_ptr(), self.d_rgba.data_ptr(),
                                                   self.d_img.data_ptr(), self.d_ires.data_ptr(), self.n,
                                                   C.c_void_p(stream.cuda_stream))
        N.check(rc, "debig_hip_png_defilter_batch")

    def results(self):
        self.torch.cuda.synchronize()
        if self.last_hybrid:  # the two halves back in the caller's order
            from .batch import RESULT_DTYPE
            res = np.empty(self.n, dtype=RESULT_DTYPE)
            ires = np.empty(self.n, dtype=np.dtype([("good", "<u4"), ("bad_row", "<u4")]))
            for key, idx, sub, side in self.hybrid["parts"]:
                r, ir = sub.results()
                res[idx], ires[idx] = r, ir
            return res, ires
        if self.last_fused:  # back to the caller's order
            from .batch import RESULT_DTYPE
            f = self.fused
            res = np.empty(self.n, dtype=RESULT_DTYPE)
            res[f["order"]] = f["d_res"].cpu().numpy().view(RESULT_DTYPE)
            ires = np.empty(self.n, dtype=np.dtype([("good", "<u4"), ("bad_row", "<u4")]))
            ires[f["order"]] = f["d_ires"].cpu().numpy().view(ires.dtype)
            return res, ires
        if self.last_piped:  # back to the caller's order
            ires = np.empty(self.n, dtype=np.dtype([("good", "<u4"), ("bad_row", "<u4")]))
            ires[self.piped["order"]] = self.piped["d_ires"].cpu().numpy().view(ires.dtype)
            return self.inflate.results(), ires
        ires = self.d_ires.cpu().numpy().view(np.dtype([("good", "<u4"), ("bad_row", "<u4")]))
        return self.inflate.results(), ires

    def rgba(self, i):
        if self.last_hybrid:
            for key, idx, sub, side in self.hybrid["parts"]:
                k = np.nonzero(idx == i)[0]
                if len(k):
                    return sub.rgba(int(k[0]))
        it = self.items[i]
        o = self.rgba_off[i]
        return self.d_rgba[o:o + 4 * it["w"] * it["h"]].cpu().numpy()
