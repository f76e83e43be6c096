"""inflate -> de-filter in ONE kernel (csrc/png_fused_kernel.inc, debig_hip_png_decode_fused_batch).

CPU part: the kernel on the lock-step emulator against (a) the unfused pair of kernels on the same emulator, field by
field and byte by byte, and (b) the reference-made digests of tests/golden.  GPU part (-m gpu): the reference's sample
files and synthetic images through the C-ABI against the same digests / the oracle."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import emu_binding as eb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
PNG_ROW_REDO = 0xFFFFFFFD


class DebigPngImage(C.Structure):
    _fields_ = [("stream_off", C.c_uint64), ("rgba_off", C.c_uint64), ("pal_off", C.c_uint64),
                ("width", C.c_uint32), ("height", C.c_uint32), ("color_type", C.c_uint32),
                ("asserts_off", C.c_uint32), ("tmp_off", C.c_uint64), ("replay_p3", C.c_uint32), ("reserved", C.c_uint32)]


class DebigPngResult(C.Structure):
    _fields_ = [("good", C.c_uint32), ("bad_row", C.c_uint32)]


@pytest.fixture(scope="module")
def emu():
    L = eb.load_emu(asan=bool(os.environ.get("DEBIG_EMU_ASAN")))
    L.emu_png_fused_batch.restype = C.c_int
    L.emu_png_fused_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32)]
    L.emu_png_defilter_batch_w.restype = C.c_int
    L.emu_png_defilter_batch_w.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    return L


def _split(png):
    """the container walk of csrc/host/debig_png.c, in python for the test"""
    at, z, pal, w, h, ct = 8, b"", None, 0, 0, 0
    while at + 8 <= len(png):
        ln = int.from_bytes(png[at:at + 4], "big")
        typ, body = png[at + 4:at + 8], png[at + 8:at + 8 + ln]
        if typ == b"IHDR":
            w, h, ct = int.from_bytes(body[0:4], "big"), int.from_bytes(body[4:8], "big"), body[9]
        elif typ == b"PLTE":
            p = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
            pal = np.zeros(768, dtype=np.uint8)
            pal[0:len(p)], pal[256:256 + len(p)], pal[512:512 + len(p)] = p[:, 0], p[:, 1], p[:, 2]
        elif typ == b"IDAT":
            z += body
        at += 12 + ln
    return {"w": w, "h": h, "ct": ct, "raw": z[2:-4], "palette": pal}


def _layout(items, p2=True):
    """arenas for n images: compressed bytes | scanline streams (+ palettes) | pixels"""
    raws = [it["raw"] for it in items]
    ests = [4 * it["w"] * it["h"] + it["h"] + 1 for it in items]
    p2s = [(e - 772 + ((16 - (e & 15)) & 15), e) for e in ests] if p2 else None
    in_arena, s_arena, streams, results, offs = eb.layout_batch(raws, ests, in_misalign=3, out_misalign=5, p2=p2s)
    n = len(items)
    pal_base = len(s_arena)
    s_arena = np.concatenate([s_arena, np.zeros(768 * n + 64, dtype=np.uint8)])
    img = (DebigPngImage * n)()
    off = 0
    rgba_off = []
    for i, it in enumerate(items):
        img[i].stream_off = streams[i].out_off
        img[i].rgba_off = off
        img[i].pal_off = pal_base + 768 * i
        img[i].width, img[i].height, img[i].color_type, img[i].asserts_off = it["w"], it["h"], it["ct"], 0
        rgba_off.append(off)
        off += (4 * it["w"] * it["h"] + 31) // 16 * 16
        if it["palette"] is not None:
            s_arena[pal_base + 768 * i: pal_base + 768 * (i + 1)] = it["palette"]
    rgba = np.zeros(off + 64, dtype=np.uint8)
    return in_arena, s_arena, streams, results, img, rgba, rgba_off


def _run_both(emu, items, ws_bytes=None, p2=True):
    n = len(items)
    if ws_bytes is None:
        ws_bytes = n * (32 + 24576) + 12 * sum(len(it["raw"]) for it in items)
    # fused
    in_a, s_a, streams, res_f, img, rgba_f, rgba_off = _layout(items, p2)
    pres_f = (DebigPngResult * n)()
    nr = C.c_uint32(0)
    assert emu.emu_png_fused_batch(in_a.ctypes.data, s_a.ctypes.data, streams, res_f, rgba_f.ctypes.data, img, pres_f, n, ws_bytes,
                                   C.byref(nr)) == 0
    # the pair of kernels
    in_b, s_b, streams_b, res_u, img_b, rgba_u, _ = _layout(items, p2)
    pres_u = (DebigPngResult * n)()
    os.environ["DEBIG_EMU_STRAND_PIPE"] = "1"
    try:
        nr2 = C.c_uint32(0)
        assert emu.emu_inflate_split_batch(in_b.ctypes.data, s_b.ctypes.data, streams_b, res_u, n, ws_bytes, C.byref(nr2)) == 0
    finally:
        os.environ.pop("DEBIG_EMU_STRAND_PIPE", None)
    assert emu.emu_png_defilter_batch_w(s_b.ctypes.data, rgba_u.ctypes.data, img_b, pres_u, n, 8) == 0
    assert nr.value == nr2.value
    for i in range(n):
        a, b = res_f[i], res_u[i]
        assert (a.good, a.status, a.final_size, a.final_set, a.n_blocks, a.in_end_bits) == \
               (b.good, b.status, b.final_size, b.final_set, b.n_blocks, b.in_end_bits), i
        assert (pres_f[i].good, pres_f[i].bad_row) == (pres_u[i].good, pres_u[i].bad_row), i
    assert np.array_equal(s_a, s_b)  # the scanline streams, the poison between them included
    assert np.array_equal(rgba_f, rgba_u)
    out = []
    for i, it in enumerate(items):
        out.append((res_f[i].good, pres_f[i].good, rgba_f[rgba_off[i]:rgba_off[i] + 4 * it["w"] * it["h"]]))
    return out, nr.value


def test_fused_kernel_on_emulator_synthetic_images(emu):
    """tests/golden/png_synth.json: every filter type, a palette image, a stored stream the reference's gate refuses, 1 x 1"""
    synth = json.load(open(os.path.join(GOLD, "png_synth.json")))
    items = [_split(bytes.fromhex(p["png_hex"])) for p in synth]
    out, _ = _run_both(emu, items)
    for p, (ig, pg, rgba) in zip(synth, out):
        assert int(ig and pg) == p["good"], p["seed"]
        if p["good"]:
            assert hashlib.sha256(rgba.tobytes()).hexdigest() == p["rgba_sha256"], p["seed"]


def test_fused_kernel_on_emulator_reference_sample_files(emu):
    """the small sample files of the reference, digests made by the reference itself (tests/golden/resources.json)"""
    gold = json.load(open(os.path.join(GOLD, "resources.json")))["png"]
    names = ["structuredart1.png", "immunetomustsurvive.png", "structuredart2.png", "extraturns.png", "font.png", "structuredart3.png"]
    items = [_split(open(os.path.join(GOLD, "resources", f), "rb").read()) for f in names]
    out, _ = _run_both(emu, items)
    for f, (ig, pg, rgba) in zip(names, out):
        assert ig == 1 and pg == 1, f
        assert hashlib.sha256(rgba.tobytes()).hexdigest() == gold[f]["rgba_sha256"], f


def test_fused_kernel_on_emulator_aliasing_replay(emu):
    """phoebus.png: the decode_png aliasing replay (P2) rewrites the tail of the scanline stream when the inflate ends --
    the de-filter wavefronts must not have taken those rows earlier"""
    gold = json.load(open(os.path.join(GOLD, "resources.json")))["png"]["phoebus.png"]
    items = [_split(open(os.path.join(GOLD, "resources", "phoebus.png"), "rb").read())]
    out, _ = _run_both(emu, items)
    ig, pg, rgba = out[0]
    assert ig == 1 and pg == 1
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == gold["rgba_sha256"]


def test_fused_kernel_on_emulator_streams_handed_back(emu):
    """a workspace too small for some streams: the scan hands them back (DEBIG_E_RETRY), the image is given up as REDO,
    the launches that follow decode it -- same pixels"""
    gold = json.load(open(os.path.join(GOLD, "resources.json")))["png"]
    names = ["immunetomustsurvive.png", "structuredart2.png", "font.png"]
    items = [_split(open(os.path.join(GOLD, "resources", f), "rb").read()) for f in names]
    out, retried = _run_both(emu, items, ws_bytes=3 * 1024 + 3 * 96 + 4096 + 40000)
    assert retried >= 1
    for f, (ig, pg, rgba) in zip(names, out):
        assert ig == 1 and pg == 1, f
        assert hashlib.sha256(rgba.tobytes()).hexdigest() == gold[f]["rgba_sha256"], f


# ---------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_gpu_fused_reference_sample_files(gpu_device):
    """14 sample files of the reference (all but the colour type 2 one that needs the caller's prior buffer), several copies
    each, through debig_hip_png_decode_fused_batch: the reference's own digests"""
    import glob

    from debigulator_amd.png_device import DevicePngBatch

    gold = json.load(open(os.path.join(GOLD, "resources.json")))["png"]
    files = [f for f in sorted(glob.glob(os.path.join(GOLD, "resources", "*.png"))) if not f.endswith("backgrounddetailed1.png")]
    datas = [open(f, "rb").read() for f in files]
    n = 3 * len(files)
    b = DevicePngBatch([datas[i % len(datas)] for i in range(n)], device=gpu_device)
    b.launch_fused()
    res, ires = b.results()
    assert (res["good"] == 1).all() and (ires["good"] == 1).all()
    for i in range(n):
        name = os.path.basename(files[i % len(files)])
        assert hashlib.sha256(b.rgba(i).tobytes()).hexdigest() == gold[name]["rgba_sha256"], (i, name)
    # and again over the same buffers (stale progress words, results of the run before)
    b.launch_fused()
    res, ires = b.results()
    assert (res["good"] == 1).all() and (ires["good"] == 1).all()
    for i in (0, 7, n - 1):
        name = os.path.basename(files[i % len(files)])
        assert hashlib.sha256(b.rgba(i).tobytes()).hexdigest() == gold[name]["rgba_sha256"], (i, name)


@pytest.mark.gpu
def test_gpu_fused_synthetic_images_vs_oracle(oracle, gpu_device):
    """every filter type / palette / RGB, 512 x 512 and odd sizes, against the oracle's decode_png"""
    from debigulator_amd import workload
    from debigulator_amd.png_device import DevicePngBatch

    pngs = []
    for k, (w, h, ct, ft, enc) in enumerate([(512, 512, 6, 4, "dynamic"), (512, 512, 6, 5, "dynamic"), (333, 211, 6, 3, "fixed"),
                                              (640, 480, 3, 5, "dynamic"), (257, 129, 2, 5, "dynamic"), (1, 1, 6, 0, "dynamic"),
                                              (4096, 70, 6, 2, "dynamic"), (64, 2000, 6, 1, "dynamic")]):
        pngs.append(workload.make_png(500 + k, w, h, ct=ct, ftype=ft, enc=enc)[0])
    b = DevicePngBatch(pngs * 3, device=gpu_device)
    b.launch_fused()
    res, ires = b.results()
    b2 = DevicePngBatch(pngs * 3, device=gpu_device)
    b2.launch()
    res2, ires2 = b2.results()
    assert (res["good"] == res2["good"]).all() and (ires["good"] == ires2["good"]).all()
    assert (res["final_size"] == res2["final_size"]).all() and (res["status"] == res2["status"]).all()
    for i, png in enumerate(pngs * 3):
        assert np.array_equal(b.rgba(i), b2.rgba(i)), i
        if b.items[i]["ct"] == 2:
            continue  # (the oracle replays the reference's RGB quirk P3, which needs the caller's prior buffer: test_gpu_dropin)
        ok, px = oracle.decode_png(png)
        assert bool(res["good"][i] and ires["good"][i]) == bool(ok), i
        if ok:
            assert np.array_equal(b.rgba(i), px), i


@pytest.mark.gpu
def test_gpu_fused_streams_handed_back(gpu_device):
    """a workspace that cannot hold every stream's tokens: those streams come back as DEBIG_E_RETRY, their images as REDO,
    and the same call decodes both"""
    import glob

    from debigulator_amd.png_device import DevicePngBatch

    gold = json.load(open(os.path.join(GOLD, "resources.json")))["png"]
    files = [f for f in sorted(glob.glob(os.path.join(GOLD, "resources", "*.png"))) if not f.endswith("backgrounddetailed1.png")]
    datas = [open(f, "rb").read() for f in files]
    b = DevicePngBatch(datas * 2, device=gpu_device)
    for ws in (6 << 20, 4096):  # 12 x the input would be ~ 150 MB; 4 KB: no usable workspace at all (a workgroup per stream)
        b.d_rgba.zero_()
        b.launch_fused(workspace_bytes=ws)
        res, ires = b.results()
        assert (res["good"] == 1).all() and (ires["good"] == 1).all(), ws
        for i in range(len(datas) * 2):
            name = os.path.basename(files[i % len(files)])
            assert hashlib.sha256(b.rgba(i).tobytes()).hexdigest() == gold[name]["rgba_sha256"], (ws, i, name)


@pytest.mark.gpu
def test_gpu_hybrid_launch_reference_sample_files(gpu_device):
    """what DevicePngBatch.launch() picks for a config-3-like mix: the long streams as chunk tasks on a side stream, the rest
    through the fused kernel -- the reference's digests for every image, twice over the same buffers"""
    import glob

    from debigulator_amd.png_device import DevicePngBatch

    gold = json.load(open(os.path.join(GOLD, "resources.json")))["png"]
    files = [f for f in sorted(glob.glob(os.path.join(GOLD, "resources", "*.png"))) if not f.endswith("backgrounddetailed1.png")]
    datas = [open(f, "rb").read() for f in files]
    n = 5 * len(files)
    b = DevicePngBatch([datas[i % len(datas)] for i in range(n)], device=gpu_device)
    for rep in range(2):
        b.launch()
        assert b.last_hybrid
        res, ires = b.results()
        assert (res["good"] == 1).all() and (ires["good"] == 1).all()
        for i in range(n) if rep == 0 else (0, 3, 17, n - 1):
            name = os.path.basename(files[i % len(files)])
            assert hashlib.sha256(b.rgba(i).tobytes()).hexdigest() == gold[name]["rgba_sha256"], (rep, i, name)
    # and the same batch through the two launches
    b.launch(fused=False)
    assert not b.last_hybrid
    res2, ires2 = b.results()
    assert (res2["good"] == 1).all() and (res2["final_size"] == res["final_size"]).all()
