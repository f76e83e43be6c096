#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: decompressed GB/s + fraction of the HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg4|cfg5]

--config cfg2 (default; BASELINE config 2 = configs[1], SURVEY.md 8d "cfg2"): per GPU 4096
  fixed-Huffman streams + 4096 stored-block streams, every stream inflating to one 64 KiB
  block, generated deterministically (tools/streamgen.c, seed 0xDEB16 + global stream index).
  Weak scaling: per-GPU work is fixed as N grows.
--config cfg5 (BASELINE config 5 = configs[4]): --members (default 65536) gzip members of
  1 MiB (text-like payload, dynamic Huffman, ratio about 2.5:1), member i -> GPU i mod N.
  Strong scaling: the total is fixed.  The payload bytes are inflated by the same kernel; the
  CRC-32 trailer of every member is verified on the GPU outside the timed region.

--config cfg4 (BASELINE config 4 = configs[3]): per GPU --images (default 32: 256 images on 8 GPUs)
  PNG files of 8192 x 8192 RGBA, every row Paeth-filtered, IDAT ratio about 3:1, 4 distinct seeds
  cycled; a step is inflate + de-filter (decode_png's hot path) and `value` is GB/s of RGBA.  Weak.

One "step" = one pass of the hot path (the batched inflate through the C-ABI,
include/debig_hip.h) over this rank's whole shard, inputs already resident in HBM.

  value         decompressed GB/s, whole job = sum over ranks of D bytes / max-over-ranks time
  roofline      the whole batch: algorithmic bytes C + D per step / average step duration measured
                with events on the launch stream.  cfg2 also carries roofline_huffman and
                roofline_stored: each stream kind launched alone, timed the same way.
  cpu_baseline  the compiled reference (oracle/_ref; or the oracle port if that prebuilt library
                is absent) on the host cores: one thread per kind, blended, and all cores.

Default run (`--config cfg2`, no --no-cfg5): the line also carries `cfg5_strong`, BASELINE config 5
measured by the same command on the same ranks -- 65536 gzip members of 1 MiB, member i -> GPU
i mod N, strong scaling (value, ms_per_step, roofline on C + D, max-over-ranks time) -- so the
SCALE runs (`--gpus 1/2/4/8`, no other flag) report both the weak-scaling headline and config 5.
`value` of the line itself stays the cfg2 headline at every N.  At N = 1 the line also carries `cfg3_png`,
BASELINE config 3 (1024 PNGs cycled from the reference's sample files, decode_png -> RGBA, resident in HBM): what
DevicePngBatch.launch() picks, the pair of launches and the fused kernel, digests against the reference's.

`--dist`: initialise torch.distributed even at N = 1 (backend nccl = RCCL at world size 1: loads
librccl, runs init_process_group and the shard-map broadcast on a cuda tensor).

Multi-GPU: `--gpus N` with N > 1 started WITHOUT a torch.distributed environment spawns
`python -m torch.distributed.run --nproc-per-node N` on itself (before anything touches the
GPU) and passes rank 0's JSON line through; started by torch.distributed.run it is a rank.
Rank 0 builds the shard map (stream id -> rank) and broadcasts it (RCCL); every rank inflates
its own shard; no payload collective.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable
STREAMS_PER_KIND = 4096
STREAM_BYTES = 65536
CFG5_MEMBERS = 65536
CFG5_MEMBER_BYTES = 1 << 20
CFG5_DISTINCT = 512  # distinct member seeds in the whole job; member i has seed i mod 512


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg4", "cfg5"])
    ap.add_argument("--images", type=int, default=32, help="cfg4: PNG images per rank")
    ap.add_argument("--side", type=int, default=8192, help="cfg4: image width and height")
    ap.add_argument("--streams", type=int, default=STREAMS_PER_KIND, help="cfg2: streams per kind per rank")
    ap.add_argument("--members", type=int, default=CFG5_MEMBERS, help="cfg5: gzip members in the whole job")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=64, help="streams per kind checked byte for byte")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the N>1 code path on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--dist", action="store_true", help="initialise torch.distributed even at N = 1")
    ap.add_argument("--no-cfg5", action="store_true", help="cfg2: skip the cfg5_strong record")
    ap.add_argument("--no-cfg3", action="store_true", help="cfg2: skip the cfg3_png record (rank 0 at N = 1 only)")
    ap.add_argument("--no-kinds", action="store_true", help="cfg2: skip roofline_huffman / roofline_stored / ms_with_plan / roofline_interleaved "
                    "(profiling runs: the kernel statistics then hold whole-batch launches only)")
    ap.add_argument("--cfg5-steps", type=int, default=3, help="timed steps of the cfg5_strong record")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 100 if args.config == "cfg2" else 5
    if args.warmup is None:
        args.warmup = 5 if args.config == "cfg2" else 1
    return args


def spawn_ranks(args):
    """Parent of an N-rank run: never imports torch, never touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def host_description():
    model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"cpu_model": model, "nproc": os.cpu_count() or 1, "usable_cores": usable}


# ---------------------------------------------------------------- CPU baseline (checker side)
def _cpu_engine():
    from oracle import binding

    if binding.ref_available("A"):
        try:
            return binding.Reference("A"), "reference"
        except OSError:
            pass
    return binding.Oracle(), "port"


def _cpu_pass(eng, sample):
    n = 0
    for raw, cap, want in sample:
        out = eng.inflate(raw, cap)
        assert out[0] == 1 and out[1] == want
        n += want
    return n


def _cpu_timed(eng, sample, budget_s, max_passes=64):
    nbytes, passes, t0 = 0, 0, time.perf_counter()
    while True:
        nbytes += _cpu_pass(eng, sample)
        passes += 1
        if time.perf_counter() - t0 > budget_s or passes >= max_passes:
            break
    dt = time.perf_counter() - t0
    return nbytes / dt / 1e9, passes, dt


def _cpu_worker(job):
    sample, budget_s = job
    eng, _ = _cpu_engine()
    t0 = time.perf_counter()
    nbytes = 0
    while time.perf_counter() - t0 < budget_s:
        nbytes += _cpu_pass(eng, sample)
    return nbytes, time.perf_counter() - t0


def cpu_baseline(samples, label, do_all_cores=True):
    """samples: {kind: [(raw, recipient_size, decompressed size)]}.  Reference (or port) on the host
    cores: one thread per kind, one thread blended, and every usable core (one process each)."""
    import multiprocessing as mp

    eng, kind = _cpu_engine()
    host = host_description()
    per_kind = {}
    for k, smp in samples.items():
        v, passes, dt = _cpu_timed(eng, smp, 4.0)
        per_kind[k] = {"value": v, "passes": passes, "seconds": round(dt, 2)}
    blend = [x for smp in samples.values() for x in smp]
    v1, passes, dt1 = _cpu_timed(eng, blend, 5.0)
    # the GPU box gives one GPU's job a share of 16 host cores (more worker processes than that
    # only measure process start-up); nproc of the whole machine is reported beside it
    cores = max(1, min(host["usable_cores"], 16))
    all_core = None
    try:
        if not do_all_cores:
            raise RuntimeError("not run (this process may hold a GPU context)")
        ctx = mp.get_context("spawn")  # never fork a process that holds a GPU context
        with ctx.Pool(cores) as pool:
            t0 = time.perf_counter()
            outs = pool.map(_cpu_worker, [(blend, 5.0)] * cores)
            wall = time.perf_counter() - t0
        all_core = {"value": sum(o[0] for o in outs) / max(o[1] for o in outs) / 1e9, "cores": cores,
                    "seconds": round(wall, 2)}
    except Exception as e:  # a report, never a reason to lose the GPU number
        all_core = {"value": None, "cores": cores, "error": str(e)[:200]}
    return {
        "value": v1,
        "unit": "GB/s decompressed",
        "cores": 1,
        "kind": kind,
        "sample": f"{label}; whole passes, 1 thread, {dt1:.1f} s ({passes} passes)",
        "per_kind_1thread": per_kind,
        "all_cores": all_core,
        "host": host,
    }


def run_cpu_baseline(args, rank, samples, label, all_cores=True):
    """rank 0 only, and BEFORE this process initialises the GPU or torch.distributed (cfg2 / cfg5: the
    samples are global units 0.., the same at every N): the all-core leg starts worker processes,
    which a process holding a GPU context must not do on this pool.  all_cores = False: no child
    processes (cfg4 at N > 1, where the sample is this rank's first image)."""
    if args.no_cpu_baseline or rank != 0:
        return None
    try:
        return cpu_baseline(samples, label, all_cores)
    except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
        return {"value": None, "unit": "GB/s decompressed", "cores": 1, "kind": "port", "sample": f"failed: {e}"}


def early_samples(args):
    """cfg2 / cfg5 baseline samples, built without the shard map: global units 0.. of the workload"""
    from debigulator_amd import workload

    if args.config == "cfg5":
        n = 8
        pairs = [workload.make_stream("dynamic", g % CFG5_DISTINCT, CFG5_MEMBER_BYTES) for g in range(n)]
        raws = [workload.gzip_member(r, p)[10:-8] for r, p in pairs]
        return ({"gzip_dynamic": [(r, max(CFG5_MEMBER_BYTES + 1, len(r)), CFG5_MEMBER_BYTES) for r in raws]},
                f"gzip members 0..{n - 1} of the timed job (1 MiB each, dynamic Huffman)")
    out = {}
    for kind in ("fixed", "stored"):
        ps = [workload.make_stream(kind, g, STREAM_BYTES) for g in range(128)]
        out[kind] = [(p[0], max(STREAM_BYTES + 1, len(p[0])), STREAM_BYTES) for p in ps]
    return out, "fixed-Huffman and stored streams 0..127 (64 KiB each) of the timed workload"


# ---------------------------------------------------------------- GPU side
def time_launches(torch, fn, steps):
    """average duration of fn() over `steps` launches, events on torch's current stream (the
    stream the kernels are launched on)"""
    e = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    e[0].record()
    for k in range(steps):
        fn()
        e[k + 1].record()
    torch.cuda.synchronize()
    return sum(e[k].elapsed_time(e[k + 1]) for k in range(steps)) / steps


def pmc_traffic(kernel_sources_digest, scale):
    """HBM bytes per launch from the PMC passes of this same command (tools/pmc_traffic.sh +
    tools/pmc_summary.py -> profiles/pmc_traffic.json).  Only reported when that file was made
    from the kernel sources that are being timed now; otherwise null."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tj.get("kernel_sources_sha256") != kernel_sources_digest:
            return None, "profiles/pmc_traffic.json is from other kernel sources: not reported"
        return tj["bytes_per_launch"] * scale, tj.get("source", "profiles/pmc_traffic.json")
    except (OSError, ValueError, KeyError):
        return None, None


def cfg3_record(torch, np, dev):
    """BASELINE config 3 through debigulator_amd.png_device.DevicePngBatch: the pair of launches, the fused kernel
    and what launch() picks (the three-way split), each the median of 5 launches between events on the launch stream."""
    import glob
    import hashlib

    from debigulator_amd.png_device import DevicePngBatch

    rdir = os.path.join(ROOT, "tests", "golden", "resources")
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "resources.json")))["png"]
    files = [f for f in sorted(glob.glob(os.path.join(rdir, "*.png"))) if not f.endswith("backgrounddetailed1.png")]
    datas = [open(f, "rb").read() for f in files]
    b = DevicePngBatch([datas[i % len(datas)] for i in range(1024)], device=dev)

    def med(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    ms_pair = med(lambda: b.launch(fused=False, hybrid=False))
    ms_fused = med(b.launch_fused)
    ms_default = med(b.launch)
    how = "three batches side by side (long streams as chunk tasks, tiny ones a workgroup each, the rest through the fused kernel)" \
        if b.last_hybrid else ("the fused kernel" if b.last_fused else "two launches")
    res, ires = b.results()
    ok = bool((res["good"] == 1).all() and (ires["good"] == 1).all())
    for i, f in enumerate(files):  # the first copy of every file against the reference-made digest
        ok = ok and hashlib.sha256(b.rgba(i).tobytes()).hexdigest() == gold[os.path.basename(f)]["rgba_sha256"]
    assert ok, "cfg3: an image differs from the reference's digest"
    P, Cb, Sb = b.rgba_bytes, b.c_bytes, b.s_bytes
    return {
        "workload": "cfg3: 1024 PNGs cycled from 14 of the reference's sample files (tests/golden/resources), decode_png -> RGBA, "
                    "everything resident in HBM",
        "value": P / ms_default / 1e6, "unit": "GB/s of RGBA", "ms_per_step": ms_default,
        "launch": how,
        "ms_two_launches": ms_pair, "ms_fused_kernel": ms_fused,
        "compressed_bytes": Cb, "scanline_bytes": Sb, "rgba_bytes": P,
        "roofline": {"bound": "hbm", "achieved": (Cb + P) / ms_default / 1e6, "peak": 8000.0, "unit": "GB/s",
                     "frac": (Cb + P) / ms_default / 1e6 / 8000.0, "traffic": None, "algorithmic_bytes": Cb + P},
        "bit_exact_checked": "every image's good flags; sha256 of the RGBA of one copy of each file against tests/golden/resources.json "
                             "(made by the compiled reference)",
    }


def kernel_sources_digest():
    import glob
    import hashlib

    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "debigulator_amd", "csrc", "*.inc")) +
                    glob.glob(os.path.join(ROOT, "debigulator_amd", "csrc", "*.hip"))):
        h.update(open(p, "rb").read())
    return h.hexdigest()


def build_cfg5(args, torch, np, mine, world, dev, ncpu):
    """BASELINE config 5 on this rank: member i -> GPU i mod N.  Distinct payloads: seed = i mod 512;
    this rank's members cycle through its 512/N seeds, replicated ON THE DEVICE into one arena per
    direction (every member has its own input and output bytes in HBM).  Returns the batch, its
    verify() and the byte counts."""
    from concurrent.futures import ThreadPoolExecutor

    from debigulator_amd import _native, workload
    from debigulator_amd.batch import RESULT_DTYPE, DeviceBatch, pack_streams

    assert args.members % world == 0
    n_mine = len(mine)
    uniq = max(1, min(n_mine, CFG5_DISTINCT // world))
    with ThreadPoolExecutor(ncpu) as ex:
        pairs = list(ex.map(lambda g: workload.make_stream("dynamic", int(g) % CFG5_DISTINCT, CFG5_MEMBER_BYTES),
                            mine[:uniq]))
    members = [workload.gzip_member(r, p) for r, p in pairs]
    raws_u = [m[10:-8] for m in members]  # decode_gz's host side: 10-byte header, 8-byte trailer
    caps_u = [max(CFG5_MEMBER_BYTES + 1, len(r)) for r in raws_u]
    in_block, st_block, out_block = pack_streams(raws_u, caps_u)
    reps = (n_mine + uniq - 1) // uniq
    streams = np.tile(st_block, reps)[:n_mine].copy()
    k = np.arange(n_mine) // uniq
    streams["in_off"] += (k * len(in_block)).astype(np.uint64)
    streams["out_off"] += (k * out_block).astype(np.uint64)
    batch = DeviceBatch.__new__(DeviceBatch)
    batch.torch, batch.device, batch.n, batch.streams_host = torch, torch.device(dev), n_mine, streams
    blk = torch.from_numpy(in_block).to(dev)
    batch.d_in = blk.repeat(reps)
    batch.d_out = torch.zeros(out_block * reps, dtype=torch.uint8, device=dev)
    batch.order, batch.planned_waves, batch.d_ws = None, 0, None
    batch.d_streams = torch.from_numpy(streams.view(np.uint8).reshape(-1)).to(dev)
    batch.d_results = torch.zeros(n_mine * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    batch.lib = _native.lib()

    def verify():
        import struct

        from debigulator_amd.checksum import CRC32, DeviceChecksums

        res = batch.results()
        assert (res["good"] == 1).all(), "a member failed"
        assert (res["final_size"] == CFG5_MEMBER_BYTES).all(), "wrong size"
        spans = [(int(streams[i]["out_off"]), CFG5_MEMBER_BYTES) for i in range(n_mine)]
        ck = DeviceChecksums(batch.d_out, spans, CRC32)
        ck.launch()
        want = np.array([struct.unpack("<I", members[i % uniq][-8:-4])[0] for i in range(n_mine)], dtype=np.uint32)
        assert (np.asarray(ck.results(), dtype=np.uint32) == want).all(), "a member's CRC-32 differs"
        for i in sorted({0, n_mine // 2, n_mine - 1}):
            assert batch.output(i, res) == pairs[i % uniq][1].tobytes(), f"member {i} differs"
        return res

    name = (f"cfg5: {args.members} gzip members of 1 MiB (text-like, dynamic Huffman), member i -> GPU "
            f"i mod {world}; seed 0xDEB16 + (i mod {CFG5_DISTINCT}), each member has its own bytes in HBM")
    return batch, verify, int(streams["in_len"].sum()), CFG5_MEMBER_BYTES * n_mine, name


def timed_steps(torch, np, dist, coll_dev, batch, verify, steps, warmup, c_bytes, d_bytes):
    """W untimed launches, the bit-exactness gate, then exactly K launches between barrier + sync on
    both sides.  Returns (max-over-ranks seconds, this rank's mean step in ms from events on the
    launch stream, job D bytes, job C bytes, results)."""
    for _ in range(warmup):
        batch.launch()
    torch.cuda.synchronize()
    res = verify()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(steps):
        batch.launch()
        ev[k + 1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    tot = torch.tensor([float(d_bytes), float(c_bytes)], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    step_ms = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]))
    return float(t.item()), step_ms, float(tot[0].item()), float(tot[1].item()), res


def roof(c, d, ms, what, launches):
    ach = (c + d) / (ms * 1e-3) / 1e9
    return {"kernel": what, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_step": c + d,
            "avg_step_ms": ms, "decompressed_GBps": d / ms / 1e6, "launches_per_step": launches}


SPLIT_WHAT = ("one step = debig_scanlz_kernel (the scan and LZ77 halves fused: one wavefront per stream) + "
              "debig_inflate_kernel for streams handed back (none here), over a workspace carved once by "
              "debig_split_plan_kernel (debig_hip_inflate_plan_ws / _planned_ws_ex; more than 16384 streams: plan + "
              "scanlz + hand-back per group of 16384), whole batch on rank 0")


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    cfg5 = args.config == "cfg5"
    cfg4 = args.config == "cfg4"

    # ---- CPU baseline first (rank 0, any N): nothing has touched the GPU or torch.distributed yet
    cpu_line = None
    cpu5_line = None  # the reference on config 5's data (dynamic Huffman, 1 MiB members): beside cfg5_strong
    if not cfg4 and rank == 0 and not args.no_cpu_baseline:
        samples, sample_label = early_samples(args)
        cpu_line = run_cpu_baseline(args, rank, samples, sample_label)
        if not cfg5 and not args.no_cfg5:
            try:
                a5 = argparse.Namespace(**vars(args))
                a5.config = "cfg5"
                s5, l5 = early_samples(a5)
                eng, kind = _cpu_engine()
                v, passes, dt = _cpu_timed(eng, s5["gzip_dynamic"], 5.0)
                cpu5_line = {"value": v, "unit": "GB/s decompressed", "cores": 1, "kind": kind,
                             "sample": f"{l5}; whole passes, 1 thread, {dt:.1f} s ({passes} passes)"}
            except Exception as e:  # a report, never a reason to lose the GPU number
                cpu5_line = {"value": None, "unit": "GB/s decompressed", "cores": 1, "kind": "port", "sample": f"failed: {e}"}

    import numpy as np
    import torch

    dist = None
    if world > 1 or args.dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # --dist at N = 1, started without a launcher
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            s.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.one_device:
            local_rank = 0
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)
    dev = f"cuda:{local_rank}"
    coll_dev = dev if args.backend == "nccl" else "cpu"  # where collective tensors live
    if dist is not None:
        torch.cuda.set_device(local_rank)

    from concurrent.futures import ThreadPoolExecutor

    from debigulator_amd import shard, workload
    from debigulator_amd.batch import DeviceBatch

    ncpu = max(1, min(16, (os.cpu_count() or 8) // max(1, min(world, 8))))

    # ---- shard map: rank 0 builds it, RCCL broadcasts it (the only collective on the path)
    def my_units(n_units):
        if dist is not None:
            smap = shard.broadcast_shard_map(n_units, coll_dev, dist)
            return shard.my_streams(smap, rank)  # global ids of this rank's units, in local order
        return np.arange(n_units, dtype=np.int64)  # one rank owns everything

    mine = my_units(args.members if cfg5 else world * (args.images if cfg4 else args.streams))

    kinds = {}
    if cfg4:
        # ---- cfg4: image g has seed 9000 + (g mod 4); this rank's images resident in HBM as one batch
        from debigulator_amd.png_device import DevicePngBatch, split_png

        seeds = sorted({int(g) % 4 for g in mine})
        with ThreadPoolExecutor(min(4, ncpu)) as ex:
            made = dict(zip(seeds, ex.map(lambda sd: workload.make_png(9000 + sd, args.side, args.side, ct=6, ftype=4,
                                                                       noise=workload.CFG4_NOISE, enc="dynamic",
                                                                       idat_chunk=65536), seeds)))
        pngs = [made[int(g) % 4][0] for g in mine]
        it0 = split_png(pngs[0])
        est0 = 4 * args.side * args.side + args.side + 1
        samples = {"png_idat_stream": [(it0["raw"], est0, est0 - 1)]}
        sample_label = (f"inflate() of ONE image's IDAT stream ({len(it0['raw']) / 1e6:.0f} MB -> {(est0 - 1) / 1e6:.0f} MB of "
                        f"filtered rows); the reference's de-filter loops are not in this number")
        cpu_line = run_cpu_baseline(args, rank, samples, sample_label, all_cores=dist is None)
        torch.cuda.set_device(local_rank)
        pbatch = DevicePngBatch(pngs, device=dev)
        batch = pbatch  # .launch() = inflate + de-filter
        c_bytes = pbatch.c_bytes
        d_bytes = pbatch.rgba_bytes
        unit_bytes = 4 * args.side * args.side

        def verify():
            res, ires = pbatch.results()
            assert (res["good"] == 1).all() and (ires["good"] == 1).all(), "an image failed"
            for i in sorted({0, 1 % len(pngs), 2 % len(pngs), 3 % len(pngs), len(pngs) - 1}):
                want = np.asarray(made[int(mine[i]) % 4][1]).reshape(-1)
                assert np.array_equal(pbatch.rgba(i), want), f"image {i} differs from the generator's pixels"
            return res

        workload_name = (f"cfg4: per GPU {len(pngs)} PNG images {args.side}x{args.side} RGBA, all rows Paeth, IDAT ratio "
                         f"about 3:1, seeds 9000 + (i mod 4); one step = inflate + de-filter of the whole batch")
    elif not cfg5:
        # ---- cfg2: this rank's shard, synthesised deterministically, parked in HBM as ONE batch:
        # the long-running Huffman streams first, the stored ones behind them
        assert len(mine) == args.streams
        per = args.streams
        pairs_keep = {}
        for kind in ("fixed", "stored"):
            with ThreadPoolExecutor(ncpu) as ex:
                pairs_keep[kind] = list(ex.map(lambda g: workload.make_stream(kind, int(g), STREAM_BYTES), mine))
        all_pairs = pairs_keep["fixed"] + pairs_keep["stored"]
        raws = [p[0] for p in all_pairs]
        caps = [max(STREAM_BYTES + 1, len(r)) for r in raws]
        torch.cuda.set_device(local_rank)
        batch = DeviceBatch.from_streams(raws, caps, device=dev)
        c_kind = {k: sum(len(p[0]) for p in v) for k, v in pairs_keep.items()}
        c_bytes = sum(c_kind.values())
        d_bytes = STREAM_BYTES * len(raws)
        unit_bytes = STREAM_BYTES

        def verify():
            res = batch.results()
            assert (res["good"] == 1).all(), "a stream failed"
            assert (res["final_size"] == STREAM_BYTES).all(), "wrong size"
            for base in (0, per):
                for i in range(base, base + min(args.verify, per)):
                    assert batch.output(i, res) == all_pairs[i][1].tobytes(), f"stream {i} differs"
            return res

        # each kind alone (its own descriptors over the same arenas would change nothing: own batch)
        for kind, lo in (() if args.no_kinds else (("fixed", 0), ("stored", per))):
            kinds[kind] = (DeviceBatch.from_streams(raws[lo:lo + per], caps[lo:lo + per], device=dev),
                           c_kind[kind], per * STREAM_BYTES)
        workload_name = (f"cfg2: per GPU {per} fixed-Huffman + {per} stored DEFLATE streams, 64 KiB each, one "
                         f"batch per step (one 64 KiB block per stream; stored = 65535+1 byte blocks), "
                         f"seed 0xDEB16+i")
    else:
        torch.cuda.set_device(local_rank)
        batch, verify, c_bytes, d_bytes, workload_name = build_cfg5(args, torch, np, mine, world, dev, ncpu)
        unit_bytes = CFG5_MEMBER_BYTES

    # ---- timed region: exactly K steps between barrier+sync on both sides
    dt_max, step_ms, job_d, job_c, res = timed_steps(torch, np, dist, coll_dev, batch, verify, args.steps, args.warmup,
                                                     c_bytes, d_bytes)

    line = None
    if rank == 0:
        value = job_d * args.steps / dt_max / 1e9
        digest = kernel_sources_digest()
        split = len(res) > 768  # include/debig_hip.h: what the library picks from the batch size (a workspace path)
        strand = split and len(res) <= 3072  # ... DEBIG_WAVES_STRAND up to 3072 streams, DEBIG_WAVES_SPLIT beyond
        if cfg4:
            # inflate reads C and writes the filtered rows S; the de-filter reads S and writes the pixels P
            s_bytes = pbatch.s_bytes
            what = ("one step = inflate of the IDAT streams (chunk-parallel path, inflate_chunk_kernel.inc: 14 small "
                    "launches per stream group) + debig_png_defilter_kernel; algorithmic bytes C + 2 S + P")
            rl = roof(c_bytes + 2 * s_bytes, d_bytes, step_ms, what, None)
            rl["decompressed_GBps"] = d_bytes / step_ms / 1e6
        else:
            rl = roof(c_bytes, d_bytes, step_ms, (SPLIT_WHAT.replace("debig_scanlz_kernel", "debig_strand_kernel") if strand
                                                  else SPLIT_WHAT) if split else
                      "one step = one debig_inflate(_mw)_kernel launch, whole batch on rank 0",
                      (2 if len(res) <= 16384 else 3 * ((len(res) + 16383) // 16384)) if split else 1)
        if not cfg5 and not cfg4:
            tr, src = pmc_traffic(digest, args.streams / STREAMS_PER_KIND)
            rl["traffic"] = tr
            if src:
                rl["traffic_source"] = src
        line = {
            "metric": ("decoded RGBA GB/s (whole node) + % HBM roofline, bit-exact vs reference" if cfg4 else
                       "decompressed GB/s (whole node) + % HBM roofline, bit-exact vs reference"),
            "value": value,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if cfg5 else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": workload_name,
                "units_per_gpu": len(res),
                "unit_decompressed_bytes": unit_bytes,
                "decompressed_bytes_job": job_d,
                "compressed_bytes_job": job_c,
                "ratio": job_d / job_c,
                "bit_exact_checked": ("every image's good flags; images 0-3 and the last byte for byte against the "
                                      "generator's pixels" if cfg4 else
                                      "every member's size/good flag + CRC-32 (on the GPU), 3 members byte for byte"
                                      if cfg5 else f"{min(args.verify, args.streams)} streams per kind byte for byte "
                                      f"+ all sizes/good flags"),
            },
            "roofline": rl,
        }
        if dist is not None:
            line["config"]["sharding"] = (f"round-robin by unit id over {world} ranks, shard map broadcast "
                                          f"({args.backend}), no payload collective")
        nw = res["n_windows"].astype(np.float64).sum()
        if nw > 0:
            line["config"]["avg_spec_rounds_per_window"] = float(res["n_rounds"].astype(np.float64).sum() / nw)
        # each stream kind alone (untimed extras, rank 0)
        for kind, (sub, cb, db) in kinds.items():
            for _ in range(2):
                sub.launch()
            ms = time_launches(torch, sub.launch, 10)
            r = sub.results()
            assert (r["good"] == 1).all()
            name = "roofline_huffman" if kind == "fixed" else "roofline_stored"
            line[name] = roof(cb, db, ms, f"{kind} streams launched alone ({len(r)} x 64 KiB), same kernels",
                              2 if len(r) > 768 else 1)
        if not cfg4 and not cfg5 and split and len(res) <= 16384 and not args.no_kinds:
            # (a) the timed steps run over a workspace carved ONCE (DeviceBatch: debig_hip_inflate_plan_ws before the
            # warm-up); a caller of debig_hip_inflate_batch(_ws) with fresh descriptors pays debig_split_plan_kernel
            # on every call: the same step with the plan kernel in it
            def launch_with_plan():
                batch._planned = False
                batch.launch()

            for _ in range(2):
                launch_with_plan()
            ms_plan = time_launches(torch, launch_with_plan, 20)
            verify()
            rl["plan_in_timed_region"] = False
            rl["ms_with_plan"] = ms_plan
            rl["frac_with_plan"] = (c_bytes + d_bytes) / (ms_plan * 1e-3) / 1e9 / HBM_PEAK_GBS
            # (b) the headline parks all Huffman streams in front of all stored ones (the measured best order: one
            # workgroup per stream is dealt to the shader engines by index).  The same streams ALTERNATING kinds
            # through the same default dispatch: what a caller who does not sort by kind gets
            order = [i // 2 + (per if i % 2 else 0) for i in range(2 * per)]
            ib = DeviceBatch.from_streams([raws[i] for i in order], [caps[i] for i in order], device=dev)
            for _ in range(2):
                ib.launch()
            ms_il = time_launches(torch, ib.launch, 20)
            ri = ib.results()
            assert (ri["good"] == 1).all() and (ri["final_size"] == STREAM_BYTES).all()
            for k in (0, 1, 2 * per - 2, 2 * per - 1):
                assert ib.output(k, ri) == all_pairs[order[k]][1].tobytes(), f"interleaved stream {k} differs"
            line["roofline_interleaved"] = roof(c_bytes, d_bytes, ms_il,
                                                "the same streams, kinds alternating (fixed, stored, fixed, ...), default "
                                                "dispatch (DEBIG_WAVES_SPLIT): order sensitivity of one workgroup per stream", 2)
            del ib
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line

    # ---- BASELINE config 5 beside the cfg2 headline (every rank takes part: it has its own barriers)
    if not cfg4 and not cfg5 and not args.no_cfg5:
        del batch
        kinds.clear()
        torch.cuda.empty_cache()
        mine5 = my_units(args.members)
        b5, verify5, c5, d5, name5 = build_cfg5(args, torch, np, mine5, world, dev, ncpu)
        dt5, ms5, jd5, jc5, res5 = timed_steps(torch, np, dist, coll_dev, b5, verify5, args.cfg5_steps, 1, c5, d5)
        if rank == 0:
            r5 = roof(c5, d5, ms5, SPLIT_WHAT + " (rank 0's shard; member payloads, CRC-32 checked outside the timed region)",
                      (2 if len(res5) <= 16384 else 3 * ((len(res5) + 16383) // 16384)) if len(res5) > 768 else 1)
            line["cfg5_strong"] = {
                "value": jd5 * args.cfg5_steps / dt5 / 1e9,
                "unit": "GB/s",
                "scaling": "strong",
                "n_gpus": world,
                "steps": args.cfg5_steps,
                "warmup": 1,
                "ms_per_step": dt5 / args.cfg5_steps * 1e3,
                "members_job": args.members,
                "members_per_gpu": len(res5),
                "decompressed_bytes_job": jd5,
                "compressed_bytes_job": jc5,
                "workload": name5,
                "bit_exact_checked": "every member's size/good flag + CRC-32 (on the GPU), 3 members byte for byte",
                "roofline": r5,
            }
            if cpu5_line is not None:
                line["cfg5_strong"]["cpu_baseline"] = cpu5_line
        del b5
    # ---- BASELINE config 3 beside it (N = 1 only: 1024 PNGs cycled from the reference's sample files, decoded to RGBA,
    # everything resident in HBM; data files from tests/golden/resources, digests made by the compiled reference)
    if not cfg4 and not cfg5 and not args.no_cfg3 and world == 1 and rank == 0:
        torch.cuda.empty_cache()
        line["cfg3_png"] = cfg3_record(torch, np, dev)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
