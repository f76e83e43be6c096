#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on BASELINE config 2 (configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (per rank; SURVEY.md 8d "cfg2"): 4096 fixed-Huffman streams + 4096
stored-block streams, every stream inflating to one 64 KiB block; streams are
generated deterministically (tools/streamgen.c, seed 0xDEB16 + global stream index).
One "step" = one pass of the hot path (ONE launch of the batched inflate kernel through
the C-ABI, include/debig_hip.h) over the whole batch, inputs already resident in HBM.

  value      decompressed GB/s, whole job  = sum over ranks of D bytes / max-over-ranks time
  roofline   that launch: algorithmic bytes C + D of the batch / average launch duration
             (events on the launch stream); --variants adds each stream kind timed alone
  cpu_baseline  the compiled reference (oracle/_ref, single thread) or, if that prebuilt
             library is absent, the oracle port -- timed on a bounded sample of the same streams

Multi-GPU (--gpus N, launched by torch.distributed.run): one process per GPU; rank 0
builds the shard map (stream id -> rank, offsets) and broadcasts it over RCCL; every
rank inflates its own shard; no payload collective (weak scaling: per-GPU work fixed).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable
STREAMS_PER_KIND = 4096
STREAM_BYTES = 65536


def cpu_baseline(sample_fixed, sample_stored):
    """Reference (or oracle port) single-thread throughput on a bounded sample."""
    from oracle import binding

    kind = "port"
    eng = None
    if binding.ref_available("A"):
        try:
            eng = binding.Reference("A")
            kind = "reference"
        except OSError:
            eng = None
    if eng is None:
        eng = binding.Oracle()
    nbytes = 0
    passes = 0
    t0 = time.perf_counter()
    while True:  # whole passes over the sample until ~10 s of single-thread CPU work
        for raw, plain in sample_fixed + sample_stored:
            cap = max(len(plain) + 1, len(raw))
            out = eng.inflate(raw, cap)
            assert out[0] == 1 and out[1] == len(plain)
            nbytes += len(plain)
        passes += 1
        if time.perf_counter() - t0 > 10.0 or passes >= 8:
            break
    dt = time.perf_counter() - t0
    return {
        "value": nbytes / dt / 1e9,
        "unit": "GB/s decompressed",
        "cores": 1,
        "kind": kind,
        "sample": f"{len(sample_fixed)} fixed-Huffman + {len(sample_stored)} stored streams of 64 KiB "
                  f"(same generator), {passes} pass(es), 1 thread, {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=STREAMS_PER_KIND, help="streams per kind per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=64, help="streams per kind checked against the generator")
    ap.add_argument("--variants", action="store_true", help="also time each stream kind alone (untimed extra)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N>1 code path on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.one_device:
            local_rank = 0
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    coll_dev = dev if args.backend == "nccl" else "cpu"  # where collective tensors live

    from debigulator_amd import workload
    from debigulator_amd.batch import DeviceBatch

    per = args.streams
    # ---- shard map: rank 0 builds it, RCCL broadcasts it (the only collective on the path)
    from debigulator_amd import shard

    smap = shard.broadcast_shard_map(world * per, coll_dev, dist if world > 1 else None)
    mine = shard.my_streams(smap, rank)  # global stream ids of this rank
    assert len(mine) == per

    # ---- synthesise this rank's shard (deterministic) and park it in HBM
    ncpu = max(1, min(16, (os.cpu_count() or 8) // max(1, min(world, 8))))
    from concurrent.futures import ThreadPoolExecutor

    pairs_keep = {}
    for kind in ("fixed", "stored"):
        with ThreadPoolExecutor(ncpu) as ex:
            pairs_keep[kind] = list(ex.map(lambda g: workload.make_stream(kind, int(g), STREAM_BYTES), mine))
    # ONE batch = this rank's whole shard: the long-running Huffman streams first, the stored
    # ones behind them (workgroups are dispatched in stream order)
    all_pairs = pairs_keep["fixed"] + pairs_keep["stored"]
    raws = [p[0] for p in all_pairs]
    caps = [max(STREAM_BYTES + 1, len(r)) for r in raws]
    batch = DeviceBatch.from_streams(raws, caps, device=dev)
    c_fixed = sum(len(p[0]) for p in pairs_keep["fixed"])
    c_stored = sum(len(p[0]) for p in pairs_keep["stored"])
    c_bytes = c_fixed + c_stored
    d_bytes = STREAM_BYTES * len(raws)

    for _ in range(args.warmup):
        batch.launch()
    torch.cuda.synchronize()

    # ---- bit-exactness gate (untimed): sizes/flags of every stream, bytes of a sample
    res = batch.results()
    assert (res["good"] == 1).all(), "a stream failed"
    assert (res["final_size"] == STREAM_BYTES).all(), "wrong size"
    for base in (0, per):
        for i in range(base, base + args.verify):
            assert batch.output(i, res) == all_pairs[i][1].tobytes(), f"stream {i} differs"

    # ---- timed region: exactly K steps between barrier+sync on both sides
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        batch.launch()
        ev[k][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    launch_ms = float(np.mean([ev[k][0].elapsed_time(ev[k][1]) for k in range(args.steps)]))

    variants = None
    if args.variants and rank == 0:  # untimed extra: each kind launched alone
        variants = {}
        for kind, lo, cb in (("fixed_huffman", 0, c_fixed), ("stored", per, c_stored)):
            sub = DeviceBatch.from_streams(raws[lo:lo + per], caps[lo:lo + per], device=dev)
            for _ in range(2):
                sub.launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                sub.launch()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            variants[kind] = {"kernel_ms": ms, "decompressed_GBps": per * STREAM_BYTES / ms / 1e6,
                              "roofline_frac": (cb + per * STREAM_BYTES) / ms / 1e6 / HBM_PEAK_GBS}

    if rank == 0:
        value = world * d_bytes * args.steps / dt_max / 1e9
        alg = c_bytes + d_bytes
        ach = alg / (launch_ms * 1e-3) / 1e9
        rf = res[:per]
        # HBM traffic per launch from the PMC passes of this same command (profiles/pmc_traffic.json,
        # made by tools/pmc_traffic.sh + tools/pmc_summary.py); null if that file is absent
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = tj["bytes_per_launch"] * (per / STREAMS_PER_KIND)
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "decompressed GB/s (whole node) + % HBM roofline, bit-exact vs reference",
            "value": value,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"cfg2: per GPU {per} fixed-Huffman + {per} stored DEFLATE streams, 64 KiB each, one "
                            f"batch = one kernel launch (one 64 KiB block per stream; stored = 65535+1 byte "
                            f"blocks), seed 0xDEB16+i",
                "streams_per_gpu": 2 * per,
                "decompressed_bytes_per_gpu": d_bytes,
                "compressed_bytes_per_gpu": c_bytes,
                "fixed_huffman_ratio": per * STREAM_BYTES / c_fixed,
                "sharding": "round-robin by stream id, shard map broadcast over RCCL, no payload collective",
                "bit_exact_checked": f"{args.verify} streams per kind byte-for-byte + all sizes/good flags",
                "avg_spec_rounds_per_window": float(rf["n_rounds"].sum()) / max(1, float(rf["n_windows"].sum())),
            },
            "roofline": {
                "kernel": "debig_inflate_kernel (the step's only launch)",
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg,
                "avg_launch_ms": launch_ms,
            },
        }
        if variants:
            line["variants"] = variants
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            try:
                line["cpu_baseline"] = cpu_baseline(pairs_keep["fixed"], pairs_keep["stored"])
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "GB/s decompressed", "cores": 1, "kind": "port",
                                        "sample": f"failed: {e}"}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
