"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- rank 0 builds the shard map,
broadcasts it, every rank takes its round-robin share; no payload collective."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from debigulator_amd import shard, workload

    smap = shard.broadcast_shard_map(n_streams, "cpu", dist)
    mine = shard.my_streams(smap, rank)
    # every rank synthesises exactly its own streams (deterministic by global id)
    digests = [int(np.frombuffer(workload.make_stream("fixed", int(g), 4096)[0][:8], dtype=np.uint64)[0] % (1 << 31))
               for g in mine[:4]]
    t = torch.tensor([len(mine)], dtype=torch.int64)
    dist.all_reduce(t)  # bench.py's only other collective: timing/size reductions
    q.put((rank, mine.tolist(), digests, int(t.item()), smap.numpy().tolist()))
    dist.destroy_process_group()


def test_shard_map_broadcast_world2():
    world, n = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, m0, d0, tot0, map0), (r1, m1, d1, tot1, map1) = got
    assert map0 == map1  # everyone holds rank 0's map
    assert m0 == [0, 2, 4, 6, 8] and m1 == [1, 3, 5, 7, 9]  # member i -> GPU i mod n
    assert tot0 == tot1 == n
    assert d0 != d1  # different shards, different data
