"""Sharding of independent streams over the GPUs of one node.

Every stream / image / gzip member is independent, so the data path has NO collective:
rank 0 builds the shard map {stream id -> rank, local index} (round-robin, as BASELINE
config 5 asks: member i -> GPU i mod n) and broadcasts it -- over RCCL ("nccl" backend) on
GPUs, gloo in CPU tests.  Each rank then inflates its own shard into its own HBM.
"""
import numpy as np


def build_shard_map(world, n_streams):
    """[n_streams, 3] int64: global id, owning rank, index within that rank's shard."""
    ids = np.arange(n_streams, dtype=np.int64)
    return np.stack([ids, ids % world, ids // world], axis=1)


def broadcast_shard_map(n_streams, device, dist=None):
    """rank 0 builds, everyone receives.  dist = torch.distributed (initialised) or None."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    smap = torch.zeros((n_streams, 3), dtype=torch.int64, device=device)
    if rank == 0:
        smap.copy_(torch.from_numpy(build_shard_map(world, n_streams)))
    if dist is not None:  # world size 1 included: the collective runs whenever a process group exists
        dist.broadcast(smap, src=0)
    return smap


def my_streams(smap, rank):
    """global ids owned by `rank`, ordered by local index"""
    mine = smap[smap[:, 1] == rank]
    order = mine[:, 2].argsort()
    return mine[order][:, 0].cpu().numpy()
