"""Data-parallel plumbing shared by bench.py and the tests: one process per GPU, torch.distributed only for the
control plane (rendezvous, the 128-byte RCCL unique id, barriers, timing reductions).  The data-path collective --
one ncclAllReduce(sum) of the padded gradient + loss sums per minibatch -- lives inside libppo_hip (ppo_dist_init)."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_rows(n_rows, world, rank):
    """Contiguous equal shards of a minibatch: rank r owns rows [r*n/world, (r+1)*n/world)."""
    if n_rows % world:
        raise ValueError("minibatch rows %d not divisible by world size %d" % (n_rows, world))
    per = n_rows // world
    return rank * per, (rank + 1) * per


def env_offset(n_envs_per_rank, rank):
    """Weak scaling: rank r simulates the global environments [r*E, (r+1)*E)."""
    return rank * n_envs_per_rank


def broadcast_unique_id(dist, rank, make_uid):
    """Rank 0 creates the ncclUniqueId (bytes of length 128); everybody receives it through the control plane."""
    import torch
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid = make_uid()
        assert len(uid) == 128
        buf = torch.frombuffer(bytearray(uid), dtype=torch.uint8).clone()
    dist.broadcast(buf, 0)
    return bytes(buf.numpy().tobytes())


def allreduce_max(dist, value):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def allgather_bytes(dist, blob, n):
    """Every rank contributes n bytes; returns the list of all ranks' blobs in rank order (IPC handles, digests)."""
    import torch
    assert len(blob) == n
    mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8).clone()
    out = [torch.zeros(n, dtype=torch.uint8) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [bytes(t.numpy().tobytes()) for t in out]
