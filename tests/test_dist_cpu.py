"""world_size-2 gloo tests (CPU): the data-parallel decomposition of the train step is exact -- summing the shard
gradients computed with the GLOBAL 1/N equals the full-minibatch gradient, loss sums add up, the seeded env shards
tile the global env set, and the control-plane helpers (unique-id broadcast, max-reduce) work."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as o
from ppo_cpp_amd import dist as ppodist
from tests import helpers as H

CR = 0.16102319955825806


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = o.Oracle(18, 18, [64, 64]); orc.init_orthogonal(3)
        n = 128
        mb = H.synth_minibatch(orc, n, seed=5)                                # identical on every rank (same seed)
        lo, hi = ppodist.shard_rows(n, world, rank)
        sl = {k: v[lo:hi] for k, v in mb.items()}
        losses, grad = orc.loss_grad(sl["obs"], sl["actions"], sl["advs"], sl["returns"], sl["old_neglogp"], sl["old_values"], CR)
        # local gradient is of the LOCAL mean (1/n_local); the library uses 1/(n_local*world) and sums: same thing / world
        payload = torch.from_numpy(np.concatenate([grad.astype(np.float64) / world, losses.astype(np.float64) / world]))
        dist.all_reduce(payload, op=dist.ReduceOp.SUM)
        full_losses, full_grad = orc.loss_grad(mb["obs"], mb["actions"], mb["advs"], mb["returns"], mb["old_neglogp"], mb["old_values"], CR)
        g = payload.numpy()
        np.testing.assert_allclose(g[:-5], full_grad, rtol=2e-4, atol=2e-6 * np.abs(full_grad).max())
        np.testing.assert_allclose(g[-5:], full_losses, rtol=1e-5, atol=1e-7)
        # every rank ends with bit-identical reduced data -> replicated clip + Adam stay in lock step
        gathered = [torch.zeros_like(payload) for _ in range(world)]
        dist.all_gather(gathered, payload)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        # control plane helpers
        uid = ppodist.broadcast_unique_id(dist, rank, lambda: bytes(range(128)))
        assert uid == bytes(range(128))
        assert ppodist.allreduce_max(dist, 1.0 + rank) == float(world)
        assert ppodist.allgather_bytes(dist, bytes([rank]) * 64, 64) == [bytes([r]) * 64 for r in range(world)]   # IPC-handle exchange
        # the one-shot peer all-reduce's arithmetic: every rank adds the W gathered vectors in RANK order (not "mine first"),
        # so all replicas get the same bits even where fp32 addition does not commute across three or more terms
        rngp = np.random.RandomState(100 + rank)
        mine = (rngp.normal(size=4096) * 10.0 ** rngp.uniform(-6, 6, 4096)).astype(np.float32)
        slots = [torch.zeros(4096) for _ in range(world)]
        dist.all_gather(slots, torch.from_numpy(mine))
        acc = slots[0].numpy().copy()
        for r in range(1, world):
            acc = (acc + slots[r].numpy()).astype(np.float32)
        sums = [torch.zeros(4096) for _ in range(world)]
        dist.all_gather(sums, torch.from_numpy(acc))
        assert all(torch.equal(sums[0], t) for t in sums)
        # weak-scaling env shards tile the global env set of the seeded synthetic env
        E = 8
        mine, _, _ = o.seeded_env_step(1234, ppodist.env_offset(E, rank), E, 3, 18)
        allenv, _, _ = o.seeded_env_step(1234, 0, E * world, 3, 18)
        np.testing.assert_array_equal(mine, allenv[rank * E:(rank + 1) * E])
        out.put((rank, "ok"))
    except Exception as e:       # noqa
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_decomposition_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_helpers():
    assert ppodist.shard_rows(2048, 8, 3) == (768, 1024)
    with pytest.raises(ValueError):
        ppodist.shard_rows(100, 8, 0)
    assert ppodist.env_offset(4096, 5) == 20480
