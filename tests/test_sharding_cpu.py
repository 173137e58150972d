"""The N > 1 path on CPU: world_size-2 gloo processes exercise the shard
partition and the single gather of the packed timing profile."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG_NAME


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, N, ragged, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module(PKG_NAME + ".sharding")
    try:
        if not ragged:
            lo, hi = sh.shard_bounds(total, world, rank)
            idx = torch.arange(lo, hi, dtype=torch.float64)
            # packed [4][B_shard][N]: value encodes (field, global path, sample)
            packed = (torch.arange(4, dtype=torch.float64)[:, None, None] * 1e6
                      + idx[None, :, None] * 1e3 + torch.arange(N, dtype=torch.float64)[None, None, :])
            full = sh.gather_packed(packed.contiguous(), dst=0)
            if rank == 0:
                got = torch.cat(list(full.unbind(0)), dim=1)   # [4][total][N]
                q.put(got.numpy())
        else:
            costs = [(i % 5 + 1) ** 2 for i in range(total)]
            bounds = sh.balanced_bounds(costs, world)
            lo, hi = bounds[rank]
            shard = torch.arange(lo, hi, dtype=torch.float64)[:, None].repeat(1, N)
            full = sh.gather_ragged(shard, [b[1] - b[0] for b in bounds], dst=0)
            if rank == 0:
                q.put((full.numpy(), bounds))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _pipeline_worker(rank, world, port, B, N, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module(PKG_NAME + ".sharding")
    try:
        g = sh.PipelinedGather((4, B, N), torch.float64, "cpu", depth=2)
        seen = []
        for k in range(steps):
            buf = g.buffer(k)
            if rank == 0 and k >= 2:           # the result of batch k-2 is complete by now
                seen.append(g.result(k - 2).clone().numpy())
            buf.copy_(torch.full((4, B, N), float(1000 * k + rank)))   # "solve" batch k
            g.launch(k)
        g.drain()
        if rank == 0:
            for k in range(max(0, steps - 2), steps):
                seen.append(g.result(k).clone().numpy())
            q.put(seen)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _hook_worker(rank, world, port, B, N, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module(PKG_NAME + ".sharding")
    try:
        derived = []                            # what the root derives from each landed payload

        def hook(slot):
            derived.append(float(g.recv[slot].sum()))

        g = sh.PipelinedGather((B * N + B,), torch.float64, "cpu", depth=2, on_complete=hook)
        for k in range(steps):
            buf = g.buffer(k)
            buf.copy_(torch.full((B * N + B,), float(10 * k + rank)))
            g.launch(k)
        g.drain()
        if rank == 0:
            q.put(derived)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_calls_the_roots_hook_once_per_batch_in_order():
    world, B, N, steps = 2, 3, 5, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hook_worker, args=(r, world, port, B, N, steps, q))
             for r in range(world)]
    for p in procs:
        p.start()
    derived = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = B * N + B
    assert derived == [float(n * (10 * k + 0) + n * (10 * k + 1)) for k in range(steps)]


def test_pipelined_gather_keeps_batches_apart():
    world, B, N, steps = 2, 3, 5, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, B, N, steps, q))
             for r in range(world)]
    for p in procs:
        p.start()
    seen = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(seen) == steps
    for k, full in enumerate(seen):
        assert full.shape == (world, 4, B, N)
        for r in range(world):
            assert (full[r] == 1000 * k + r).all(), (k, r)


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_gather(ragged):
    world, total, N = 2, 10, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, N, ragged, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if not ragged:
        exp = (np.arange(4)[:, None, None] * 1e6 + np.arange(total)[None, :, None] * 1e3
               + np.arange(N)[None, None, :])
        np.testing.assert_array_equal(res, exp)
    else:
        full, bounds = res
        assert bounds[0][0] == 0 and bounds[-1][1] == total and bounds[0][1] == bounds[1][0]
        np.testing.assert_array_equal(full[:, 0], np.arange(total))


def test_shard_bounds_cover_and_balance():
    sh = importlib.import_module(PKG_NAME + ".sharding")
    for total in (0, 1, 7, 1024, 65536, 1000):
        for world in (1, 2, 4, 8):
            b = [sh.shard_bounds(total, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    costs = [500 + 7 * (i % 500) for i in range(4096)]
    b = sh.balanced_bounds(costs, 8)
    loads = [sum(costs[lo:hi]) for lo, hi in b]
    assert b[0][0] == 0 and b[-1][1] == 4096
    assert max(loads) / (sum(loads) / 8) < 1.02
