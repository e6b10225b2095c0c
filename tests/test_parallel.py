"""Data-parallel layer on CPU: world_size-2 gloo process groups."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cmda_amd.parallel import GradAllReducer, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, wire, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(rank)
    flat = torch.randn(100003)
    mine = flat.clone()
    red = GradAllReducer(flat, bucket_elems=30000, wire_dtype=wire)
    assert len(red.buckets) == 4
    red.all_reduce_mean()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ref = sum(gathered) / world
    tol = 1e-6 if wire == torch.float32 else 2e-2
    ok = torch.allclose(flat, ref, rtol=tol, atol=tol)
    # data-parallel identity: the gradient of the global-batch mean equals the mean of the per-rank gradients
    w = torch.ones(5, requires_grad=True)
    torch.manual_seed(123)
    data = torch.randn(world * 2, 5)
    lo, hi = shard_range(world * 2, rank, world)
    (data[lo:hi] * w).pow(2).mean().backward()
    g = w.grad.clone()
    GradAllReducer(g).all_reduce_mean()
    w2 = torch.ones(5, requires_grad=True)
    (data * w2).pow(2).mean().backward()
    ok = ok and torch.allclose(g, w2.grad, atol=1e-6)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize('wire', [torch.float32, torch.bfloat16])
def test_grad_allreduce_world2(wire):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), wire, out), nprocs=world, join=True)
        assert all(out[r] for r in range(world))


def test_shard_range():
    assert [shard_range(16, r, 8) for r in range(8)] == [(2 * r, 2 * r + 2) for r in range(8)]
