"""world_size-2 checks of the batch sharding and of the differentiable view all-gather on gloo (CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gendr_amd.dist import shard_batch, shard_range, gather_views, sum_over_ranks
    full = torch.arange(6 * 3, dtype=torch.float32).reshape(6, 3)
    mine = shard_batch(full).clone().requires_grad_(True)
    assert shard_range(6) == (rank * 3, rank * 3 + 3)
    views = gather_views(mine * 2.0)
    assert torch.equal(views, full * 2.0)
    # a loss that couples every view: weights differ per rank so that the reduce-scatter is visible
    w = torch.arange(18, dtype=torch.float32).reshape(6, 3) * (rank + 1)
    (views * w).sum().backward()
    expect = 2.0 * torch.arange(18, dtype=torch.float32).reshape(6, 3)[rank * 3:rank * 3 + 3] * (1 + 2)
    assert torch.allclose(mine.grad, expect), (mine.grad, expect)
    g = torch.ones(4) * (rank + 1)
    sum_over_ranks(g)
    assert torch.equal(g, torch.full((4,), 3.0))
    out.put(rank)
    dist.destroy_process_group()


def test_gather_views_world2():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(out.get() for _ in range(2)) == [0, 1]


def test_shard_range_uneven():
    from gendr_amd.dist import shard_range
    spans = [shard_range(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_range(5, 0, 1) == (0, 5)
