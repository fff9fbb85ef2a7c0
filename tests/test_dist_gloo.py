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
    # unequal per-rank blocks must fail loudly on every rank instead of hanging / corrupting
    try:
        gather_views(torch.zeros(1 + rank, 2))
        raise AssertionError('uneven blocks were accepted')
    except ValueError:
        pass
    # ... also when one rank's block size is one it has gathered before (a per-rank cache of agreed sizes sent
    # rank 0 straight into the all-gather while rank 1 issued the check's all-reduce -- both ranks timed out)
    gather_views(torch.zeros(4, 2))
    try:
        gather_views(torch.zeros(4 - rank, 2))
        raise AssertionError('a ragged step after an even one was accepted')
    except ValueError:
        pass
    # the incoming gradient buffer is not reduced in place
    x = torch.ones(2, 2, requires_grad=True)
    upstream = torch.ones(4, 2)
    keep = upstream.clone()
    gather_views(x).backward(upstream)
    assert torch.equal(upstream, keep) and torch.equal(x.grad, torch.full((2, 2), 2.0))
    assert shard_range(6, group=dist.group.WORLD) == (rank * 3, rank * 3 + 3)
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


def _worker_direct(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gendr_amd.dist import shard_batch, gather_views
    torch.manual_seed(3)
    full = torch.randn(3 * world, 4, 5)
    w = torch.randn(world, 3 * world, 4, 5)[rank]            # a different loss on every rank: the reduce-scatter is visible
    res = {}
    for direct in (False, True):
        mine = shard_batch(full).clone().requires_grad_(True)
        views = gather_views(mine * 2.0, assume_equal_blocks=True, direct=direct)
        assert torch.equal(views, full * 2.0), direct
        (views * w).sum().backward()
        res[direct] = mine.grad.clone()
    # the one-round point-to-point form: the same views bit for bit, the gradient to the order of the sum over ranks
    assert torch.allclose(res[True], res[False], rtol=1e-6, atol=1e-6)
    os.environ['GENDR_ALLGATHER'] = 'direct'                 # the environment knob selects it where the caller does not
    assert torch.equal(gather_views(shard_batch(full).clone(), assume_equal_blocks=True), full)
    out.put(rank)
    dist.destroy_process_group()


def test_direct_all_gather_equals_the_collective():
    """SURVEY 8(e) / VERDICT r5 missing 4: the all-gather of views as one round of point-to-point transfers (xGMI is point to
    point: a ring serialises N - 1 hops) -- same result as the library's collective, forward and backward; world sizes 2 and 3."""
    for world in (2, 3):
        ctx = mp.get_context('spawn')
        out = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_direct, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert sorted(out.get() for _ in range(world)) == list(range(world))


def test_shard_range_uneven():
    from gendr_amd.dist import shard_range
    spans = [shard_range(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_range(5, 0, 1) == (0, 5)


def test_bench_launcher_world2_stub():
    """`bench.py --gpus 2` outside torchrun starts its own two ranks (gloo, stub step: no GPU here) and rank 0
    prints one JSON line with n_gpus == 2, STRONG scaling (the config's batch sharded evenly: SURVEY 8(d)) as the headline, plus the
    weak figure (the config's batch per rank) under extra."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--stub', '--batch', '6'], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['scaling'] == 'strong' and j['steps'] == 3 and j['warmup'] == 1
    assert j['config']['global_batch'] == 6 and j['extra']['weak']['global_batch'] == 12
    assert j['value'] > 0 and j['unit'] == 'frames/s'


def test_bench_refuses_world_mismatch():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--stub'],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and 'WORLD_SIZE' in (out.stderr + out.stdout)
