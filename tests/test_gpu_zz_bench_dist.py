"""bench.py's multi-rank plumbing on the GPU (the driver launches it with N > 1 only when an 8-GPU node is free, so nothing else in the
suite touches these calls on a GPU): the RCCL process group of one rank (GENDR_BENCH_FORCE_DIST=1: init with device_id, barrier,
all-reduce of the elapsed time, BASELINE config 4's all-gather / reduce-scatter of views over backend nccl), and the driver's own
launch line -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` -- with both ranks on this GPU over gloo
(GENDR_BENCH_OVERSUBSCRIBE=1: RCCL refuses two ranks on one device).
(File name: last in the suite's order -- the driver runs `pytest -x`, and a launcher problem of the box must not hide the parity tests.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("config,batch", [('c2', 16), ('c4', 8)])
def test_one_rank_over_rccl(native_lib, config, batch):
    env = dict(os.environ, GENDR_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', config, '--batch', str(batch), '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--no-extra'], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d['config']['backend'] == 'nccl' and d['n_gpus'] == 1 and d['value'] > 0 and d['steps'] == 3
    assert d['config']['global_batch'] == batch and d['scaling'] == 'strong'
    if config == 'c4':
        assert 'all-gather' in d['config']['workload'] and d['config']['launch'] == 'eager'      # a collective in the step: not captured


def test_two_ranks_by_the_drivers_launch_line(native_lib):
    env = dict(os.environ, GENDR_BENCH_OVERSUBSCRIBE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('MASTER_PORT', None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'c4', '--batch', '8',
                        '--steps', '3', '--warmup', '1'], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d['n_gpus'] == 2 and d['config']['backend'] == 'gloo' and d['value'] > 0
    assert d['config']['global_batch'] == 8 and '(4 on rank 0)' in d['config']['workload']
    assert 'all-gather' in d['config']['parallelism'] or 'all-gather' in d['config']['workload']
