"""The N > 1 path on the REAL backend: one rank per GPU over `nccl` (= RCCL over xGMI).  Needs >= 2 GPUs on the box and
skips visibly otherwise (the 1-GPU boxes of the round-end GPU tier); the same code path runs on CPU under gloo in
tests/test_sharding_gloo.py and with two ranks on one GPU in tests/test_stress_gpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0
needs_two = pytest.mark.skipif(NGPU < 2, reason=f'needs >= 2 GPUs for one nccl rank per device ({NGPU} visible)')


def _port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@needs_two
@pytest.mark.parametrize('world', [2, 4, 8])
def test_sharded_matmuls_over_nccl_match_single_gpu_bits(world):
    if NGPU < world:
        pytest.skip(f'{world} ranks need {world} GPUs ({NGPU} visible)')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr',
           '127.0.0.1', '--master-port', str(_port()), os.path.join(ROOT, 'tools', 'nccl_ranks.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert out.returncode == 0, out.stderr[-3000:]
    assert f'nccl ranks ok: backend nccl world {world}' in out.stdout, out.stdout[-2000:]


@needs_two
def test_bench_self_launches_nccl_ranks():
    """`python bench.py --gpus 2` with no launcher around it becomes two nccl ranks (bench.py re-executes itself under
    torch.distributed.run) and reports n_gpus = 2 with the RCCL all-gather legs."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--scale', '0.1', '--no-sampler', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert r['n_gpus'] == 2 and r['allgather']['backend'] == 'nccl' and 'incl_allgather' in r['c4'], r


def test_bench_refuses_more_gpus_than_visible():
    """Asking for more ranks than devices fails loudly instead of silently running one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(NGPU + 1), '--steps', '1',
                          '--warmup', '0'], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and 'HIP device(s) are visible' in out.stderr, (out.returncode, out.stderr[-500:])
