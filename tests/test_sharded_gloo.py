"""
SURVEY.md §8e on CPU: two ``gloo`` ranks run the product's row-sharded training step and
catalog-sharded evaluation (``collie_amd/sharded.py``) with the oracle as the arithmetic engine and
compare against the oracle on the unsharded problem (tests/sharded_worker.py holds the checks).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


import pytest


@pytest.mark.parametrize('world', [2, 3])
def test_world_size_n_training_and_eval_match_unsharded_oracle(world):
    """world 3: shards, batch slices and wire blocks of unequal sizes."""
    env = dict(os.environ, OMP_NUM_THREADS='2', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'sharded_worker.py')]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    for r in range(world):
        assert f'rank {r}: sharded checks ok' in out.stdout


def test_world_size_1_needs_no_process_group():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from sharded_cpu_engine import CpuOracleEngine
    from collie_amd.sharded import ShardedMatrixFactorizationModel, shard_batch
    from oracle import mf_oracle as O

    U, N, D, K, B = 30, 20, 4, 3, 16
    m = ShardedMatrixFactorizationModel(U, N, K, embedding_dim=D, loss='bpr', device='cpu',
                                        engine=CpuOracleEngine(), seed=1)
    sd = m.state_dict()
    ref = O.TrainState({k[:-7]: v.numpy().copy() for k, v in sd.items()}, lr=1e-3, bias_lr=1e-2)
    rng = np.random.default_rng(0)
    users, pos, neg = rng.integers(0, U, B), rng.integers(0, N, B), rng.integers(0, N, (B, K))
    out = ref.step(users, pos, neg, 'adaptive_bpr', num_items=N)
    batch = ((torch.from_numpy(users), torch.from_numpy(pos)), torch.from_numpy(neg))
    assert shard_batch(batch, 0, 1)[1].shape == (B, K)
    got = m.train_step(batch)
    np.testing.assert_allclose(got.item(), out['loss'], rtol=2e-5)
    np.testing.assert_allclose(m.state_dict()['item_embeddings.weight'].numpy(),
                               ref.t['item_embeddings'], rtol=1e-4, atol=2e-7)


def test_hip_engine_refuses_cpu():
    import pytest
    from collie_amd._lib import CollieHipError
    from collie_amd.sharded import ShardedMatrixFactorizationModel
    with pytest.raises(CollieHipError):
        ShardedMatrixFactorizationModel(10, 10, 2, device='cpu')
