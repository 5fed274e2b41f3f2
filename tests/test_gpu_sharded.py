"""
GPU tests (``-m gpu``) of the row-sharded training step and the catalog-sharded evaluation
(``collie_amd/sharded.py``) with the HIP engine:
  * world size 1: the sharded step against the single-GPU product model on the same batches
    (same kernels underneath, different route: compact tables + owner-side COO update);
  * world size 2 on the one GPU of the test box (gloo, host-staged collectives): the checks of
    tests/sharded_worker.py against the oracle on the unsharded problem.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
from scipy.sparse import coo_matrix

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')


class Shape:
    def __init__(self, U, N, K):
        self.num_users, self.num_items, self.num_negative_samples = U, N, K


def _step(model, opts, batch):
    for o in opts:
        o.zero_grad()
    loss = model.calculate_loss(batch)
    loss.backward()
    for i, o in enumerate(opts):
        model.optimizer_step(optimizer=o, optimizer_idx=i)
    return loss


@pytest.mark.parametrize('users_local', [False, True])
@pytest.mark.parametrize('loss,optimizer,K,D', [('bpr', 'sparse_adam', 10, 64),
                                                ('warp', 'adagrad', 5, 32),
                                                ('hinge', 'sgd', 1, 10)])
def test_world_1_sharded_step_matches_single_gpu_model(loss, optimizer, K, D, users_local):
    """users_local: user rows are read and updated in place, only item rows take the exchange."""
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.sharded import ShardedMatrixFactorizationModel
    dev = torch.device('cuda:0')
    U, N, B = 5000, 3000, 4096
    torch.manual_seed(0)
    single = MatrixFactorizationModel(train=Shape(U, N, K), embedding_dim=D, sparse=True, lr=1e-2,
                                      bias_lr=1e-2, optimizer=optimizer, bias_optimizer='sgd',
                                      loss=loss, lr_scheduler_func=None).to(dev).train()
    with torch.no_grad():
        single.user_embeddings.weight.mul_(10.0)
        single.item_embeddings.weight.mul_(10.0)
        single.item_biases.weight.normal_(0, 0.01)
    sharded = ShardedMatrixFactorizationModel.from_model(single, K, device=dev)
    opts = single.configure_optimizers()
    rng = np.random.default_rng(4)
    for it in range(3):
        users = torch.tensor(rng.zipf(1.3, B) % U, dtype=torch.int32, device=dev)
        pos = torch.tensor(rng.zipf(1.2, B) % N, dtype=torch.int32, device=dev)
        neg = torch.tensor(rng.integers(0, N, (B, K)), device=dev)
        l1 = _step(single, opts, ((users, pos), neg))
        l2 = sharded.train_step(((users, pos), neg), users_local=users_local)
        np.testing.assert_allclose(l2.item(), l1.item(), rtol=1e-6)
    single.check_status()
    # at most user + positive + one negative per sample carry a gradient: only those go home
    sent = sharded.last_exchange['gradient_rows_sent']
    assert 0 < sent <= (2 if users_local else 3) * B
    assert sent < sharded.last_exchange['rows_requested'] or K == 1
    sd = sharded.state_dict()
    for name in NAMES:
        want = getattr(single, name).weight.detach().cpu().numpy()
        have = sd[f'{name}.weight'].cpu().numpy()
        # same per-sample terms; a row's sum is formed in batch order on both routes
        np.testing.assert_allclose(have, want, rtol=1e-4, atol=2e-7, err_msg=name)


def test_world_1_sharded_eval_matches_single_gpu_eval():
    from collie_amd import metrics as M
    from collie_amd.interactions import Interactions
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.sharded import ShardedMatrixFactorizationModel
    dev = torch.device('cuda:0')
    U, N, D = 700, 900, 32
    torch.manual_seed(1)
    single = MatrixFactorizationModel(train=Shape(U, N, 3), embedding_dim=D, sparse=True,
                                      optimizer='sparse_adam', loss='bpr',
                                      lr_scheduler_func=None).to(dev)
    with torch.no_grad():
        single.user_embeddings.weight.mul_(20.0)
        single.item_embeddings.weight.mul_(20.0)
        single.item_biases.weight.normal_(0, 0.01)
        single.user_biases.weight.normal_(0, 0.01)
    sharded = ShardedMatrixFactorizationModel.from_model(single, 3, device=dev)
    rng = np.random.default_rng(5)
    rows = np.repeat(np.arange(0, U, 2), 6)
    cols = rng.integers(0, N, rows.shape[0])
    test = Interactions(users=rows, items=cols, num_users=U, num_items=N, allow_missing_ids=True,
                        num_negative_samples=1, max_number_of_samples_to_consider=0, seed=0)
    want = M.evaluate_in_batches([M.mapk, M.mrr, M.auc], test, single, k=10, verbose=False)
    got = sharded.evaluate_in_batches([M.mapk, M.mrr, M.auc], test, k=10)
    assert got == want          # same kernels, same order: identical, not merely close


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_world_2_on_one_gpu_matches_unsharded_oracle():
    env = dict(os.environ, OMP_NUM_THREADS='2', HSA_ENABLE_IPC_MODE_LEGACY='0',
               COLLIE_SHARDED_ENGINE='hip')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'sharded_worker.py')]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert 'rank 0: sharded checks ok' in out.stdout and 'rank 1: sharded checks ok' in out.stdout


@pytest.mark.parametrize('shape', [(1, 1, 5, 2), (3, 63, 40, 1), (64, 64, 300, 4), (65, 129, 7, 63),
                                   (16385, 16383, 2000, 3)])
@pytest.mark.parametrize('G', [1, 5, 16])
def test_hip_shard_plan_edge_sizes(shape, G):
    """Key spaces that end inside / at / just behind a 64-key word or a 16,384-key workgroup range,
    one-row tables, the largest world: the marks-and-ranks route against plan_torch."""
    from collie_amd import ops
    from collie_amd.sharded import plan_torch
    dev = torch.device('cuda:0')
    U, N, B, K = shape
    D = 8
    rng = np.random.default_rng(U + N + G)
    planner = ops.ShardPlanner(U, N, G, D, dev)
    for local_rank in (-1, 0):
        users = rng.integers(0, U, B)
        if local_rank >= 0:
            users = users // G * G                             # users owned by rank 0
        users = torch.tensor(users, dtype=torch.int64, device=dev)
        pos = torch.tensor(rng.integers(0, N, B), dtype=torch.int32, device=dev)
        neg = torch.tensor(rng.integers(0, N, (B, K)), device=dev)
        cu, cp, cn, rows, counts = planner.plan(users, pos, neg, local_rank)
        tu, tp, tn, trows, tcounts = plan_torch(users, pos, neg, U, N, G, D, local_rank)
        assert torch.equal(counts, tcounts)
        n = int(counts[2 * G])
        assert torch.equal(rows[:n], trows)
        assert torch.equal(cu, tu) and torch.equal(cp, tp) and torch.equal(cn, tn)


@pytest.mark.parametrize('local_rank', [-1, 0, 2])
@pytest.mark.parametrize('G', [1, 2, 3, 8])
@pytest.mark.parametrize('neg_dtype', [torch.int64, torch.int32])
@pytest.mark.parametrize('route', ['direct', 'sort', 'direct_many_blocks'])
def test_hip_shard_plan_equals_torch_plan(G, neg_dtype, local_rank, route):
    from collie_amd import ops
    from collie_amd.sharded import plan_torch
    dev = torch.device('cuda:0')
    # the library marks the looked-up rows of tables with at most 64 rows per lookup ("direct"),
    # and sorts the ids of sparser batches: same outputs either way
    U, N, B, K, D = {'direct': (9001, 5003, 3000, 7, 24),
                     'sort': (3_000_001, 2_000_003, 500, 3, 24),
                     'direct_many_blocks': (1_200_001, 700_001, 40_000, 5, 8)}[route]
    if local_rank >= G:
        pytest.skip('rank outside the world')
    rng = np.random.default_rng(G)
    planner = ops.ShardPlanner(U, N, G, D, dev)
    for it in range(3):                                   # the sort scratch is reused
        users = rng.zipf(1.2, B) % U
        if local_rank >= 0:                               # users this rank owns
            users = np.minimum(users // G * G + local_rank, (U - 1 - local_rank) // G * G + local_rank)
        users = torch.tensor(users, dtype=torch.int32, device=dev)
        pos = torch.tensor(rng.integers(0, N, B), dtype=torch.int64, device=dev)
        neg = torch.tensor(rng.integers(0, N, (B, K)), device=dev).to(neg_dtype)
        if it == 2:
            neg = neg.t().contiguous().t()                # strided negatives
        cu, cp, cn, rows, counts = planner.plan(users, pos, neg, local_rank)
        tu, tp, tn, trows, tcounts = plan_torch(users, pos, neg, U, N, G, D, local_rank)
        assert int(counts[2 * G + 1]) == 0
        assert torch.equal(counts, tcounts)
        n = int(counts[2 * G])
        assert torch.equal(rows[:n], trows)
        assert torch.equal(cu, tu) and torch.equal(cp, tp) and torch.equal(cn, tn)
    bad = pos.clone()
    bad[5] = N
    counts = planner.plan(users, bad, neg, local_rank)[4]
    assert int(counts[2 * G + 1]) == 1
    if local_rank >= 0 and G > 1:                         # a user of another rank is reported too
        foreign = users.clone()
        foreign[7] = (local_rank + 1) % G
        assert int(planner.plan(foreign, pos, neg, local_rank)[4][2 * G + 1]) == 1
        assert int(plan_torch(foreign, pos, neg, U, N, G, D, local_rank)[4][2 * G + 1]) == 1
