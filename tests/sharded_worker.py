"""
One rank of the world_size-2 ``gloo`` test of ``collie_amd.sharded`` (started by
``tests/test_sharded_gloo.py`` through ``torch.distributed.run``).  The exchange code is the
product's; the arithmetic engine is the CPU oracle (tests/sharded_cpu_engine.py).  Every rank
checks the sharded result against the oracle run on the unsharded tables and the global batch.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
from scipy.sparse import coo_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from collie_amd import metrics as M                      # noqa: E402
from collie_amd.sharded import (ShardedMatrixFactorizationModel, counter_normal,  # noqa: E402
                                route_batch_to_user_owners, shard_batch)
from oracle import mf_oracle as O                        # noqa: E402
from sharded_cpu_engine import CpuOracleEngine           # noqa: E402

# COLLIE_SHARDED_ENGINE=hip: the same checks with the HIP engine on cuda:0 (both ranks share the one
# GPU of the test box; gloo stages the collectives through the host)
HIP = os.environ.get('COLLIE_SHARDED_ENGINE') == 'hip'
DEV = 'cuda:0' if HIP else 'cpu'


def engine():
    return None if HIP else CpuOracleEngine()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


class _Inter:
    def __init__(self, mat):
        self.mat = mat
        self.num_users, self.num_items = mat.shape


def full_tables(U, N, D, seed):
    rng = np.random.default_rng(seed)
    t = O.init_tables(U, N, D, rng)
    t['user_biases'] = (rng.standard_normal((U, 1)) * 0.01).astype(np.float32)
    t['item_biases'] = (rng.standard_normal((N, 1)) * 0.01).astype(np.float32)
    return t


def check_training(rank, world, loss, optimizer, bias_optimizer, K, metadata=None, route=False):
    """route: every sample is moved to the rank that owns its user (route_batch_to_user_owners) and
    trained with users_local=True -- same global batch, same result."""
    U, N, D, B, steps = 53, 41, 8, 64, 3
    t0 = full_tables(U, N, D, 1)
    weights = {'genre': 0.4} if metadata is not None else None
    model = ShardedMatrixFactorizationModel(
        U, N, K, embedding_dim=D, lr=1e-2, bias_lr=1e-2, optimizer=optimizer,
        bias_optimizer=bias_optimizer, loss=loss, metadata_for_loss=metadata,
        metadata_for_loss_weights=weights, device=DEV, engine=engine(), init=False)
    model.load_full_state_dict({f'{k}.weight': dev(v.copy()) for k, v in t0.items()})
    ref = O.TrainState(t0, optimizer=optimizer, bias_optimizer=bias_optimizer, lr=1e-2,
                       bias_lr=1e-2, sparse=True)
    kind = model._loss_kind
    rng = np.random.default_rng(7)
    host, slices = [], []
    for step in range(steps):
        users = rng.integers(0, U, B)
        pos = rng.integers(0, N, B)
        neg = rng.integers(0, N, (B, K))
        if step == 1:
            users[:8] = users[0]                      # duplicates across the two slices
            pos[B // 2:B // 2 + 4] = pos[0]
        host.append((users, pos, neg))
        batch = ((dev(users.astype(np.int32)), dev(pos.astype(np.int32))), dev(neg))
        piece = shard_batch(batch, rank, world)
        if route:
            piece = route_batch_to_user_owners(piece)
            assert bool((piece[0][0].long() % world == rank).all())
        slices.append(piece)
    for step in range(steps):
        users, pos, neg = host[step]
        kw = {}
        if metadata is not None:
            kw = dict(metadata={'genre': metadata['genre'].cpu().numpy()}, metadata_weights=weights)
        out = ref.step(users, pos, neg, kind, num_items=N, **kw)
        # the plan of a step is made, in turn: inside the step itself, on a side stream after the
        # previous step (plan_ahead), inside the previous step (next_batch=)
        nxt = slices[step + 1] if step % 3 == 1 and step + 1 < steps else None
        got = model.train_step(slices[step], users_local=route, next_batch=nxt)
        if step % 3 == 0 and step + 1 < steps:
            model.plan_ahead(slices[step + 1], users_local=route)
        np.testing.assert_allclose(got.item(), out['loss'], rtol=2e-5, atol=1e-7)
    sd = model.state_dict()
    for name, want in ref.t.items():
        have = sd[f'{name}.weight'].cpu().numpy()
        assert have.shape == want.shape
        if name == 'user_biases':
            # the reference's d loss / d user_bias is fp32 cancellation residue (ours is exactly
            # 0); Adam-type optimizers turn that residue into +-lr random steps
            noise = 2e-6 if bias_optimizer == 'sgd' else steps * 1e-2 * 1.01
            np.testing.assert_allclose(have, want, atol=noise)
        else:
            np.testing.assert_allclose(have, want, rtol=1e-4, atol=2e-7, err_msg=name)


def check_empty_slice(rank, world):
    """Every user of the step belongs to rank 0: after routing, the other ranks have no sample.
    They still serve rows, take gradients and join the collectives; the result is the oracle's."""
    U, N, D, B, K = 40, 30, 8, 24, 3
    t0 = full_tables(U, N, D, 5)
    model = ShardedMatrixFactorizationModel(U, N, K, embedding_dim=D, lr=1e-2, bias_lr=1e-2,
                                            optimizer='sparse_adam', bias_optimizer='sgd', loss='bpr',
                                            device=DEV, engine=engine(), init=False)
    model.load_full_state_dict({f'{k}.weight': dev(v.copy()) for k, v in t0.items()})
    ref = O.TrainState(t0, optimizer='sparse_adam', bias_optimizer='sgd', lr=1e-2, bias_lr=1e-2,
                       sparse=True)
    rng = np.random.default_rng(3)
    for step in range(2):
        users = rng.integers(0, U // world, B) * world          # owner 0
        pos = rng.integers(0, N, B)
        neg = rng.integers(0, N, (B, K))
        out = ref.step(users, pos, neg, model._loss_kind, num_items=N)
        batch = ((dev(users.astype(np.int32)), dev(pos.astype(np.int32))), dev(neg))
        piece = route_batch_to_user_owners(shard_batch(batch, rank, world))
        assert piece[0][0].shape[0] == (B if rank == 0 else 0)
        got = model.train_step(piece, users_local=True)
        np.testing.assert_allclose(got.item(), out['loss'], rtol=2e-5, atol=1e-7)
    sd = model.state_dict()
    for name in ('user_embeddings', 'item_embeddings', 'item_biases'):
        np.testing.assert_allclose(sd[f'{name}.weight'].cpu().numpy(), ref.t[name], rtol=1e-4,
                                   atol=2e-7, err_msg=name)


def check_init(rank, world):
    U, N, D = 37, 29, 16
    m = ShardedMatrixFactorizationModel(U, N, 3, embedding_dim=D, device=DEV, engine=engine(),
                                        seed=5)
    sd = m.state_dict()
    want_u = counter_normal(torch.arange(U, device=DEV), D, 5, 0, 1.0 / (2.5 * D))
    want_i = counter_normal(torch.arange(N, device=DEV), D, 5, 1, 1.0 / (2.5 * D))
    assert torch.equal(sd['user_embeddings.weight'], want_u)        # independent of G
    assert torch.equal(sd['item_embeddings.weight'], want_i)
    # ... and of the device up to the last bits of the float64 log / cos behind it
    cpu_u = counter_normal(torch.arange(U), D, 5, 0, 1.0 / (2.5 * D))
    np.testing.assert_allclose(want_u.cpu().numpy(), cpu_u.numpy(), rtol=1e-6, atol=1e-9)
    assert float(sd['user_biases.weight'].abs().max()) == 0.0
    # streamed gather to rank 0's host (tiny chunks: several rounds, ragged last chunk) and the
    # reference-format file written from it
    host = m.state_dict_to_rank0(chunk_rows=5)
    assert (host is None) == (rank != 0)
    if rank == 0:
        for k_, v in sd.items():
            assert torch.equal(host[k_], v.cpu()), k_
    import os
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f'collie_sharded_save_{os.getppid()}.pth')
    m.save_model(path, chunk_rows=7)
    dist.barrier()
    if rank == 0:
        loaded = torch.load(path, weights_only=False)
        assert set(loaded) == {'state_dict', 'hparams'}
        for k_, v in sd.items():
            assert torch.equal(loaded['state_dict'][k_], v.cpu()), k_
        assert loaded['hparams']['num_users'] == U and loaded['hparams']['embedding_dim'] == D
        os.remove(path)
    big = counter_normal(torch.arange(4096), 64, 0, 0, 1.0)
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01


def check_eval(rank, world):
    U, N, D, k = 61, 47, 8, 10
    t0 = full_tables(U, N, D, 3)
    t0['item_embeddings'][5] = t0['item_embeddings'][6]          # engineered score ties
    t0['item_biases'][5] = t0['item_biases'][6]
    t0['item_embeddings'][11] = t0['item_embeddings'][20]
    t0['item_biases'][11] = t0['item_biases'][20]
    model = ShardedMatrixFactorizationModel(U, N, 3, embedding_dim=D, device=DEV, engine=engine(),
                                            init=False)
    model.load_full_state_dict({f'{k_}.weight': dev(v.copy()) for k_, v in t0.items()})
    rng = np.random.default_rng(11)
    rows, cols = [], []
    for u in range(U):
        if u % 7 == 3:
            continue                                              # users without test items
        c = rng.choice(N, size=int(rng.integers(1, 13)), replace=False)
        rows += [u] * len(c)
        cols += c.tolist()
    mat = coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, N))
    inter = _Inter(mat)
    for bs in (20, 7):
        got = model.evaluate_in_batches([M.mapk, M.mrr, M.auc], inter, k=k, batch_size=bs,
                                        users_per_launch=16)
        want = O.evaluate_in_batches([O.mapk, O.mrr, O.auc], mat, t0, k=k, batch_size=bs)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
    one = model.evaluate_in_batches([M.mapk], inter, k=k)
    np.testing.assert_allclose(one, O.evaluate_in_batches([O.mapk], mat, t0, k=k), rtol=1e-12)
    # any k <= num_items (metrics.py:170-175): beyond the 16 candidates one scoring round keeps,
    # beyond a whole shard (47 items over 2 ranks = 24 + 23), and the whole catalogue
    for big_k in (20, 30, N):
        got = model.evaluate_in_batches([M.mapk, M.mrr, M.auc], inter, k=big_k, batch_size=20)
        want = O.evaluate_in_batches([O.mapk, O.mrr, O.auc], mat, t0, k=big_k, batch_size=20)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
        one = model.evaluate_in_batches([M.mapk], inter, k=big_k)
        np.testing.assert_allclose(one, O.evaluate_in_batches([O.mapk], mat, t0, k=big_k), rtol=1e-12)
    try:
        model.evaluate_in_batches([M.mapk], inter, k=N + 1)
    except ValueError:
        pass
    else:
        raise AssertionError('k > num_items must be refused')


def check_errors(rank, world):
    model = ShardedMatrixFactorizationModel(10, 10, 2, embedding_dim=4, device=DEV,
                                            engine=engine())
    users = torch.tensor([1, 2, 3, 4], dtype=torch.int32, device=DEV)
    pos = torch.tensor([1, 2, 3, 10 if rank == 1 else 4], dtype=torch.int32,
                       device=DEV)                                   # bad on rank 1 only
    neg = torch.zeros(4, 2, dtype=torch.int64, device=DEV)
    try:
        model.train_step(((users, pos), neg))
    except IndexError:
        pass                                   # every rank learns of it through the exchange
    else:
        raise AssertionError('out-of-range id was not reported')
    ok = torch.tensor([1, 2, 3, 4], dtype=torch.int32, device=DEV)
    try:                                       # users 1..4 cannot all belong to one of two ranks
        model.train_step(((users, ok), neg), users_local=True)
    except IndexError:
        pass
    else:
        raise AssertionError('a user owned by another rank was not reported')
    try:
        ShardedMatrixFactorizationModel(10, 10, 2, optimizer='adam', device=DEV, engine=engine())
    except ValueError:
        pass
    else:
        raise AssertionError('dense adam must be refused')


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    genre = dev(np.random.default_rng(2).integers(0, 4, 41))
    check_init(rank, world)
    check_training(rank, world, 'bpr', 'sparse_adam', 'sgd', 10)
    check_training(rank, world, 'hinge', 'adagrad', 'adagrad', 1)
    check_training(rank, world, 'warp', 'sgd', 'sgd', 5)
    check_training(rank, world, 'adaptive', 'sparse_adam', 'sparse_adam', 4, {'genre': genre})
    check_training(rank, world, 'bpr', 'sparse_adam', 'sgd', 10, route=True)
    check_training(rank, world, 'warp', 'adagrad', 'sparse_adam', 5, {'genre': genre}, route=True)
    check_empty_slice(rank, world)
    check_eval(rank, world)
    check_errors(rank, world)
    dist.barrier()
    dist.destroy_process_group()
    print(f'rank {rank}: sharded checks ok', flush=True)


if __name__ == '__main__':
    main()
