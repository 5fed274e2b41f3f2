"""
GPU tests of the device-side batch assembly + negative sampler (``collie_sample_batch``) through
the C ABI: bit-exact against oracle/sampler_oracle.py on seeded inputs, the reference's sampling
invariants at a larger size, and the loader / trainer built on it.
"""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def make_data(U, N, nnz, seed=0):
    rng = np.random.default_rng(seed)
    keys = rng.choice(U * N, nnz, replace=False)
    rows, cols = (keys // N).astype(np.int32), (keys % N).astype(np.int32)
    csr = sp.coo_matrix((np.ones(nnz), (rows, cols)), shape=(U, N)).tocsr()
    csr.sort_indices()
    return rows, cols, csr


def to_dev(dev, *arrays):
    return [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrays]


@pytest.mark.parametrize('exact,shuffle,max_tries,K', [(False, False, 0, 10), (True, False, 200, 5),
                                                       (True, True, 200, 7), (True, True, 2, 6),
                                                       (False, True, 0, 1), (True, True, 200, 20),
                                                       (True, False, 50, 70), (True, True, 0, 3)])
def test_device_sampler_is_bit_exact_vs_oracle(dev, exact, shuffle, max_tries, K):
    from oracle import sampler_oracle as S
    from collie_amd import ops
    # dense rows: rejections, second candidate windows (K = 20) and budget exhaustion happen;
    # K = 70 exceeds the lane-parallel path's list and takes the sequential rule
    U, N, nnz = (50, 40, 900) if K <= 20 else (50, 200, 4000)
    rows, cols, csr = make_data(U, N, nnz, seed=K)
    order = np.random.default_rng(1).permutation(nnz).astype(np.int64) if shuffle else None
    indptr = csr.indptr.astype(np.int64) if exact else None
    indices = csr.indices.astype(np.int32) if exact else None
    d_rows, d_cols, d_order, d_indptr, d_indices = to_dev(dev, rows, cols, order, indptr, indices)
    for start, batch, base in ((0, 300, 0), (300, 600, 0), (17, 101, 5 * nnz)):
        want = S.sample_batch(rows, cols, order, start, batch, indptr, indices, N, K, max_tries,
                              seed=(1 << 40) + 12345, position_base=base)
        (u, p), n = ops.sample_batch(d_rows, d_cols, d_order, start, batch, d_indptr, d_indices, N,
                                     K, max_tries, (1 << 40) + 12345, base)
        assert u.dtype == torch.int32 and p.dtype == torch.int32 and n.dtype == torch.int64
        np.testing.assert_array_equal(u.cpu().numpy(), want[0])
        np.testing.assert_array_equal(p.cpu().numpy(), want[1])
        np.testing.assert_array_equal(n.cpu().numpy(), want[2])


def test_device_sampler_equals_the_references_negative_sample(dev):
    """``collie_sample_batch`` against tests/golden/ref_sampler.npz: the reference's own
    ``Interactions.__getitem__`` / ``_negative_sample`` (datasets.py:378-445) reading the Philox
    stream (exact mode with and without an exhausted budget, K = 1 / 4 / 10, tiny catalogues,
    approximate mode scalar and batched), bit for bit, whole and in ragged pieces."""
    from collie_amd import ops
    from test_sampler_cpu import ref_sampler_cases
    for c in ref_sampler_cases():
        exact = c['max_tries'] > 0
        n = len(c['rows'])
        indptr = c['csr'].indptr.astype(np.int64) if exact else None
        indices = c['csr'].indices.astype(np.int32) if exact else None
        d_rows, d_cols, d_order, d_indptr, d_indices = to_dev(dev, c['rows'], c['cols'], c['order'],
                                                              indptr, indices)
        for start, batch in ((0, n), (0, n // 3), (n // 3, n - n // 3)):
            (u, p), neg = ops.sample_batch(d_rows, d_cols, d_order, start, batch, d_indptr,
                                           d_indices, c['N'], c['K'], c['max_tries'], c['seed'],
                                           c['base'])
            sl = slice(start, start + batch)
            np.testing.assert_array_equal(u.cpu().numpy(), c['want'][0][sl], err_msg=c['tag'])
            np.testing.assert_array_equal(p.cpu().numpy(), c['want'][1][sl], err_msg=c['tag'])
            np.testing.assert_array_equal(neg.cpu().numpy(), c['want'][2][sl], err_msg=c['tag'])


def test_exact_sampling_invariants_at_scale(dev):
    """tests/test_interactions.py:401-447 of the reference, on 200 k interactions."""
    from collie_amd import ops
    U, N, nnz, K = 5000, 2000, 200_000, 10
    rows, cols, csr = make_data(U, N, nnz, seed=3)
    d = to_dev(dev, rows, cols, csr.indptr.astype(np.int64), csr.indices.astype(np.int32))
    (u, p), neg = ops.sample_batch(d[0], d[1], None, 0, nnz, d[2], d[3], N, K, 200, 7, 0)
    neg = neg.cpu().numpy()
    pos_keys = np.sort(rows.astype(np.int64) * N + cols)
    keys = (rows.astype(np.int64)[:, None] * N + neg).ravel()
    hit = pos_keys[np.minimum(np.searchsorted(pos_keys, keys), nnz - 1)] == keys
    assert not hit.any()                                        # every negative is a true negative
    srt = np.sort(neg, axis=1)
    assert (np.diff(srt, axis=1) != 0).all()                    # distinct inside a sample
    counts = np.bincount(neg.ravel(), minlength=N)
    assert counts.min() > 0.8 * counts.mean() and counts.max() < 1.2 * counts.mean()


def test_device_loader_epoch_and_trainer(dev):
    from collie_amd.interactions import DeviceInteractionsDataLoader, Interactions
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.trainer import CollieMinimalTrainer
    U, N, nnz = 300, 200, 6000
    rows, cols, _ = make_data(U, N, nnz, seed=5)
    inter = Interactions(users=rows, items=cols, num_users=U, num_items=N, allow_missing_ids=True,
                         num_negative_samples=10, seed=11)
    loader = DeviceInteractionsDataLoader(inter, batch_size=1024, shuffle=True, device=dev)
    assert len(loader) == 6
    seen, negs = [], []
    for (u, p), n in loader:
        assert u.is_cuda and u.dtype == torch.int32 and n.dtype == torch.int64
        assert n.shape == (u.shape[0], 10)
        seen.append(u.cpu().numpy().astype(np.int64) * N + p.cpu().numpy())
        negs.append(n.cpu().numpy())
    # one epoch visits every interaction exactly once, in a shuffled order
    got = np.concatenate(seen)
    np.testing.assert_array_equal(np.sort(got), np.sort(rows.astype(np.int64) * N + cols))
    assert not np.array_equal(got, rows.astype(np.int64) * N + cols)
    second = np.concatenate([n.cpu().numpy() for _, n in loader])
    assert not np.array_equal(np.concatenate(negs), second)     # fresh negatives every epoch
    # the trainers consume it unchanged (no host round trip per batch)
    torch.manual_seed(0)
    model = MatrixFactorizationModel(train=loader, embedding_dim=16, lr=5e-2, sparse=True,
                                     optimizer='sparse_adam', loss='adaptive_bpr')
    trainer = CollieMinimalTrainer(model, max_epochs=1, gpus=1, verbosity=0, logger=None,
                                   early_stopping_patience=None)
    trainer.fit(model)
    with torch.no_grad():
        model.train()
        first = float(np.mean([model.calculate_loss(b).item() for b in loader]))
    trainer.max_epochs = 6
    trainer.fit(model)
    with torch.no_grad():
        model.train()
        last = float(np.mean([model.calculate_loss(b).item() for b in loader]))
    assert last < first
