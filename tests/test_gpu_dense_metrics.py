"""
GPU parity tests (``-m gpu``) of the metric boundary at real sizes: the ``mapk`` / ``mrr`` / ``auc``
callables on a materialised ``preds [users, items]`` matrix of ANY catalogue size and ANY ``k``
(reference: ``collie/metrics.py:144-282``, which take any N and k), the MFMA evaluator with
``k > 16`` (one more scoring pass per 16 entries), and ``evaluate_in_batches`` on
``HybridPretrainedModel`` (generic ``get_preds`` route) at N >= 50,000, k = 20.
Bars: top-k ids bit-exact against the oracle's (score desc, id asc) order, ties included; AP and RR
exact (integer ranks), AUC 1e-6.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _targets(rng, users, U, N, max_pos):
    nnz = rng.integers(0, max_pos + 1, len(users))
    nnz[0] = 0                                    # a user without positives
    nnz[-1] = max_pos
    rows = np.repeat(users, nnz)
    cols = np.concatenate([rng.choice(N, n, replace=False) for n in nnz] + [np.empty(0, np.int64)])
    return sp.coo_matrix((np.ones(len(rows)), (rows, cols.astype(np.int64))), shape=(U, N)).tocsr()


@pytest.mark.parametrize('n,N,ks', [(3, 4, (1, 4)), (7, 300, (1, 10, 17, 300)),
                                    (5, 5000, (16, 100, 1024, 1025, 3000, 5000)),
                                    (3, 70001, (10, 20, 2500)), (40, 1682, (10, 50))])
def test_preds_topk_any_k_equals_stable_sort(dev, n, N, ks):
    """``preds.topk(k)`` on a dense matrix: descending score, ascending id on ties, any k <= N."""
    from oracle import mf_oracle as O
    from collie_amd import ops
    rng = np.random.default_rng(n + N)
    # quantised scores: thousands of exact ties per row, both zeros, +-inf
    preds = (rng.integers(-40, 41, (n, N)) / 8).astype(np.float32)
    preds[0, ::7] = -0.0
    preds[0, 1::7] = 0.0
    preds[-1, N // 2] = np.inf
    preds[-1, N // 3] = -np.inf
    wide = torch.zeros(n, N + 13, device=dev)
    wide[:, :N] = torch.tensor(preds)
    view = wide[:, :N]                             # row stride != N: no copy inside the call
    for k in ks:
        scores, ids = ops.preds_topk(view, k)
        ref = O.topk_ids(preds, k)
        np.testing.assert_array_equal(ids.cpu().numpy(), ref)
        np.testing.assert_array_equal(scores.cpu().numpy(), np.take_along_axis(preds, ref, axis=1))
    from collie_amd import _lib
    with pytest.raises(_lib.CollieHipError):
        ops.preds_topk(view, N + 1)


@pytest.mark.parametrize('N,Bu,k,max_pos', [(60_000, 12, 20, 40), (300, 20, 10, 12),
                                            (1_200_000, 3, 50, 1500), (27_000, 64, 100, 30)])
def test_metric_callables_on_large_catalogues_vs_oracle(dev, N, Bu, k, max_pos):
    """mapk / mrr / auc(targets, user_ids, preds, k) at catalogue sizes the reference accepts
    (metrics.py:144-282), with exact score ties and sigmoid-saturated scores."""
    from oracle import mf_oracle as O
    from collie_amd.metrics import auc, mapk, mrr
    rng = np.random.default_rng(N + k)
    U = 500
    users = np.sort(rng.choice(U, Bu, replace=False))
    preds = rng.normal(0, 3, (Bu, N)).astype(np.float32)
    preds[:, ::5] = np.round(preds[:, ::5] * 4) / 4          # exact ties
    preds[:, 1::9] += 30                                     # sigmoid == 1.0 in fp32: ties there only
    csr = _targets(rng, users, U, N, max_pos)
    pt = torch.tensor(preds, device=dev)
    assert mapk(csr, users, pt, k) == pytest.approx(O.mapk(csr, users, preds, k), rel=1e-12)
    assert mrr(csr, users, pt) == pytest.approx(O.mrr(csr, users, preds), rel=1e-6)
    assert auc(csr, users, pt) == pytest.approx(
        float(O.auc_per_user(csr, users, preds).astype(np.float32).mean(dtype=np.float32)), rel=1e-6)
    with pytest.raises(ValueError):
        mapk(csr, users, pt, N + 1)


@pytest.mark.parametrize('D,N,U,k,stride', [(64, 5003, 300, 20, 1), (128, 70001, 200, 50, 1),
                                            (10, 1682, 943, 17, 1), (32, 4100, 100, 33, 3),
                                            (128, 3000, 64, 3000, 1)])
def test_mfma_topk_with_k_above_16_is_exact(dev, D, N, U, k, stride):
    """``collie_mf_eval_topk`` / ``..._topk_rank_counts`` with k > 16 (rounds of 16 entries, each
    resuming strictly after the previous round's last entry): ids equal the oracle's order bit for
    bit on exactly representable scores, ties across round boundaries included."""
    from oracle import mf_oracle as O
    from collie_amd import _lib, ops
    rng = np.random.default_rng(D + N + k)
    tables = {'user_embeddings': (rng.integers(-2, 3, (U, D)) / 4).astype(np.float32),
              'item_embeddings': (rng.integers(-2, 3, (N, D)) / 4).astype(np.float32),
              'user_biases': (rng.integers(-8, 9, (U, 1)) / 16).astype(np.float32),
              'item_biases': (rng.integers(-8, 9, (N, 1)) / 16).astype(np.float32)}
    td = {n_: torch.tensor(v, device=dev) for n_, v in tables.items()}
    t = _lib.make_tables(td['user_embeddings'], td['item_embeddings'], td['user_biases'],
                         td['item_biases'])
    users = np.sort(rng.choice(U, min(U, 150), replace=False))
    slots = torch.tensor(users, dtype=torch.int64, device=dev)
    preds = O.get_preds(tables, users, N)
    ref = O.topk_ids(preds, k)
    ws = ops.EvalWorkspace(dev)
    scores, ids = ops.eval_topk(t, slots, 0, N, k, None, ws, id_stride=stride, id_offset=stride - 1)
    np.testing.assert_array_equal(ids.cpu().numpy(), ref * stride + (stride - 1))
    np.testing.assert_array_equal(scores.cpu().numpy(), np.take_along_axis(preds, ref, axis=1))
    # the fused pass (rank counts + round 0 of the top-k in one launch) gives the same ids
    csr = _targets(rng, users, U, N * stride, 9)
    from collie_amd.metrics import _csr_slice
    indptr, indices = _csr_slice(csr, users, dev)
    pos, lo, hi, best_s, best_id = ops.eval_pos_scores(t, slots, indptr, indices, None, True,
                                                       id_stride=stride, id_offset=stride - 1)
    counts = torch.zeros(len(users), 3, dtype=torch.int64, device=dev)
    _, ids2 = ops.eval_topk_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, 0, N, k, None,
                                        counts, ws, id_stride=stride, id_offset=stride - 1)
    assert torch.equal(ids, ids2)
    counts1 = torch.zeros_like(counts)
    ops.eval_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, 0, N, counts1,
                         id_stride=stride, id_offset=stride - 1)
    assert torch.equal(counts, counts1)


def test_evaluate_in_batches_mf_fast_route_k20_vs_oracle(dev):
    from oracle import mf_oracle as O
    from collie_amd.interactions import Interactions
    from collie_amd.metrics import auc, evaluate_in_batches, mapk, mrr
    from collie_amd.model import MatrixFactorizationModel
    rng = np.random.default_rng(3)
    U, N, D = 400, 9000, 64
    tables = {'user_embeddings': (rng.integers(-2, 3, (U, D)) / 4).astype(np.float32),
              'item_embeddings': (rng.integers(-2, 3, (N, D)) / 4).astype(np.float32),
              'user_biases': (rng.integers(-8, 9, (U, 1)) / 16).astype(np.float32),
              'item_biases': (rng.integers(-8, 9, (N, 1)) / 16).astype(np.float32)}

    class Shape:
        num_users, num_items, num_negative_samples = U, N, 10

    model = MatrixFactorizationModel(train=Shape(), embedding_dim=D, lr_scheduler_func=None)
    with torch.no_grad():
        for n_, v in tables.items():
            getattr(model, n_).weight.copy_(torch.tensor(v))
    model.to(dev).eval()
    rows, cols = rng.integers(0, U, 3000), rng.integers(0, N, 3000)
    test = Interactions(users=rows, items=cols, num_users=U, num_items=N, allow_missing_ids=True,
                        check_num_negative_samples_is_valid=False, num_negative_samples=1,
                        max_number_of_samples_to_consider=0, seed=0)
    got = evaluate_in_batches([mapk, mrr, auc], test, model, k=20, batch_size=20, verbose=False)
    want = O.evaluate_in_batches([O.mapk, O.mrr, O.auc], test.mat, tables, k=20, batch_size=20)
    np.testing.assert_allclose(got, want, rtol=1e-6)


def test_evaluate_in_batches_hybrid_k20_at_50k_items_vs_oracle(dev):
    """``evaluate_in_batches([mapk, mrr, auc], ..., HybridPretrainedModel, k=20)`` on a 50,000-item
    catalogue: the generic route (``get_preds`` through the head's forward, then the dense metric
    callables), against the oracle's metrics on the same score matrix."""
    from oracle import mf_oracle as O
    from collie_amd.hybrid import HybridPretrainedModel
    from collie_amd.interactions import Interactions
    from collie_amd.metrics import auc, evaluate_in_batches, get_preds, mapk, mrr
    from collie_amd.model import MatrixFactorizationModel
    U, N, D, K = 300, 50_000, 16, 3
    rng = np.random.default_rng(9)

    class Shape:
        num_users, num_items, num_negative_samples = U, N, K

    torch.manual_seed(1)
    mf = MatrixFactorizationModel(train=Shape(), embedding_dim=D, lr_scheduler_func=None)
    with torch.no_grad():
        mf.user_embeddings.weight.mul_(8.0)
        mf.item_embeddings.weight.mul_(8.0)
        mf.item_biases.weight.normal_(0, 0.1)
    item_md = rng.normal(size=(N, 12)).astype(np.float32)
    model = HybridPretrainedModel(train=Shape(), trained_model=mf, item_metadata=item_md,
                                  item_metadata_layers_dims=[8], combined_layers_dims=[16, 8],
                                  lr_scheduler_func=None).to(dev).eval()
    test_users = np.sort(rng.choice(U, 45, replace=False))
    nnz = rng.integers(1, 30, len(test_users))
    rows = np.repeat(test_users, nnz)
    cols = np.concatenate([rng.choice(N, n, replace=False) for n in nnz])
    test = Interactions(users=rows, items=cols, num_users=U, num_items=N, allow_missing_ids=True,
                        check_num_negative_samples_is_valid=False, num_negative_samples=1,
                        max_number_of_samples_to_consider=0, seed=0)
    got = evaluate_in_batches([mapk, mrr, auc], test, model, k=20, batch_size=20, verbose=False)
    targets = test.mat.tocsr()
    acc = [0.0, 0.0, 0.0]
    for i in range(int(np.ceil(len(test_users) / 20))):
        ur = test_users[i * 20:(i + 1) * 20]
        scores = get_preds(model, ur, N, dev).cpu().numpy()
        for j, fn in enumerate((O.mapk, O.mrr, O.auc)):
            acc[j] += fn(targets, ur, scores, 20) * len(ur)
    want = [a / len(test_users) for a in acc]
    np.testing.assert_allclose(got, want, rtol=1e-6)
    assert got[0] >= 0 and 0 < got[1] <= 1 and 0 < got[2] < 1
