"""
GPU parity tests for the MFMA evaluator (``-m gpu``): top-k ids bit-exact (ties by index),
MAP@k / MRR / AUC within 1e-5 relative of the oracle, the reference's own known answers reproduced
through ``evaluate_in_batches``.
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
NAMES = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def make_model(tables, dev, **kw):
    from collie_amd.model import MatrixFactorizationModel

    class Shape:
        num_users, num_items = tables['user_embeddings'].shape[0], tables['item_embeddings'].shape[0]
        num_negative_samples = 10

    model = MatrixFactorizationModel(train=Shape(), embedding_dim=tables['user_embeddings'].shape[1],
                                     lr_scheduler_func=None, **kw)
    with torch.no_grad():
        for n in NAMES:
            getattr(model, n).weight.copy_(torch.tensor(tables[n]))
    return model.to(dev).eval()


def test_reference_known_answers_through_evaluate_in_batches(golden_dir, dev):
    """tests/test_metrics.py:230-256 of the reference: the 3x4 score matrix, k=4, batches 20/2/1.
    The fixed score matrix is realised exactly as U = scores, I = identity, zero biases."""
    from collie_amd.interactions import Interactions
    from collie_amd.metrics import auc, evaluate_in_batches, mapk, mrr
    with open(os.path.join(golden_dir, 'ref_known_answers.json')) as f:
        M = json.load(f)['metrics']
    P = np.array(M['predicted_scores'], np.float32)
    tables = {'user_embeddings': P, 'item_embeddings': np.eye(4, dtype=np.float32),
              'user_biases': np.zeros((3, 1), np.float32), 'item_biases': np.zeros((4, 1), np.float32)}
    model = make_model(tables, dev)
    test = Interactions(users=M['users'], items=M['items'], ratings=[1] * 8,
                        check_num_negative_samples_is_valid=False)
    exp = M['evaluate_in_batches_k4']
    for batch_size in (20, 2, 1):
        m, r, a = evaluate_in_batches([mapk, mrr, auc], test, model, k=4, batch_size=batch_size,
                                      verbose=False)
        np.testing.assert_almost_equal(m, exp['mapk'])
        np.testing.assert_almost_equal(r, exp['mrr'])
        np.testing.assert_almost_equal(a, exp['auc'])
    # single metric -> bare float (metrics.py:395)
    assert isinstance(evaluate_in_batches([mapk], test, model, k=4, verbose=False), float)
    # k > n_items raises ValueError (metrics.py:170-175)
    with pytest.raises(ValueError):
        evaluate_in_batches([mapk], test, model, k=5, verbose=False)
    with pytest.raises(ValueError):
        evaluate_in_batches([mapk], 'not interactions', model)


def test_standalone_metric_callables_known_answers(golden_dir, dev):
    """tests/test_metrics.py:101-154: mapk == .75 (k=4, users 1,2), MAP@1 = 2/3, MRR, AUC."""
    from collie_amd.metrics import auc, mapk, mrr
    with open(os.path.join(golden_dir, 'ref_known_answers.json')) as f:
        M = json.load(f)['metrics']
    P = torch.tensor(M['predicted_scores'], dtype=torch.float32, device=dev)
    tg = sp.coo_matrix((np.ones(8), (M['users'], M['items'])), shape=(3, 4)).tocsr()
    assert mapk(tg, np.array([1, 2]), P[[1, 2]], k=4) == M['mapk_k4_users_1_2']
    np.testing.assert_almost_equal(mapk(tg, np.arange(3), P, k=1), M['mapk_k1_all'])
    np.testing.assert_almost_equal(mrr(tg, np.arange(3), P), M['mrr_all'])
    np.testing.assert_almost_equal(auc(tg, np.arange(3), P), np.mean(M['auc_sklearn_per_user']))
    with pytest.raises(ValueError):
        mapk(tg, np.arange(3), P, k=5)


def test_eval_case_with_engineered_ties(golden_dir, dev):
    from oracle import mf_oracle as O
    from collie_amd import _lib, ops
    z = np.load(os.path.join(golden_dir, 'eval_case.npz'))
    tables = {n: z[f't|{n}'] for n in NAMES}
    model = make_model(tables, dev)
    t = model._tables()
    users = z['test_users']
    N = tables['item_embeddings'].shape[0]
    coo = sp.coo_matrix((np.ones(len(z['rows'])), (z['rows'], z['cols'])), shape=(9, N))
    csr = coo.tocsr()
    slots = torch.tensor(users, dtype=torch.int64, device=dev)
    ws = ops.EvalWorkspace(dev)
    scores, ids = ops.eval_topk(t, slots, 0, N, 10, None, ws)
    # ids: exact against torch's own ``preds.topk(10)`` of the REFERENCE'S ``get_preds``
    # (ref_pipeline.npz, from collie/metrics.py run in the build container) wherever the leading
    # scores hold neither an exact tie (torch's tie order is implementation-defined) nor a
    # near-tie inside the rounding difference of the two summation orders
    r = np.load(os.path.join(golden_dir, 'ref_pipeline.npz'))
    np.testing.assert_array_equal(r['eval|users'], users)
    ref_preds = r['eval|preds_reference']
    ref_ids = r['eval|top10_ids_reference']
    got = ids.cpu().numpy()
    checked = 0
    for i in range(len(users)):
        lead = np.sort(ref_preds[i])[::-1][:12]
        if np.all(-np.diff(lead) > 1e-6):
            np.testing.assert_array_equal(got[i], ref_ids[i])
            checked += 1
    assert checked >= 3
    stable = O.topk_ids(ref_preds, 10)
    np.testing.assert_allclose(scores.cpu().numpy(), np.take_along_axis(ref_preds, stable, axis=1),
                               rtol=1e-5, atol=1e-6)
    # tie groups come out in ascending id order
    s = scores.cpu().numpy()
    for i in range(len(users)):
        same = np.nonzero(np.diff(s[i]) == 0)[0]
        assert np.all(got[i][same] < got[i][same + 1])
    # per-user metrics against the reference's own mapk / mrr / auc, one user at a time
    from collie_amd.metrics import _csr_slice, _per_user
    from tests.test_oracle_golden import rr_tie_interval
    indptr, indices = _csr_slice(csr, users, dev)
    ap, rr, au = _per_user(t, slots, indptr, indices, N, 10, None, (True, True, True), ws)
    np.testing.assert_allclose(ap.cpu().numpy(), r['eval|ap_reference'], rtol=1e-5)
    np.testing.assert_allclose(au.cpu().numpy(), r['eval|auc_reference'], rtol=1e-5)
    np.testing.assert_allclose(au.cpu().numpy(), z['auc_sklearn_on_torch_sigmoid'], rtol=1e-5)
    rr = rr.cpu().numpy()
    for i, u in enumerate(users):
        lo, hi = rr_tie_interval(ref_preds[i], csr[u].indices)
        if lo == hi:
            np.testing.assert_allclose(rr[i], r['eval|rr_reference'][i], rtol=1e-5)
        else:   # an exact tie with the best positive: the build ranks tied items by ascending id
            assert lo - 1e-9 <= rr[i] <= hi + 1e-9
            np.testing.assert_allclose(rr[i], O.mrr_per_user(csr, users[i:i + 1],
                                                             ref_preds[i:i + 1])[0], rtol=1e-6)


@pytest.mark.parametrize('D,N,U,k', [(64, 5003, 300, 10), (10, 1682, 943, 10), (128, 2048, 70, 16),
                                     (30, 777, 129, 1), (256, 300, 40, 5), (8, 50000, 33, 10),
                                     (32, 4100, 700, 7), (128, 70001, 520, 10)])
def test_topk_ids_exact_on_exactly_representable_scores(dev, D, N, U, k):
    """Embeddings on a coarse grid: every dot product is exact in fp32 in any order, so the ids
    must equal the oracle's (score desc, id asc) order bit for bit, ties included."""
    from oracle import mf_oracle as O
    from collie_amd import ops
    rng = np.random.default_rng(D + N)
    tables = {'user_embeddings': (rng.integers(-4, 5, (U, D)) / 8).astype(np.float32),
              'item_embeddings': (rng.integers(-4, 5, (N, D)) / 8).astype(np.float32),
              'user_biases': (rng.integers(-8, 9, (U, 1)) / 16).astype(np.float32),
              'item_biases': (rng.integers(-8, 9, (N, 1)) / 16).astype(np.float32)}
    model = make_model(tables, dev)
    t = model._tables()
    users = np.sort(rng.choice(U, min(U, 600), replace=False))
    preds = O.get_preds(tables, users, N)
    ws = ops.EvalWorkspace(dev)
    slots = torch.tensor(users, dtype=torch.int64, device=dev)
    scores, ids = ops.eval_topk(t, slots, 0, N, k, None, ws)
    ref = O.topk_ids(preds, k)
    np.testing.assert_array_equal(ids.cpu().numpy(), ref)
    np.testing.assert_array_equal(scores.cpu().numpy(), np.take_along_axis(preds, ref, axis=1))
    # sharded: two item shards + merge give the same ids (catalog-sharded evaluation)
    cap = ops.eval_topk_capacity()
    half = N // 2
    parts = []
    for lo, hi in ((0, half), (half, N)):
        s_, i_ = ops.eval_topk(t, slots, lo, hi, min(k, hi - lo), None, ws)
        cs = torch.full((len(users), 1, cap), float('-inf'), device=dev)
        ci = torch.full((len(users), 1, cap), 2 ** 31 - 1, dtype=torch.int32, device=dev)
        cs[:, 0, :s_.shape[1]] = s_
        ci[:, 0, :i_.shape[1]] = i_.to(torch.int32)
        parts.append((cs, ci))
    ms, mi = ops.eval_merge_topk(torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1), k)
    np.testing.assert_array_equal(mi.cpu().numpy(), ref)
    # metrics against the oracle
    nnz_per_user = rng.integers(1, 12, len(users))
    nnz_per_user[:6] = (14, 15, 16, 17, 31, 40)   # around and beyond the 15 positives of the LDS tree
    rows = np.repeat(users, nnz_per_user)
    cols = np.concatenate([rng.choice(N, n, replace=False) for n in nnz_per_user])
    csr = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, N)).tocsr()
    from collie_amd.metrics import _csr_slice, _per_user
    indptr, indices = _csr_slice(csr, users, dev)
    ap, rr, au = _per_user(t, slots, indptr, indices, N, k, None, (True, True, True), ws)
    np.testing.assert_allclose(ap.cpu().numpy(), O.mapk_per_user(csr, users, preds, k)[0], rtol=1e-12)
    np.testing.assert_array_equal(rr.cpu().numpy(), O.mrr_per_user(csr, users, preds))
    np.testing.assert_allclose(au.cpu().numpy(), O.auc_per_user(csr, users, preds), rtol=1e-6)


@pytest.mark.parametrize('D,N,U,k,stride,y_range', [
    (128, 70001, 600, 10, 1, None), (64, 5003, 300, 16, 1, None), (32, 4100, 700, 7, 1, (0.0, 5.0)),
    (128, 3000, 130, 10, 3, None),           # a catalogue shard: global id = 3 * row + 1
    (10, 1682, 943, 10, 1, None),            # not an LDS-DMA shape: two passes inside the call
])
def test_fused_topk_and_rank_counts_equal_the_two_passes(dev, D, N, U, k, stride, y_range):
    """collie_mf_eval_topk_rank_counts against collie_mf_eval_topk + collie_mf_eval_rank_counts:
    ids, scores and the three rank counters bit for bit (ragged last tile, users with 0 ... 40
    positives, i.e. beyond the 15 positives of the per-user LDS search tree)."""
    from oracle import mf_oracle as O
    from collie_amd import ops
    from collie_amd.metrics import _csr_slice
    rng = np.random.default_rng(D * 7 + N)
    tables = O.init_tables(U, N, D, rng)
    tables['item_biases'] = rng.normal(0, 0.01, (N, 1)).astype(np.float32)
    tables['item_embeddings'] *= 20.0
    model = make_model(tables, dev)
    t = model._tables()
    users = np.sort(rng.choice(U, min(U, 520), replace=False))
    n = len(users)
    slots = torch.tensor(users, dtype=torch.int64, device=dev)
    off = 1 if stride > 1 else 0
    n_global = N * stride + off
    nnz_per_user = rng.integers(0, 41, n)
    nnz_per_user[:3] = (0, 1, 40)
    rows = np.repeat(users, nnz_per_user)
    cols = np.concatenate([rng.choice(N, c, replace=False) * stride + off for c in nnz_per_user])
    csr = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, n_global)).tocsr()
    indptr, indices = _csr_slice(csr, users, dev)
    ws = ops.EvalWorkspace(dev)
    pos, lo, hi, best_s, best_id = ops.eval_pos_scores(t, slots, indptr, indices, y_range, True,
                                                       id_stride=stride, id_offset=off)
    s_ref, i_ref = ops.eval_topk(t, slots, 0, N, k, y_range, ws, id_stride=stride, id_offset=off)
    c_ref = torch.zeros(n, 3, dtype=torch.int64, device=dev)
    ops.eval_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, 0, N, c_ref, id_stride=stride,
                         id_offset=off)
    c_got = torch.zeros(n, 3, dtype=torch.int64, device=dev)
    s_got, i_got = ops.eval_topk_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, 0, N, k,
                                             y_range, c_got, ws, id_stride=stride, id_offset=off)
    assert torch.equal(i_got, i_ref)
    assert torch.equal(s_got, s_ref)
    assert torch.equal(c_got, c_ref)
    assert int(c_ref.sum()) > 0
    # a sub-range of the items (what one rank of a contiguous catalogue split would run)
    a, b = N // 3, N - 5
    s_ref, i_ref = ops.eval_topk(t, slots, a, b, k, y_range, ws, id_stride=stride, id_offset=off)
    c_ref.zero_()
    ops.eval_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, a, b, c_ref, id_stride=stride,
                         id_offset=off)
    c_got.zero_()
    s_got, i_got = ops.eval_topk_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, a, b, k,
                                             y_range, c_got, ws, id_stride=stride, id_offset=off)
    assert torch.equal(i_got, i_ref) and torch.equal(s_got, s_ref) and torch.equal(c_got, c_ref)


def test_positive_scores_are_bit_identical_to_gemm_scores(dev):
    """The fmaf-chain restatement of the MFMA order: each user's top-1 item scored as a
    'positive' must reproduce the GEMM's score bit for bit (random real-valued embeddings)."""
    from oracle import mf_oracle as O
    from collie_amd import ops
    rng = np.random.default_rng(5)
    for D in (10, 64, 128):
        U, N = 500, 3000
        tables = O.init_tables(U, N, D, rng)
        tables['item_biases'] = rng.normal(0, 0.01, (N, 1)).astype(np.float32)
        model = make_model(tables, dev)
        t = model._tables()
        slots = torch.arange(U, dtype=torch.int64, device=dev)
        ws = ops.EvalWorkspace(dev)
        scores, ids = ops.eval_topk(t, slots, 0, N, 3, None, ws)
        indptr = torch.arange(0, 3 * U + 1, 3, dtype=torch.int64, device=dev)
        indices = ids.sort(dim=1).values.reshape(-1).contiguous()
        pos, _, _, _, _ = ops.eval_pos_scores(t, slots, indptr, indices, None, False)
        torch.cuda.synchronize()
        want = torch.gather(scores, 1, ids.argsort(dim=1)).reshape(-1)
        assert torch.equal(pos[:3 * U], want)


def test_evaluate_in_batches_vs_oracle_ml100k_shape(dev):
    from oracle import mf_oracle as O
    from collie_amd.interactions import Interactions
    from collie_amd.metrics import auc, evaluate_in_batches, mapk, mrr
    rng = np.random.default_rng(1)
    U, N, D = 943, 1682, 10
    tables = O.init_tables(U, N, D, rng)
    tables['user_embeddings'] *= 25
    tables['item_embeddings'] *= 25
    tables['item_biases'] = rng.normal(0, 0.3, (N, 1)).astype(np.float32)
    model = make_model(tables, dev)
    rows = rng.integers(0, U, 6000)
    cols = rng.integers(0, N, 6000)
    test = Interactions(users=rows, items=cols, num_users=U, num_items=N, allow_missing_ids=True,
                        check_num_negative_samples_is_valid=False)
    got = evaluate_in_batches([mapk, mrr, auc], test, model, k=10, batch_size=64, verbose=False)
    ref = O.evaluate_in_batches([O.mapk, O.mrr, O.auc], test.mat, tables, k=10, batch_size=64)
    np.testing.assert_allclose(got, ref, rtol=1e-5)
    # every subset of the metrics takes its own route through the evaluator (MAP alone: scoring
    # launch before the CSR rows; MRR / AUC alone: rank counts only; together: the fused pass) and
    # must return the same numbers
    assert evaluate_in_batches([mapk], test, model, k=10, batch_size=64, verbose=False) == got[0]
    assert evaluate_in_batches([mrr], test, model, k=10, batch_size=64, verbose=False) == got[1]
    assert evaluate_in_batches([auc], test, model, k=10, batch_size=64, verbose=False) == got[2]
    assert evaluate_in_batches([auc, mapk], test, model, k=10, batch_size=64,
                               verbose=False) == [got[2], got[0]]


def test_evaluate_in_batches_metric_subsets_agree_on_dma_shape(dev):
    """The same on a shape the LDS-DMA kernel serves (D = 64), where the three-metric call is the
    fused MODE 2 pass and MAP alone is MODE 0."""
    from oracle import mf_oracle as O
    from collie_amd.interactions import Interactions
    from collie_amd.metrics import auc, evaluate_in_batches, mapk, mrr
    rng = np.random.default_rng(2)
    U, N, D = 700, 5003, 64
    tables = O.init_tables(U, N, D, rng)
    tables['user_embeddings'] *= 25
    tables['item_embeddings'] *= 25
    tables['item_biases'] = rng.normal(0, 0.3, (N, 1)).astype(np.float32)
    model = make_model(tables, dev)
    test = Interactions(users=rng.integers(0, U, 9000), items=rng.integers(0, N, 9000),
                        num_users=U, num_items=N, allow_missing_ids=True,
                        check_num_negative_samples_is_valid=False)
    got = evaluate_in_batches([mapk, mrr, auc], test, model, k=10, batch_size=20, verbose=False)
    ref = O.evaluate_in_batches([O.mapk, O.mrr, O.auc], test.mat, tables, k=10, batch_size=20)
    np.testing.assert_allclose(got, ref, rtol=1e-5)
    assert evaluate_in_batches([mapk], test, model, k=10, verbose=False) == got[0]
    assert evaluate_in_batches([mrr, auc], test, model, k=10, verbose=False) == got[1:]


# ---- the opt-in bf16 x 3 scoring pass (collie_mf_eval_topk_bf16x3) ----------------------------------
@pytest.mark.parametrize('D,N,U,k', [(64, 5003, 300, 10), (128, 2048, 70, 12), (128, 70001, 520, 10),
                                     (64, 777, 129, 1)])
def test_bf16x3_topk_exact_on_exactly_representable_scores(dev, D, N, U, k):
    """Grid-valued embeddings are bf16 numbers: the split is exact (mid = lo = 0), every product is
    exact, so the pass must give the oracle's (score desc, id asc) ids and scores, ties included."""
    from oracle import mf_oracle as O
    from collie_amd import ops
    rng = np.random.default_rng(D + N)
    tables = {'user_embeddings': (rng.integers(-4, 5, (U, D)) / 8).astype(np.float32),
              'item_embeddings': (rng.integers(-4, 5, (N, D)) / 8).astype(np.float32),
              'user_biases': (rng.integers(-8, 9, (U, 1)) / 16).astype(np.float32),
              'item_biases': (rng.integers(-8, 9, (N, 1)) / 16).astype(np.float32)}
    model = make_model(tables, dev)
    t = model._tables()
    users = np.sort(rng.choice(U, min(U, 600), replace=False))
    preds = O.get_preds(tables, users, N)
    ws = ops.EvalWorkspace(dev)
    slots = torch.tensor(users, dtype=torch.int64, device=dev)
    split = ops.eval_split_items(model.item_embeddings.weight.data)
    scores, ids = ops.eval_topk_bf16x3(t, split, slots, 0, N, k, None, ws)
    ref = O.topk_ids(preds, k)
    np.testing.assert_array_equal(ids.cpu().numpy(), ref)
    np.testing.assert_array_equal(scores.cpu().numpy(), np.take_along_axis(preds, ref, axis=1))


@pytest.mark.parametrize('D,N,U,k,stride,y_range', [(128, 70001, 1500, 10, 1, None),
                                                    (64, 30011, 700, 12, 1, (0.0, 5.0)),
                                                    (128, 9000, 260, 5, 3, None),
                                                    (64, 129, 40, 10, 1, None)])
def test_bf16x3_topk_equals_the_f32_pass_on_random_tables(dev, D, N, U, k, stride, y_range):
    """Random tables (every mantissa bit in use): ids and scores of the bf16 x 3 pass equal the
    exact-f32 pass -- the scores bit for bit (they come from the same chain), the ids because 16
    candidates per user absorb the ~1e-7 ranking error of the split products."""
    from collie_amd import ops
    rng = np.random.default_rng(N)
    tables = {'user_embeddings': rng.normal(0, 0.3, (U, D)).astype(np.float32),
              'item_embeddings': rng.normal(0, 0.3, (N, D)).astype(np.float32),
              'user_biases': rng.normal(0, 0.05, (U, 1)).astype(np.float32),
              'item_biases': rng.normal(0, 0.05, (N, 1)).astype(np.float32)}
    model = make_model(tables, dev)
    t = model._tables()
    ws, ws2 = ops.EvalWorkspace(dev), ops.EvalWorkspace(dev)
    slots = torch.arange(U, dtype=torch.int64, device=dev)
    split = ops.eval_split_items(model.item_embeddings.weight.data)
    # the split really is a 24-bit representation
    parts = split.float().cpu().numpy()
    back = (parts[:, 0].astype(np.float64) + parts[:, 1] + parts[:, 2])
    np.testing.assert_allclose(back, tables['item_embeddings'], rtol=2 ** -22, atol=1e-30)
    want_s, want_i = ops.eval_topk(t, slots, 0, N, k, y_range, ws, stride, 1 if stride > 1 else 0)
    got_s, got_i = ops.eval_topk_bf16x3(t, split, slots, 0, N, k, y_range, ws2, stride,
                                        1 if stride > 1 else 0)
    assert torch.equal(got_i, want_i)
    assert torch.equal(got_s, want_s)
    # an item sub-range (a shard of the catalogue)
    lo, hi = N // 3, N - 7
    want_s, want_i = ops.eval_topk(t, slots, lo, hi, k, y_range, ws)
    got_s, got_i = ops.eval_topk_bf16x3(t, split, slots, lo, hi, k, y_range, ws2)
    assert torch.equal(got_i, want_i) and torch.equal(got_s, want_s)


def test_bf16x3_limits_and_evaluate_in_batches_opt_in(dev):
    from collie_amd import _lib, metrics as M, ops
    from collie_amd.interactions import Interactions
    assert ops.eval_bf16x3_ok(128, 10) and ops.eval_bf16x3_ok(64, 12)
    assert not ops.eval_bf16x3_ok(32, 10) and not ops.eval_bf16x3_ok(128, 13)
    rng = np.random.default_rng(5)
    U, N, D = 400, 3000, 64
    tables = {'user_embeddings': rng.normal(0, 0.3, (U, D)).astype(np.float32),
              'item_embeddings': rng.normal(0, 0.3, (N, D)).astype(np.float32),
              'user_biases': rng.normal(0, 0.05, (U, 1)).astype(np.float32),
              'item_biases': rng.normal(0, 0.05, (N, 1)).astype(np.float32)}
    model = make_model(tables, dev)
    ws = ops.EvalWorkspace(dev)
    slots = torch.arange(U, dtype=torch.int64, device=dev)
    split = ops.eval_split_items(model.item_embeddings.weight.data)
    with pytest.raises(_lib.CollieHipError):
        ops.eval_topk_bf16x3(model._tables(), split, slots, 0, N, 13, None, ws)
    rows = np.repeat(np.arange(U), 5)
    cols = rng.integers(0, N, rows.shape[0])
    test = Interactions(users=rows, items=cols, num_users=U, num_items=N, allow_missing_ids=True,
                        num_negative_samples=1, max_number_of_samples_to_consider=0, seed=0)
    want = M.evaluate_in_batches([M.mapk], test, model, k=10, verbose=False)
    model.eval_bf16x3 = True
    got = M.evaluate_in_batches([M.mapk], test, model, k=10, verbose=False)
    assert got == want
    both = M.evaluate_in_batches([M.mapk, M.mrr], test, model, k=10, verbose=False)   # f32 route
    assert both[0] == want
