"""
CPU tests: the numpy oracle (oracle/mf_oracle.py) against the golden vectors.

Golden vectors come from (a) the reference's own known-answer tests and (b) the reference's
``collie.loss`` functions / embedding layers + torch run in the build container
(tests/golden/make_golden.py).  Tolerance: 1e-5 relative (fp32), the bar BASELINE.json states.
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mf_oracle as O

RTOL = 1e-5


def close(a, b, rtol=RTOL, atol_scale=None):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.abs(b).max() if atol_scale is None else atol_scale
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(scale, 1e-30))


@pytest.fixture(scope='module')
def known(golden_dir):
    with open(os.path.join(golden_dir, 'ref_known_answers.json')) as f:
        return json.load(f)


# ------------------------------------------------------------------ reference known answers
def _loss_fixture(known):
    L = known['loss']
    S = np.array(L['scores'], np.float32)
    pi = np.array(L['positive_items'])
    ni = np.array(L['negative_items'])
    mni = np.array(L['many_negative_items'])
    meta = {'a': np.array(L['metadata_a']), 'b': np.array(L['metadata_b'])}
    return L, S, pi, ni, mni, meta


def test_ideal_difference_known_answers(known):
    L, S, pi, ni, mni, meta = _loss_fixture(known)
    d = O.ideal_difference(pi, ni, {'a': meta['a']}, {'a': .2})
    np.testing.assert_array_equal(d, np.array(L['metadata_a_diff'], np.float32))
    d2 = O.ideal_difference(np.tile(pi, (4, 1)), mni, meta, {'a': .2, 'b': .3})
    np.testing.assert_array_equal(d2, np.array(L['metadata_a_and_2_diff'], np.float32))
    with pytest.raises(ValueError, match='sum of metadata weights was 1.1, must be <=1'):
        O.ideal_difference(pi, ni, meta, {'a': .2, 'b': .9})


def test_losses_known_answers(known):
    L, S, pi, ni, mni, meta = _loss_fixture(known)
    ps, ns, mns = S[pi], S[ni], S[mni]
    w2 = {'a': .2, 'b': .3}
    got = {
        'bpr': O.loss_and_score_grads('bpr', ps, ns)[0],
        'hinge': O.loss_and_score_grads('hinge', ps, ns)[0],
        'adaptive_bpr': O.loss_and_score_grads('adaptive_bpr', ps, mns)[0],
        'adaptive_hinge': O.loss_and_score_grads('adaptive_hinge', ps, mns)[0],
        'warp_num_items_4': O.loss_and_score_grads('warp', ps, mns, num_items=4)[0],
        'bpr_meta_a_0.2': O.loss_and_score_grads('bpr', ps, ns, None, pi, ni, {'a': meta['a']}, {'a': .2})[0],
        'hinge_meta_a_0.2': O.loss_and_score_grads('hinge', ps, ns, None, pi, ni, {'a': meta['a']}, {'a': .2})[0],
        'adaptive_bpr_meta_a0.2_b0.3': O.loss_and_score_grads('adaptive_bpr', ps, mns, None, pi, mni, meta, w2)[0],
        'adaptive_hinge_meta_a0.2_b0.3': O.loss_and_score_grads('adaptive_hinge', ps, mns, None, pi, mni, meta, w2)[0],
        'warp_meta_a0.2_b0.3_num_items_4': O.loss_and_score_grads('warp', ps, mns, 4, pi, mni, meta, w2)[0],
    }
    for key, val in got.items():
        # the reference's own tests: 3 decimals (tests/test_losses.py)
        np.testing.assert_almost_equal(float(val), L['expected_3dp'][key], decimal=3)
        # the reference's functions run in the build container: 1e-5 relative
        close(val, L['reference_run'][key])
    close(O.loss_and_score_grads('bpr', np.float32([3.0]), np.float32([1.5]))[0],
          L['reference_run']['doc_bpr_3.0_1.5'])
    np.testing.assert_almost_equal(L['reference_run']['doc_bpr_3.0_1.5'], 0.2157, decimal=4)
    close(O.loss_and_score_grads('warp', np.float32([1.5]), np.float32([[2.0], [3.0], [2.5]]), 30)[0],
          L['reference_run']['doc_warp_1.5_[2,3,2.5]_n30'])
    np.testing.assert_almost_equal(L['reference_run']['doc_warp_1.5_[2,3,2.5]_n30'], 31.1301, decimal=3)


def test_metrics_known_answers(known):
    M = known['metrics']
    tg = sp.coo_matrix((np.ones(8), (M['users'], M['items'])), shape=(3, 4))
    csr = tg.tocsr()
    P = np.array(M['predicted_scores'], np.float32)
    # tests/test_metrics.py:78-98 labels of the given recs
    np.testing.assert_array_equal(O._labels(csr, np.arange(3), np.array(M['recs'])), np.array(M['labels'], float))
    np.testing.assert_array_equal(O.topk_ids(P, 4), np.array(M['recs']))
    assert O.mapk(csr, np.array([1, 2]), P[[1, 2]], k=4) == M['mapk_k4_users_1_2']
    np.testing.assert_almost_equal(O.mapk(csr, np.arange(3), P, k=1), M['mapk_k1_all'])
    np.testing.assert_almost_equal(O.mrr(csr, np.arange(3), P), M['mrr_all'])
    np.testing.assert_almost_equal(O.auc_per_user(csr, np.arange(3), P), M['auc_sklearn_per_user'])
    with pytest.raises(ValueError):
        O.mapk(csr, np.arange(3), P, k=5)


class _FixedScores(dict):
    """tables stand-in whose forward returns a fixed score matrix (the reference mocks the model)."""


@pytest.mark.parametrize('batch_size', [20, 2, 1])
def test_evaluate_in_batches_known_answers(known, batch_size, monkeypatch):
    M = known['metrics']
    tg = sp.coo_matrix((np.ones(8), (M['users'], M['items'])), shape=(3, 4))
    P = np.array(M['predicted_scores'], np.float32)
    monkeypatch.setattr(O, 'get_preds', lambda tables, user_ids, n_items, y_range=None: P[np.asarray(user_ids)])
    m, r, a = O.evaluate_in_batches([O.mapk, O.mrr, O.auc], tg, None, k=4, batch_size=batch_size)
    exp = M['evaluate_in_batches_k4']
    np.testing.assert_almost_equal(m, exp['mapk'])
    np.testing.assert_almost_equal(r, exp['mrr'])
    np.testing.assert_almost_equal(a, exp['auc'])


# ------------------------------------------------------------------ reference-generated vectors
def test_loss_and_score_grads_vs_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'loss_cases.npz'))
    meta = {'a': z['meta_a'], 'b': z['meta_b']}
    w = {'a': 0.2, 'b': 0.3}
    N = int(z['num_items'])
    for tag in z['cases']:
        kind, use_meta, variant = tag.split('|')
        sp_, sn_ = z[tag + '|s_pos'], z[tag + '|s_neg']
        pos, neg = z[tag + '|pos'], z[tag + '|neg']
        single = kind in ('bpr', 'hinge')
        kw = dict(metadata=meta, metadata_weights=w) if use_meta == '1' else {}
        loss, gp, gn, sel = O.loss_and_score_grads(
            kind, sp_, sn_[0] if single else sn_, N, pos, neg[0] if single else neg, **kw)
        close(loss, z[tag + '|loss'])
        scale = max(np.abs(z[tag + '|g_pos']).max(), 1e-12)
        close(gp, z[tag + '|g_pos'], atol_scale=scale)
        close(gn.reshape(sn_.shape), z[tag + '|g_neg'], atol_scale=scale)


def test_forward_vs_torch(golden_dir):
    z = np.load(os.path.join(golden_dir, 'forward_cases.npz'))
    for D in (1, 3, 8, 10, 30, 64, 100, 128, 256):
        tables = {n: z[f'D{D}|t|{n}'] for n in ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')}
        for y, yr in ((0, None), (1, (1.0, 5.0))):
            p = O.mf_forward(tables, z[f'D{D}|users'], z[f'D{D}|items'], yr)
            ref = z[f'D{D}|y{y}|preds']
            close(p, ref)


def test_calculate_loss_and_sparse_grads_vs_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'train_cases.npz'))
    meta = {'a': z['meta_a'], 'b': z['meta_b']}
    w = {'a': 0.2, 'b': 0.3}
    names = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')
    for tag in z['cases']:
        kind, D, K, m = tag.split('|')
        tables = {n: z[f'{tag}|t|{n}'] for n in names}
        kw = dict(metadata=meta, metadata_weights=w) if m == 'm1' else {}
        out = O.calculate_loss(tables, z[f'{tag}|users'], z[f'{tag}|pos'], z[f'{tag}|neg'], kind, **kw)
        close(out['loss'], z[f'{tag}|loss'])
        close(out['s_pos'], z[f'{tag}|s_pos'])
        close(out['s_neg'], z[f'{tag}|s_neg'])
        for n in names:
            rows, vals = out['grads'][n]
            np.testing.assert_array_equal(rows, z[f'{tag}|g|{n}|rows'])      # touched set: exact
            ref = z[f'{tag}|g|{n}|vals']
            close(vals, ref, atol_scale=max(np.abs(z[f'{tag}|g|item_embeddings|vals']).max(), 1e-12))


def test_optimizer_trajectories_vs_torch(golden_dir):
    z = np.load(os.path.join(golden_dir, 'optimizer_cases.npz'))
    combos = json.loads(str(z['combos']))
    names = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')
    for tag, sparse, opt, bopt, lr, blr, wd, kind, K in combos:
        tables = {n: z[f'{tag}|t0|{n}'] for n in names}
        st = O.TrainState(tables, optimizer=opt, bias_optimizer=bopt, lr=lr, bias_lr=blr,
                          weight_decay=wd, sparse=sparse)
        for s in range(4):
            out = st.step(z[f'{tag}|s{s}|users'], z[f'{tag}|s{s}|pos'], z[f'{tag}|s{s}|neg'], kind)
            close(out['loss'], z[f'{tag}|s{s}|loss'], rtol=2e-5)
            for n in names:
                ref = z[f'{tag}|s{s}|t|{n}']
                if n == 'user_biases':
                    # d(loss)/d(user bias) is analytically 0 for every pairwise loss (the bias
                    # cancels in s+ - s-).  The reference's value is the fp32 residue of
                    # (+g) + (-g) sums in coalesce order; Adam-type optimizers normalise that
                    # residue into O(lr) random steps.  Not a parity target: only check that the
                    # reference's drift (and the oracle's, which sums in the same way) is noise-sized.
                    assert np.abs(ref).max() <= 4 * (blr or lr) * (s + 1)
                    assert np.abs(st.t[n]).max() <= 4 * (blr or lr) * (s + 1)
                    continue
                close(st.t[n], ref, rtol=2e-5)
                for key, mine in (('exp_avg', 'm'), ('exp_avg_sq', 'v'), ('sum', 'sum')):
                    k2 = f'{tag}|s{s}|state|{n}|{key}'
                    if k2 in z.files and n != 'user_biases':
                        close(st.state[n][mine], z[k2], rtol=2e-5)


def test_eval_case(golden_dir):
    z = np.load(os.path.join(golden_dir, 'eval_case.npz'))
    names = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')
    tables = {n: z[f't|{n}'] for n in names}
    N = tables['item_embeddings'].shape[0]
    coo = sp.coo_matrix((np.ones(len(z['rows'])), (z['rows'], z['cols'])), shape=(9, N))
    users = z['test_users']
    preds = O.get_preds(tables, users, N)
    close(preds, z['preds_torch'])
    # AUC: pinned to sklearn on torch's sigmoid (the reference's own oracle for this metric)
    au = O.auc_per_user(coo.tocsr(), users, z['preds_torch'])
    np.testing.assert_allclose(au, z['auc_sklearn_on_torch_sigmoid'], rtol=1e-5)
    # engineered exact ties: ids must come out in ascending order inside a tie group
    ids = O.topk_ids(z['preds_torch'], N)
    for i in range(len(users)):
        s = z['preds_torch'][i][ids[i]]
        assert np.all(np.diff(s) <= 0)
        same = np.nonzero(np.diff(s) == 0)[0]
        assert np.all(ids[i][same] < ids[i][same + 1])


def rr_tie_interval(preds_row, positives):
    """[lowest, highest] reciprocal rank a user can get when items tie exactly with the best
    positive: ``Tensor.topk``'s order inside a tie group is implementation-defined (SURVEY.md
    Appendix B), the build's rule is ascending id."""
    best = preds_row[positives].max()
    above = int((preds_row > best).sum())
    tied = np.nonzero(preds_row == best)[0]
    tied_other = len(set(tied.tolist()) - set(positives.tolist()))
    return 1.0 / (above + 1 + tied_other), 1.0 / (above + 1)


def test_oracle_metrics_vs_reference_metrics_per_user(golden_dir):
    """Per-user AP@10 / RR / AUC of the eval case from the REFERENCE'S OWN ``collie/metrics.py``
    (``tests/golden/make_golden_ref_pipeline.py``: ``mapk`` / ``mrr`` / ``auc`` called one user at a
    time on ``get_preds`` of the reference's ``MatrixFactorizationModel``) pin the oracle."""
    z = np.load(os.path.join(golden_dir, 'eval_case.npz'))
    r = np.load(os.path.join(golden_dir, 'ref_pipeline.npz'))
    names = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')
    tables = {n: z[f't|{n}'] for n in names}
    N = tables['item_embeddings'].shape[0]
    users = r['eval|users']
    np.testing.assert_array_equal(users, z['test_users'])
    csr = sp.coo_matrix((np.ones(len(z['rows'])), (z['rows'], z['cols'])), shape=(9, N)).tocsr()
    preds = r['eval|preds_reference']
    close(O.get_preds(tables, users, N), preds)
    ap, ids = O.mapk_per_user(csr, users, preds, 10)
    np.testing.assert_allclose(ap, r['eval|ap_reference'], rtol=1e-12, atol=0)
    np.testing.assert_allclose(O.auc_per_user(csr, users, preds), r['eval|auc_reference'], rtol=1e-6)
    rr = O.mrr_per_user(csr, users, preds)
    n_tie_users = 0
    for i, u in enumerate(users):
        lo, hi = rr_tie_interval(preds[i], csr[u].indices)
        assert lo - 1e-9 <= r['eval|rr_reference'][i] <= hi + 1e-9
        assert lo - 1e-9 <= rr[i] <= hi + 1e-9
        if lo == hi:
            np.testing.assert_allclose(rr[i], r['eval|rr_reference'][i], rtol=1e-6)
        else:
            n_tie_users += 1
        # top-k ids equal torch's wherever the first 11 scores hold no exact tie
        top = np.sort(preds[i])[::-1][:11]
        if np.all(np.diff(top) < 0):
            np.testing.assert_array_equal(ids[i], r['eval|top10_ids_reference'][i])
    assert n_tie_users >= 1          # the engineered tie case is in the fixture


def reference_trainer_case(golden_dir, tag):
    r = np.load(os.path.join(golden_dir, 'ref_pipeline.npz'))
    names = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')
    t0 = {n: r[f'{tag}|t0|{n}'].copy() for n in names}
    t1 = {n: r[f'{tag}|t1|{n}'] for n in names}
    nb = int(r[f'{tag}|n_batches'])
    batches = [(r[f'{tag}|b{i}|users'], r[f'{tag}|b{i}|pos'], r[f'{tag}|b{i}|neg']) for i in range(nb)]
    return t0, t1, batches, r[f'{tag}|step_losses'], r[f'{tag}|epoch_losses']


REF_TRAINER_CASES = {
    'sparse_adam_bpr': dict(optimizer='sparse_adam', bias_optimizer='sgd', lr=1e-2, bias_lr=1e-2,
                            sparse=True, kind='adaptive_bpr'),
    'default_adam_hinge': dict(optimizer='adam', bias_optimizer='sgd', lr=1e-2, bias_lr=1e-2,
                               sparse=False, kind='adaptive_hinge'),
}


@pytest.mark.parametrize('tag', sorted(REF_TRAINER_CASES))
def test_oracle_steps_vs_reference_trainer_epochs(golden_dir, tag):
    """F6: two epochs of the REFERENCE'S OWN ``CollieMinimalTrainer`` +
    ``MatrixFactorizationModel`` on fixed batches: per-step losses, epoch losses and final tables
    pin the oracle's whole training step (forward, loss, sparse / dense gradients, optimizers)."""
    cfg = dict(REF_TRAINER_CASES[tag])
    kind = cfg.pop('kind')
    t0, t1, batches, step_losses, epoch_losses = reference_trainer_case(golden_dir, tag)
    st = O.TrainState(t0, **cfg)
    got = []
    for epoch in range(2):
        for users, pos, neg in batches:
            got.append(st.step(users, pos, neg, kind)['loss'])
    np.testing.assert_allclose(got, step_losses, rtol=2e-5)
    nb = len(batches)
    np.testing.assert_allclose([np.mean(got[:nb]), np.mean(got[nb:])], epoch_losses, rtol=2e-5)
    for n in t1:
        if n == 'user_biases':
            # analytically zero gradient: the reference holds fp32 cancellation residue times lr
            assert np.abs(t1[n]).max() <= 1e-6 and np.abs(st.t[n]).max() <= 1e-6
        else:
            close(st.t[n], t1[n], rtol=2e-4)


def test_hybrid_numpy_oracle_matches_torch_restatement():
    """oracle/hybrid_oracle.py (numpy) against oracle/torch_path.py:TorchHybrid (the reference's
    torch calls, hybrid_pretrained_matrix_factorization.py:327-464) run by torch here: scores and
    every gradient, with both metadata sides, stacked metadata layers and unfrozen embeddings."""
    import torch
    from oracle import hybrid_oracle as H
    from oracle.torch_path import TorchHybrid
    rng = np.random.default_rng(0)
    U, N, D, M = 30, 25, 6, 64
    tables = {'user_embeddings': rng.normal(size=(U, D)).astype(np.float32),
              'item_embeddings': rng.normal(size=(N, D)).astype(np.float32),
              'user_biases': rng.normal(size=(U, 1)).astype(np.float32),
              'item_biases': rng.normal(size=(N, 1)).astype(np.float32)}
    item_md = rng.normal(size=(N, 7)).astype(np.float32)
    user_md = rng.normal(size=(U, 4)).astype(np.float32)
    torch.manual_seed(0)
    ref = TorchHybrid(tables, item_metadata=item_md, user_metadata=user_md,
                      item_metadata_layers_dims=[9, 5], user_metadata_layers_dims=None,
                      combined_layers_dims=[12, 6], freeze_embeddings=False)
    users = rng.integers(0, U, M)
    items = rng.integers(0, N, M)
    g = rng.normal(size=M).astype(np.float32)
    scores = ref(torch.as_tensor(users), torch.as_tensor(items))
    scores.backward(torch.as_tensor(g))

    def wb(layers):
        return None if layers is None else [(l.weight.detach().numpy(), l.bias.detach().numpy())
                                            for l in layers]
    params = {'item_metadata_layers': wb(ref.item_metadata_layers), 'user_metadata_layers': None,
              'combined_layers': wb(ref.combined_layers)}
    s, cache = H.forward(params, tables, users, items, item_metadata=item_md, user_metadata=user_md)
    np.testing.assert_allclose(s, scores.detach().numpy(), rtol=1e-5, atol=1e-5)
    grads = H.backward(params, tables, cache, g)

    def close(a, b):
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-4, atol=1e-5 * max(float(b.abs().max()), 1.0))
    for (dw, db), layer in zip(grads['combined_layers'], ref.combined_layers):
        close(dw, layer.weight.grad)
        close(db, layer.bias.grad)
    for (dw, db), layer in zip(grads['item_metadata_layers'], ref.item_metadata_layers):
        close(dw, layer.weight.grad)
        close(db, layer.bias.grad)
    close(grads['user_embeddings'], ref.embeddings[0].weight.grad)
    close(grads['item_embeddings'], ref.embeddings[1].weight.grad)
    close(grads['user_biases'], ref.biases[0].weight.grad)
    close(grads['item_biases'], ref.biases[1].weight.grad)


REF_HYBRID_CASES = {
    'item_meta_adaptive': dict(item_metadata_layers_dims=[32], combined_layers_dims=[32, 16, 8],
                               freeze_embeddings=False, loss='adaptive_hinge'),
    'both_meta_bpr_frozen': dict(item_metadata_layers_dims=[10, 6], user_metadata_layers_dims=[5],
                                 combined_layers_dims=[20], freeze_embeddings=True,
                                 loss='adaptive_bpr'),
    'raw_meta_warp': dict(combined_layers_dims=[16, 4], freeze_embeddings=False, loss='warp'),
}


def load_ref_hybrid_case(golden_dir, tag):
    """Arrays of one case of ``ref_hybrid.npz`` (made by the REFERENCE'S OWN HybridPretrainedModel,
    tests/golden/make_golden_ref_hybrid.py): state, batch, loss, gradients, pair scores."""
    z = np.load(os.path.join(golden_dir, 'ref_hybrid.npz'))
    pre = tag + '|'
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


@pytest.mark.parametrize('tag', sorted(REF_HYBRID_CASES))
def test_torch_hybrid_restatement_vs_the_reference_hybrid_model(golden_dir, tag):
    """``oracle/torch_path.py:TorchHybrid`` (what the GPU tests of the hybrid head compare with)
    against the reference's own class: loss, every gradient, scores of (user, item) pairs."""
    import torch
    from oracle import torch_path as T
    c = load_ref_hybrid_case(golden_dir, tag)
    kw = dict(REF_HYBRID_CASES[tag])
    loss_kind = kw.pop('loss')
    assert str(c['loss_name']) == loss_kind + '_loss'
    tables = {'user_embeddings': c['state|embeddings.0.weight'],
              'item_embeddings': c['state|embeddings.1.weight'],
              'user_biases': c['state|biases.0.weight'], 'item_biases': c['state|biases.1.weight']}
    ref = T.TorchHybrid(tables, item_metadata=c.get('item_metadata'),
                        user_metadata=c.get('user_metadata'), **kw)
    state = {k[len('state|'):]: torch.as_tensor(v) for k, v in c.items()
             if k.startswith('state|') and not k.startswith('state|_trained_model')}
    assert set(state) == set(ref.state_dict())
    ref.load_state_dict(state)
    batch = ((torch.as_tensor(c['users']), torch.as_tensor(c['pos'])), torch.as_tensor(c['neg']))
    loss, _, _ = ref.calculate_loss(batch, T.TORCH_LOSSES[loss_kind])
    loss.backward()
    assert abs(float(loss) - float(c['loss'])) <= 1e-6 * abs(float(c['loss']))
    checked = 0
    for name, p in ref.named_parameters():
        key = 'grad|' + name
        if not p.requires_grad:
            assert key not in c
            continue
        want = c[key]
        np.testing.assert_allclose(p.grad.numpy(), want, rtol=1e-5,
                                   atol=1e-6 * max(np.abs(want).max(), 1e-3), err_msg=name)
        checked += 1
    assert checked == sum(1 for k in c if k.startswith('grad|') and '_trained_model' not in k)
    ref.eval()
    with torch.no_grad():
        got = ref(torch.as_tensor(c['pairs_users']), torch.as_tensor(c['pairs_items'])).numpy()
    np.testing.assert_allclose(got, c['pairs_scores'], rtol=1e-6, atol=1e-6)
