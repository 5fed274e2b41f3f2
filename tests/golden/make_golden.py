"""
Generates the committed golden vectors under ``tests/golden/``.  Run ONLY in the build
container (it imports ``/root/reference``); the GPU box never runs it and never sees the
reference.  Usage: ``python tests/golden/make_golden.py``.

What is the real reference here, and what is not (be precise):
  * ``collie.loss`` (bpr, hinge, adaptive_*, warp, ideal_difference_from_metadata) and
    ``collie.model.base.layers`` (ScaledEmbedding / ZeroEmbedding) are imported VERBATIM from
    ``/root/reference/collie``.  Only the package ``__init__`` files are bypassed (they pull in
    ``pytorch_lightning`` / ``fire`` / ``tables``, which are not installed); no stand-in for any
    third-party library is written.
  * ``MatrixFactorizationModel`` / ``BasePipeline`` / ``collie.metrics`` cannot be imported
    (they need ``pytorch_lightning`` and ``torchmetrics``).  Their few lines of arithmetic glue
    (matrix_factorization.py:143-159, base_pipeline.py:610-634) are replayed by
    ``oracle/torch_path.TorchMF`` with the same torch calls on the reference's own embedding
    layer classes and loss functions.  Optimizers are torch's own (the reference calls them
    unmodified, base_pipeline.py:418-433).
  * Metric vectors are the reference's own known answers (tests/test_metrics.py,
    tests/fixtures/metrics_fixtures.py) re-expressed as JSON, plus sklearn ``roc_auc_score``
    values (the reference's own AUC oracle).  The larger eval case's per-user AP / RR / AUC come
    from the reference's own ``collie/metrics.py`` (``make_golden_ref_pipeline.py``).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF = '/root/reference/collie'
for name, path in (('collie', REF), ('collie.model', REF + '/model'),
                   ('collie.model.base', REF + '/model/base')):
    mod = types.ModuleType(name)
    mod.__path__ = [path]
    sys.modules[name] = mod

import collie.loss as ref_loss                                      # noqa: E402  (verbatim reference)
from collie.model.base.layers import ScaledEmbedding, ZeroEmbedding  # noqa: E402  (verbatim reference)

from oracle import torch_path                                       # noqa: E402

REF_LOSSES = {
    'bpr': ref_loss.bpr_loss,
    'hinge': ref_loss.hinge_loss,
    'adaptive_bpr': ref_loss.adaptive_bpr_loss,
    'adaptive_hinge': ref_loss.adaptive_hinge_loss,
    'warp': ref_loss.warp_loss,
}


def ref_tables(U, N, D, seed):
    """Tables initialised by the reference's own layer classes (layers.py:6-17)."""
    torch.manual_seed(seed)
    ue = ScaledEmbedding(U, D)
    ie = ScaledEmbedding(N, D)
    ub = ZeroEmbedding(U, 1)
    ib = ZeroEmbedding(N, 1)
    t = {'user_embeddings': ue.weight.detach().numpy().copy(),
         'item_embeddings': ie.weight.detach().numpy().copy(),
         'user_biases': ub.weight.detach().numpy().copy(),
         'item_biases': ib.weight.detach().numpy().copy()}
    return t


# ---------------------------------------------------------------------------------------------
def f1_known_answers():
    """The reference's own test data, re-expressed as JSON (data only)."""
    d = {
        'source': 'tests/fixtures/loss_fixtures.py:5-83, tests/test_losses.py:31-216, '
                  'tests/fixtures/metrics_fixtures.py:7-75, tests/test_metrics.py:27-154,230-256, '
                  'docs/source/loss.rst',
        'loss': {
            'scores': [1.1, 1.3, 4.7, -7.234, -1.3, 0.7, 4.7, -2.2468, -4.7, 1.3, 3.56, -0.8924,
                       4.01, 2.7, 3.7, -5.2468, 3.89, 12.8, -1.7, -0.8143],
            'positive_items': [0, 1, 2, 3],
            'negative_items': [4, 5, 6, 7],
            'many_negative_items': [[4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15], [16, 17, 18, 19]],
            'metadata_a': [0, 1, 1, 2, 0, 0, 0, 1, 2, 1, 1, 2, 0, 0, 1, 2, 0, 0, 1, 2],
            'metadata_b': [1, 2, 2, 3, 3, 2, 1, 3, 3, 2, 2, 2, 1, 1, 2, 2, 1, 1, 2, 2],
            'metadata_a_diff': [.8, 1, 1, 1],
            'metadata_a_and_2_diff': [[.8, .7, 1, .7], [1, .5, .5, .8], [.5, 1, .5, .8],
                                      [.5, 1, .5, .8]],
            'expected_3dp': {
                'bpr': (1.93074 + 1.36897) / 4,
                'hinge': (7.3872 + 37.00656384) / 4,
                'adaptive_bpr': 6.592 / 4,
                'adaptive_hinge': 252.420 / 4,
                'warp_num_items_4': (11.366 + 72.385) / 4,
                'bpr_meta_a_0.2': (1.73073 + 1.37570) / 4,
                'hinge_meta_a_0.2': (7.3872 + 37.00656384) / 4,
                'adaptive_bpr_meta_a0.2_b0.3': (2.746 + 2.088) / 4,
                'adaptive_hinge_meta_a0.2_b0.3': 61.2829,
                'warp_meta_a0.2_b0.3_num_items_4': (10.390 + 65.063) / 4,
            },
        },
        'metrics': {
            'users': [0, 0, 0, 1, 1, 1, 2, 2],
            'items': [0, 1, 2, 1, 2, 3, 0, 2],
            'predicted_scores': [[0.9, 0.4, 0.5, 0.7], [0.1, 1.2, 1.7, 0.12], [-1, -2, -3, 4]],
            'recs': [[0, 3, 2, 1], [2, 1, 3, 0], [3, 0, 1, 2]],
            'labels': [[1, 0, 1, 1], [1, 1, 1, 0], [0, 1, 0, 1]],
            'mapk_k4_users_1_2': 0.75,
            'mapk_k1_all': 2 / 3,
            'mrr_all': (1 + 1 + 1 / 2) / 3,
            'evaluate_in_batches_k4': {'mapk': 0.7685185, 'mrr': 0.8333333, 'auc': 0.5277777},
        },
    }
    # values actually produced by the reference's loss functions on its own fixture (7+ digits)
    S = torch.tensor(d['loss']['scores'])
    pi = torch.tensor(d['loss']['positive_items'])
    ni = torch.tensor(d['loss']['negative_items'])
    mni = torch.tensor(d['loss']['many_negative_items'])
    ma = torch.tensor(d['loss']['metadata_a'])
    mb = torch.tensor(d['loss']['metadata_b'])
    ps, ns, mns = S[pi], S[ni], S[mni]
    d['loss']['reference_run'] = {
        'bpr': ref_loss.bpr_loss(ps, ns).item(),
        'hinge': ref_loss.hinge_loss(ps, ns).item(),
        'adaptive_bpr': ref_loss.adaptive_bpr_loss(ps, mns).item(),
        'adaptive_hinge': ref_loss.adaptive_hinge_loss(ps, mns).item(),
        'warp_num_items_4': ref_loss.warp_loss(ps, mns, num_items=4).item(),
        'bpr_meta_a_0.2': ref_loss.bpr_loss(ps, ns, positive_items=pi, negative_items=ni,
                                            metadata={'a': ma}, metadata_weights={'a': .2}).item(),
        'hinge_meta_a_0.2': ref_loss.hinge_loss(ps, ns, positive_items=pi, negative_items=ni,
                                                metadata={'a': ma},
                                                metadata_weights={'a': .2}).item(),
        'adaptive_bpr_meta_a0.2_b0.3': ref_loss.adaptive_bpr_loss(
            ps, mns, positive_items=pi, negative_items=mni, metadata={'a': ma, 'b': mb},
            metadata_weights={'a': .2, 'b': .3}).item(),
        'adaptive_hinge_meta_a0.2_b0.3': ref_loss.adaptive_hinge_loss(
            ps, mns, positive_items=pi, negative_items=mni, metadata={'a': ma, 'b': mb},
            metadata_weights={'a': .2, 'b': .3}).item(),
        'warp_meta_a0.2_b0.3_num_items_4': ref_loss.warp_loss(
            ps, mns, num_items=4, positive_items=pi, negative_items=mni,
            metadata={'a': ma, 'b': mb}, metadata_weights={'a': .2, 'b': .3}).item(),
        # docs/source/loss.rst known answers
        'doc_bpr_3.0_1.5': ref_loss.bpr_loss(torch.tensor([3.0]), torch.tensor([1.5])).item(),
        'doc_warp_1.5_[2,3,2.5]_n30': ref_loss.warp_loss(
            torch.tensor([1.5]), torch.tensor([[2.0], [3.0], [2.5]]), num_items=30).item(),
    }
    # the reference's AUC oracle is sklearn (tests/test_metrics.py:146-152)
    from sklearn.metrics import roc_auc_score
    import scipy.sparse as sp
    m = d['metrics']
    tg = sp.coo_matrix((np.ones(8), (m['users'], m['items'])), shape=(3, 4)).tocsr()
    P = np.array(m['predicted_scores'], np.float32)
    m['auc_sklearn_per_user'] = [float(roc_auc_score(tg[i].toarray()[0], P[i])) for i in range(3)]
    with open(os.path.join(HERE, 'ref_known_answers.json'), 'w') as f:
        json.dump(d, f, indent=1)


# ---------------------------------------------------------------------------------------------
def score_cases():
    """Loss value + d(loss)/d(scores) from the reference's loss functions (autograd)."""
    rng = np.random.default_rng(7)
    out = {}
    N = 40
    meta = {'a': rng.integers(0, 3, N), 'b': rng.integers(0, 4, N)}
    weights = {'a': 0.2, 'b': 0.3}
    out['meta_a'] = meta['a']
    out['meta_b'] = meta['b']
    out['num_items'] = np.int64(N)
    cases = []
    for kind in REF_LOSSES:
        for use_meta in (False, True):
            for variant in ('random', 'ties', 'kink'):
                B = 33
                K = 1 if kind in ('bpr', 'hinge') else 5
                sp_ = rng.normal(0, 1.5, B).astype(np.float32)
                sn_ = rng.normal(0, 1.5, (K, B)).astype(np.float32)
                pos = rng.integers(0, N, B)
                neg = rng.integers(0, N, (K, B))
                if variant == 'ties':
                    sn_[:, ::2] = sn_[0, ::2]               # all-equal negatives -> first argmax
                    if K > 1:
                        sn_[1:, 1::4] = sn_[1, 1::4]
                if variant == 'kink':
                    # margins that are exactly representable: c - (s+ - s-) == 0 for some samples
                    sp_ = np.round(sp_ * 4) / 4
                    sn_ = np.round(sn_ * 4) / 4
                    sn_[0, :8] = sp_[:8] - 1.0               # hinge exactly at the kink (c = 1)
                    sn_[:, 8:12] = sp_[8:12] - 3.0           # no violation at all (WARP sentinel)
                tp = torch.tensor(sp_, requires_grad=True)
                tn = torch.tensor(sn_ if K > 1 else sn_[0], requires_grad=True)
                kw = {}
                if use_meta:
                    kw = dict(positive_items=torch.tensor(pos),
                              negative_items=torch.tensor(neg if K > 1 else neg[0]),
                              metadata={k: torch.tensor(v) for k, v in meta.items()},
                              metadata_weights=weights)
                loss = REF_LOSSES[kind](tp, tn, num_items=N, **kw)
                loss.backward()
                tag = f'{kind}|{int(use_meta)}|{variant}'
                cases.append(tag)
                out[tag + '|s_pos'] = sp_
                out[tag + '|s_neg'] = sn_
                out[tag + '|pos'] = pos
                out[tag + '|neg'] = neg
                out[tag + '|loss'] = np.float32(loss.item())
                out[tag + '|g_pos'] = tp.grad.numpy()
                out[tag + '|g_neg'] = tn.grad.numpy().reshape(K, B)
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, 'loss_cases.npz'), **out)


# ---------------------------------------------------------------------------------------------
def make_batch(rng, U, N, B, K, dup=True):
    users = rng.integers(0, U, B).astype(np.int32)
    pos = rng.integers(0, N, B).astype(np.int32)
    neg = rng.integers(0, N, (B, K)).astype(np.int64)
    if dup and B >= 8:
        users[3] = users[0]
        users[5] = users[0]
        pos[4] = pos[1]
        neg[2, 0] = pos[1]
        neg[6, K - 1] = neg[7, 0]
    return users, pos, neg


def coalesced(grad):
    g = grad.coalesce()
    return g.indices()[0].numpy().copy(), g.values().numpy().copy()


def train_cases():
    """calculate_loss + sparse grads of all four tables, reference losses + reference layers."""
    rng = np.random.default_rng(11)
    out = {}
    cases = []
    U, N = 50, 40
    meta = {'a': rng.integers(0, 3, N), 'b': rng.integers(0, 4, N)}
    weights = {'a': 0.2, 'b': 0.3}
    out['meta_a'], out['meta_b'] = meta['a'], meta['b']
    for D in (8, 10, 64, 128):
        for kind in REF_LOSSES:
            Ks = (1,) if kind in ('bpr', 'hinge') else (4, 10)
            for K in Ks:
                for use_meta in (False, True):
                    if use_meta and (D not in (10, 64) or K == 1):
                        # K == 1 with metadata raises inside the reference itself
                        # (metadata_utils.py:64-67: in-place add of [1,B] into [B]); not a case
                        continue
                    tables = ref_tables(U, N, D, seed=D * 100 + K)
                    # non-zero biases so that the bias gathers are exercised
                    tables['user_biases'] = rng.normal(0, 0.05, (U, 1)).astype(np.float32)
                    tables['item_biases'] = rng.normal(0, 0.05, (N, 1)).astype(np.float32)
                    # scale embeddings up so that hinge/warp margins are not all violated
                    tables['user_embeddings'] *= np.float32(8.0 * np.sqrt(D))
                    tables['item_embeddings'] *= np.float32(8.0 * np.sqrt(D))
                    B = 37
                    users, pos, neg = make_batch(rng, U, N, B, K)
                    model = torch_path.TorchMF(U, N, D, sparse=True, tables=tables)
                    kw = {}
                    if use_meta:
                        kw = dict(metadata={k: torch.tensor(v) for k, v in meta.items()},
                                  metadata_weights=weights)
                    loss, sp_, sn_ = model.calculate_loss(
                        ((torch.tensor(users), torch.tensor(pos)), torch.tensor(neg)),
                        REF_LOSSES[kind], **kw)
                    loss.backward()
                    tag = f'{kind}|D{D}|K{K}|m{int(use_meta)}'
                    cases.append(tag)
                    for name, arr in tables.items():
                        out[f'{tag}|t|{name}'] = arr
                    out[f'{tag}|users'], out[f'{tag}|pos'], out[f'{tag}|neg'] = users, pos, neg
                    out[f'{tag}|loss'] = np.float32(loss.item())
                    out[f'{tag}|s_pos'] = sp_.detach().numpy()
                    out[f'{tag}|s_neg'] = sn_.detach().numpy()
                    for name in tables:
                        rows, vals = coalesced(getattr(model, name).weight.grad)
                        out[f'{tag}|g|{name}|rows'] = rows
                        out[f'{tag}|g|{name}|vals'] = vals
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, 'train_cases.npz'), **out)


# ---------------------------------------------------------------------------------------------
OPT_COMBOS = [
    # tag, sparse, optimizer, bias_optimizer, lr, bias_lr, weight_decay, loss, K
    ('sparse_adam+sgd', True, 'sparse_adam', 'sgd', 1e-2, 1e-1, 0.0, 'adaptive_bpr', 10),
    ('sparse_adam+sparse_adam', True, 'sparse_adam', 'sparse_adam', 1e-2, 1e-2, 0.0, 'adaptive_hinge', 4),
    ('adagrad_sparse+adagrad', True, 'adagrad', 'adagrad', 5e-2, 5e-2, 0.0, 'warp', 6),
    ('sgd_sparse+sgd', True, 'sgd', 'sgd', 5e-1, 1e-1, 0.0, 'bpr', 1),
    ('adam+sgd_dense', False, 'adam', 'sgd', 1e-1, 1e-2, 0.0, 'adaptive_hinge', 10),
    ('adam+sgd_dense_wd', False, 'adam', 'sgd', 1e-2, 1e-2, 1e-3, 'adaptive_bpr', 5),
    ('adam_single_dense', False, 'adam', None, 1e-2, None, 0.0, 'hinge', 1),
    ('adagrad_dense+sgd_wd', False, 'adagrad', 'sgd', 5e-2, 1e-2, 1e-3, 'adaptive_bpr', 3),
    ('sgd_dense+adam', False, 'sgd', 'adam', 5e-1, 1e-2, 0.0, 'warp', 5),
]


def optimizer_cases():
    """Tables + optimizer state after each of 4 steps, torch optimizers (the reference's own calls)."""
    rng = np.random.default_rng(23)
    out = {}
    U, N, D, B = 60, 45, 16, 48
    steps = 4
    for tag, sparse, opt, bopt, lr, blr, wd, kind, K in OPT_COMBOS:
        tables = ref_tables(U, N, D, seed=5)
        tables['user_embeddings'] *= np.float32(20.0)
        tables['item_embeddings'] *= np.float32(20.0)
        tables['item_biases'] = rng.normal(0, 0.05, (N, 1)).astype(np.float32)
        model = torch_path.TorchMF(U, N, D, sparse=sparse, tables=tables)
        optimizers = torch_path.make_optimizers(model, opt, bopt, lr, blr, wd)
        for name, arr in tables.items():
            out[f'{tag}|t0|{name}'] = arr
        for s in range(steps):
            users, pos, neg = make_batch(rng, U, N, B, K)
            if s == 1:
                # a row that is looked up at step 1 (0-based) only as a non-selected negative after
                # having received gradient at step 0: zero-gradient "touched" row semantics
                neg[:, -1] = pos[0]
            loss, _, _ = torch_path.train_step(
                model, optimizers,
                ((torch.tensor(users), torch.tensor(pos)), torch.tensor(neg)), REF_LOSSES[kind])
            out[f'{tag}|s{s}|users'], out[f'{tag}|s{s}|pos'], out[f'{tag}|s{s}|neg'] = users, pos, neg
            out[f'{tag}|s{s}|loss'] = np.float32(loss.item())
            for name, arr in model.tables().items():
                out[f'{tag}|s{s}|t|{name}'] = arr
            for o in optimizers:
                for p, st in o.state.items():
                    pname = [n for n, q in model.named_parameters() if q is p][0].split('.')[0]
                    for key, val in st.items():
                        if torch.is_tensor(val) and val.numel() > 1:
                            out[f'{tag}|s{s}|state|{pname}|{key}'] = val.detach().numpy().copy()
    out['combos'] = np.array(json.dumps(OPT_COMBOS))
    np.savez_compressed(os.path.join(HERE, 'optimizer_cases.npz'), **out)


# ---------------------------------------------------------------------------------------------
def eval_case():
    """Bu=7 users x N=300 items, D=16; engineered exact ties; users with 1..12 positives."""
    import scipy.sparse as sp
    from sklearn.metrics import roc_auc_score
    from oracle import mf_oracle
    rng = np.random.default_rng(3)
    U, N, D = 9, 300, 16
    tables = ref_tables(U, N, D, seed=9)
    tables['user_embeddings'] *= np.float32(30.0)
    tables['item_embeddings'] *= np.float32(30.0)
    tables['user_biases'] = rng.normal(0, 0.05, (U, 1)).astype(np.float32)
    tables['item_biases'] = rng.normal(0, 0.05, (N, 1)).astype(np.float32)
    # exact ties: duplicate item rows (embedding and bias) -> identical scores for every user
    for a, b in ((17, 3), (200, 3), (250, 100), (299, 0)):
        tables['item_embeddings'][a] = tables['item_embeddings'][b]
        tables['item_biases'][a] = tables['item_biases'][b]
    rows, cols = [], []
    test_users = [0, 1, 2, 4, 5, 7, 8]
    for j, u in enumerate(test_users):
        n_pos = [1, 2, 3, 5, 8, 12, 4][j]
        items = rng.choice(N, n_pos, replace=False)
        if j == 1:
            items = np.array([3, 17])          # both members of a tie group are positives
        if j == 2:
            items = np.array([200, 5, 9])      # one member of a tie group is positive
        rows += [u] * len(items)
        cols += list(items)
    model = torch_path.TorchMF(U, N, D, sparse=False, tables=tables)
    preds, _ = torch_path.eval_users_torch(model, np.array(test_users), N, 10)
    preds = preds.numpy()
    coo = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, N))
    csr = coo.tocsr()
    sig = torch.sigmoid(torch.tensor(preds)).numpy()
    au_sk = np.array([roc_auc_score(csr[u].toarray()[0], sig[i]) for i, u in enumerate(test_users)])
    out = {f't|{k}': v for k, v in tables.items()}
    out.update(rows=np.array(rows), cols=np.array(cols), test_users=np.array(test_users),
               preds_torch=preds, auc_sklearn_on_torch_sigmoid=au_sk)
    # per-user AP / RR / AUC and top-k ids of this case come from the reference's own
    # collie/metrics.py: tests/golden/make_golden_ref_pipeline.py -> ref_pipeline.npz
    np.savez_compressed(os.path.join(HERE, 'eval_case.npz'), **out)


def forward_cases():
    """Plain forward over arbitrary pairs incl. y_range (matrix_factorization.py:143-159)."""
    rng = np.random.default_rng(5)
    out = {}
    U, N = 70, 55
    for D in (1, 3, 8, 10, 30, 64, 100, 128, 256):
        tables = ref_tables(U, N, D, seed=D)
        tables['user_biases'] = rng.normal(0, 0.5, (U, 1)).astype(np.float32)
        tables['item_biases'] = rng.normal(0, 0.5, (N, 1)).astype(np.float32)
        tables['user_embeddings'] *= np.float32(10.0 * np.sqrt(D))
        users = rng.integers(0, U, 301)
        items = rng.integers(0, N, 301)
        for yr in (None, (1.0, 5.0)):
            model = torch_path.TorchMF(U, N, D, sparse=False, tables=tables, y_range=yr)
            with torch.no_grad():
                p = model(torch.tensor(users), torch.tensor(items)).numpy()
            out[f'D{D}|y{int(yr is not None)}|preds'] = p
        out[f'D{D}|users'], out[f'D{D}|items'] = users, items
        for name, arr in tables.items():
            out[f'D{D}|t|{name}'] = arr
    np.savez_compressed(os.path.join(HERE, 'forward_cases.npz'), **out)


if __name__ == '__main__':
    torch.set_num_threads(1)
    f1_known_answers()
    score_cases()
    train_cases()
    optimizer_cases()
    eval_case()
    forward_cases()
    print('golden vectors written to', HERE)
