"""
CPU oracle for Collie's matrix-factorization train/eval hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``collie_amd/`` imports it; the product path fails loudly when the HIP
library is missing instead of falling back to this file.

It is a from-scratch numpy (fp32) restatement of the reference's algorithm.  Every function
cites the reference ``file:line`` it follows (paths relative to ``/root/reference``).

Pinning status ("Oracle"):
  * losses + partial credit: pinned against the reference's own ``collie.loss`` package
    (imported verbatim in the build container by ``tests/golden/make_golden.py``) and against the
    reference's known-answer tests (``tests/test_losses.py:31-216``,
    ``tests/fixtures/loss_fixtures.py:5-83``).
  * metrics: pinned against the reference's known answers
    (``tests/test_metrics.py:101-154,230-256``, ``tests/fixtures/metrics_fixtures.py:7-75``) and
    sklearn's ``roc_auc_score`` (the reference's own oracle for AUC, ``tests/test_metrics.py:140-154``).
    Since round 2 also against ``collie/metrics.py`` itself: ``tests/golden/ref_pipeline.npz`` holds
    per-user AP / RR / AUC, ``get_preds`` rows and top-k ids computed by the reference's own
    functions (imported through ``tests/golden/ref_harness.py``, ``auroc`` mapped to sklearn as
    the reference's test does).
  * forward / sparse gradients / optimizers: the reference delegates these to torch
    (``nn.Embedding``, ``torch.optim.*``); pinned against the same torch calls run in the build
    container (``oracle/torch_path.py`` -> ``tests/golden/*.npz``).  The reference holds no
    known-answer test for them.
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------------
# A1  table initialisation  (collie/model/base/layers.py:6-17)
# --------------------------------------------------------------------------------------------
def init_tables(num_users, num_items, dim, rng):
    """ScaledEmbedding: N(0, (1/(2.5*dim))^2); ZeroEmbedding: 0  (layers.py:10,17)."""
    std = 1.0 / (dim * 2.5)
    return {
        'user_embeddings': (rng.standard_normal((num_users, dim)) * std).astype(F32),
        'item_embeddings': (rng.standard_normal((num_items, dim)) * std).astype(F32),
        'user_biases': np.zeros((num_users, 1), F32),
        'item_biases': np.zeros((num_items, 1), F32),
    }


def _sigmoid32(x):
    x = np.asarray(x, F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32)


# --------------------------------------------------------------------------------------------
# A2  MatrixFactorizationModel.forward  (collie/model/matrix_factorization.py:143-159)
# --------------------------------------------------------------------------------------------
def mf_forward(tables, users, items, y_range=None):
    ue = tables['user_embeddings'][users]
    ie = tables['item_embeddings'][items]
    preds = ((ue * ie).sum(axis=1, dtype=F32)
             + tables['user_biases'][users, 0]
             + tables['item_biases'][items, 0]).astype(F32)
    if y_range is not None:
        preds = (_sigmoid32(preds) * F32(y_range[1] - y_range[0]) + F32(y_range[0])).astype(F32)
    return preds


def dropout_multipliers(n_pairs, dim, p, seed, call):
    """The multipliers (0 or 1 / (1 - p)) ``nn.Dropout`` applies to the gathered user rows and item
    rows of one training-mode forward call (matrix_factorization.py:143-147), drawn from the
    build's counter-based stream (Philox4x32-10, counter = (pair, element, call), key = seed): the
    reference's torch generator stream cannot be reproduced on a device, so the STREAM is parity
    unpinned; the distribution (independent Bernoulli(1 - p) per element and row) and everything
    computed from a given mask are the reference's."""
    from oracle.sampler_oracle import MASK32, philox4x32_10
    thr = max(int(float(p) * 4294967296.0), 1)
    scale = F32(1.0) / (F32(1.0) - F32(p)) if p < 1 else F32(0.0)
    mu = np.zeros((n_pairs, dim), F32)
    mi = np.zeros((n_pairs, dim), F32)
    key = (seed & MASK32, ((seed >> 32) ^ (call >> 32)) & MASK32)
    for q in range(n_pairs):
        for d in range(dim):
            r = philox4x32_10((q & MASK32, (q >> 32) & MASK32, d, call & MASK32), key)
            mu[q, d] = scale if r[0] >= thr else 0.0
            mi[q, d] = scale if r[1] >= thr else 0.0
    return mu, mi


def mf_forward_dropout(tables, users, items, p, seed, call, grad_out=None):
    """Training-mode forward with dropout on both rows, and (if ``grad_out`` is given) the values
    of the uncoalesced sparse gradients autograd produces: d/dU rows, d/dI rows, d/d bias."""
    mu, mi = dropout_multipliers(len(users), tables['user_embeddings'].shape[1], p, seed, call)
    ue = (tables['user_embeddings'][users] * mu).astype(F32)
    ie = (tables['item_embeddings'][items] * mi).astype(F32)
    preds = ((ue * ie).sum(axis=1, dtype=F32) + tables['user_biases'][users, 0]
             + tables['item_biases'][items, 0]).astype(F32)
    if grad_out is None:
        return preds
    g = np.asarray(grad_out, F32)[:, None]
    return preds, ((g * ie) * mu).astype(F32), ((g * ue) * mi).astype(F32), g.astype(F32)


# --------------------------------------------------------------------------------------------
# A7  ideal_difference_from_metadata  (collie/loss/metadata_utils.py:60-71)
# --------------------------------------------------------------------------------------------
def ideal_difference(positive_items, negative_items, metadata, metadata_weights):
    weight_sum = sum(metadata_weights.values())
    if weight_sum > 1:
        raise ValueError(f'sum of metadata weights was {weight_sum}, must be <=1')
    positive_items = np.asarray(positive_items)
    negative_items = np.asarray(negative_items)
    match_frac = np.zeros(np.broadcast(positive_items, negative_items).shape, F32)
    for k, array in metadata.items():
        array = np.asarray(array).reshape(-1)
        match_frac = match_frac + (
            (array[positive_items] == array[negative_items]).astype(np.int32)
            * F32(metadata_weights[k])
        ).astype(F32)
    return (F32(1.0) - match_frac).astype(F32)


# --------------------------------------------------------------------------------------------
# A4-A6  losses with explicit gradients w.r.t. the scores
# --------------------------------------------------------------------------------------------
LOSS_KINDS = ('bpr', 'hinge', 'adaptive_bpr', 'adaptive_hinge', 'warp')


def loss_and_score_grads(kind, s_pos, s_neg, num_items=None, positive_items=None,
                         negative_items=None, metadata=None, metadata_weights=None):
    """
    Returns (loss, g_pos[B], g_neg[same shape as s_neg], sel[B]).

    ``s_neg`` is [B] for 'bpr'/'hinge' and [K, B] for the others (the layout
    ``calculate_loss`` feeds, base_pipeline.py:613-623).  ``sel[b]`` is the index k of the
    single negative that receives gradient (all five losses touch exactly one, or none).

    bpr:  collie/loss/bpr.py:75-89        hinge: collie/loss/hinge.py:71-85
    adaptive_*: bpr.py:153-167, hinge.py:157-171 (torch.max(dim=0): first argmax on ties)
    warp: collie/loss/warp.py:73-154
    """
    s_pos = np.asarray(s_pos, F32)
    s_neg = np.asarray(s_neg, F32)
    B = s_pos.shape[0]
    has_meta = metadata is not None and len(metadata) > 0
    invB = F32(1.0) / F32(B)

    if kind in ('bpr', 'hinge', 'adaptive_bpr', 'adaptive_hinge'):
        if kind.startswith('adaptive'):
            sel = np.argmax(s_neg, axis=0)            # first max on ties, as torch.max
            sn = s_neg[sel, np.arange(B)]
            neg_sel_items = None
            if negative_items is not None and positive_items is not None:
                neg_sel_items = np.asarray(negative_items)[sel, np.arange(B)]
        else:
            sel = np.zeros(B, np.int64)
            sn = s_neg.reshape(B)
            neg_sel_items = negative_items
        c = F32(1.0)
        if has_meta:
            c = ideal_difference(positive_items, neg_sel_items, metadata, metadata_weights)
        d = (s_pos - sn).astype(F32)
        if 'bpr' in kind:
            sig = _sigmoid32(d)
            ell = (c - sig).astype(F32)
            dl_dd = (-(sig * (F32(1.0) - sig)) * (F32(1.0) + F32(2.0) * ell) * invB).astype(F32)
        else:
            x = (c - d).astype(F32)
            ell = np.maximum(x, F32(0.0)).astype(F32)
            # torch.clamp(min=0) backward passes gradient where x >= 0 (kink included)
            dl_dd = np.where(x >= 0, -(F32(1.0) + F32(2.0) * ell) * invB, F32(0.0)).astype(F32)
        loss = (ell.sum(dtype=F32) + (ell * ell).sum(dtype=F32)) * invB
        g_pos = dl_dd
        g_neg = np.zeros_like(s_neg)
        if s_neg.ndim == 2:
            g_neg[sel, np.arange(B)] = -dl_dd
        else:
            g_neg[:] = -dl_dd
        return F32(loss), g_pos, g_neg, sel

    if kind == 'warp':
        K = s_neg.shape[0]
        c = F32(1.0)
        if has_meta:
            pos_rep = np.tile(np.asarray(positive_items)[None, :], (K, 1))      # warp.py:73-74
            c = ideal_difference(pos_rep, negative_items, metadata, metadata_weights).T  # [B,K]
        h = (c - s_pos[:, None] + s_neg.T).astype(F32)                          # warp.py:99 [B,K]
        viol = h > 0                                                             # warp.py:144-145
        any_v = viol.any(axis=1)
        t = np.where(any_v, np.argmax(viol, axis=1), K)                          # warp.py:148-152
        tries = (t + 1).astype(F32)                                              # warp.py:111
        w = np.log(F32(num_items) / tries, dtype=F32)                            # warp.py:120
        count = (tries <= K).astype(F32)                                         # warp.py:123
        h_ones = np.concatenate([h, np.ones((B, 1), F32)], axis=1)               # warp.py:103
        ht = h_ones[np.arange(B), t]
        ell = (w * ht * count).astype(F32)                                       # warp.py:125-131
        loss = (ell.sum(dtype=F32) + (ell * ell).sum(dtype=F32)) * invB          # warp.py:133
        dl_dh = (w * count * (F32(1.0) + F32(2.0) * ell) * invB).astype(F32)
        dl_dh = np.where(any_v, dl_dh, F32(0.0)).astype(F32)
        g_pos = (-dl_dh).astype(F32)
        g_neg = np.zeros_like(s_neg)
        sel = np.where(any_v, t, 0)
        g_neg[sel, np.arange(B)] = dl_dh
        return F32(loss), g_pos, g_neg, sel

    raise ValueError(f'{kind} is not a valid loss function.')


# --------------------------------------------------------------------------------------------
# A3 + A8  calculate_loss and the sparse gradients autograd would produce
#          (collie/model/base/base_pipeline.py:603-634; nn.Embedding backward)
# --------------------------------------------------------------------------------------------
def calculate_loss(tables, users, pos_items, neg_items_BK, kind, num_items=None,
                   metadata=None, metadata_weights=None):
    """
    ``neg_items_BK`` is the loader's [B, K] array; it is transposed to [K, B] exactly as
    base_pipeline.py:613 does.  Returns a dict with the loss, the scores and the *coalesced*
    gradients of the four tables: ``{table: (rows_sorted_unique, values)}`` where ``rows``
    contains every looked-up row (zero-gradient rows included).
    """
    users = np.asarray(users).astype(np.int64)
    pos_items = np.asarray(pos_items).astype(np.int64)
    neg = np.asarray(neg_items_BK).astype(np.int64).T                     # [K, B]
    K, B = neg.shape
    s_pos = mf_forward(tables, users, pos_items)                          # :616
    users_rep = np.tile(users, K)                                         # :619
    neg_flat = neg.reshape(-1)                                            # :620
    s_neg = mf_forward(tables, users_rep, neg_flat).reshape(K, B)         # :621-623
    if num_items is None:
        num_items = tables['item_embeddings'].shape[0]
    if kind in ('bpr', 'hinge'):
        assert K == 1
        loss, g_pos, g_neg, sel = loss_and_score_grads(
            kind, s_pos, s_neg[0], num_items, pos_items, neg[0], metadata, metadata_weights)
        g_neg = g_neg.reshape(1, B)
    else:
        loss, g_pos, g_neg, sel = loss_and_score_grads(
            kind, s_pos, s_neg, num_items, pos_items, neg, metadata, metadata_weights)

    U = tables['user_embeddings']
    I = tables['item_embeddings']
    # uncoalesced COO gradient, in the order autograd accumulates it: positive call, negative call
    u_idx = np.concatenate([users, users_rep])
    u_val = np.concatenate([g_pos[:, None] * I[pos_items], g_neg.reshape(-1)[:, None] * I[neg_flat]])
    i_idx = np.concatenate([pos_items, neg_flat])
    i_val = np.concatenate([g_pos[:, None] * U[users], g_neg.reshape(-1)[:, None] * U[users_rep]])
    b_val = np.concatenate([g_pos, g_neg.reshape(-1)])[:, None]

    def coalesce(idx, val):
        rows, inv = np.unique(idx, return_inverse=True)
        out = np.zeros((rows.shape[0], val.shape[1]), F32)
        np.add.at(out, inv, val.astype(F32))
        return rows, out

    grads = {
        'user_embeddings': coalesce(u_idx, u_val),
        'item_embeddings': coalesce(i_idx, i_val),
        'user_biases': coalesce(u_idx, b_val),
        'item_biases': coalesce(i_idx, b_val),
    }
    return {'loss': loss, 's_pos': s_pos, 's_neg': s_neg, 'g_pos': g_pos, 'g_neg': g_neg,
            'sel': sel, 'grads': grads}


def densify(grad, shape):
    rows, vals = grad
    out = np.zeros(shape, F32)
    out[rows] = vals
    return out


# --------------------------------------------------------------------------------------------
# A10/A11  optimizers (torch formulas; container copy torch/optim/_functional.py:24-84,
#          adam.py:396-547, adagrad.py:346-388, sgd.py single-tensor path)
# --------------------------------------------------------------------------------------------
def sparse_adam_step(p, m, v, step, rows, g, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """In place. ``step`` is the (already incremented) parameter-wide step count."""
    g = g.astype(F32)
    m_old = m[rows]
    v_old = v[rows]
    upd_m = ((g - m_old) * F32(1 - beta1)).astype(F32)
    upd_v = ((g * g - v_old) * F32(1 - beta2)).astype(F32)
    m[rows] = m_old + upd_m
    v[rows] = v_old + upd_v
    numer = upd_m + m_old
    denom = np.sqrt(upd_v + v_old, dtype=F32) + F32(eps)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr * math.sqrt(bc2) / bc1
    p[rows] = p[rows] + F32(-step_size) * (numer / denom).astype(F32)


def adam_step(p, m, v, step, g_dense, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    g = g_dense.astype(F32)
    if weight_decay != 0:
        g = g + F32(weight_decay) * p
    m += F32(1 - beta1) * (g - m)                       # lerp_
    v *= F32(beta2)
    v += F32(1 - beta2) * g * g                         # addcmul_
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = (np.sqrt(v, dtype=F32) / F32(bc2 ** 0.5)).astype(F32) + F32(eps)
    p += F32(-step_size) * (m / denom).astype(F32)


def adagrad_step(p, state_sum, step, lr, rows=None, g=None, g_dense=None, lr_decay=0.0,
                 eps=1e-10, weight_decay=0.0):
    clr = lr / (1 + (step - 1) * lr_decay)
    if g_dense is not None:
        gd = g_dense.astype(F32)
        if weight_decay != 0:
            gd = gd + F32(weight_decay) * p
        state_sum += gd * gd
        p += F32(-clr) * (gd / (np.sqrt(state_sum, dtype=F32) + F32(eps))).astype(F32)
    else:
        g = g.astype(F32)
        state_sum[rows] = state_sum[rows] + g * g
        std = np.sqrt(state_sum[rows], dtype=F32) + F32(eps)
        p[rows] = p[rows] + F32(-clr) * (g / std).astype(F32)


def sgd_step(p, lr, rows=None, g=None, g_dense=None, weight_decay=0.0):
    if g_dense is not None:
        gd = g_dense.astype(F32)
        if weight_decay != 0:
            gd = gd + F32(weight_decay) * p
        p += F32(-lr) * gd
    else:
        p[rows] = p[rows] + F32(-lr) * g.astype(F32)


class TrainState:
    """Tables + optimizer state for the two-optimizer setup of base_pipeline.py:361-373."""

    def __init__(self, tables, optimizer='sparse_adam', bias_optimizer='sgd', lr=1e-3,
                 bias_lr=1e-2, weight_decay=0.0, sparse=True):
        self.t = {k: v.copy() for k, v in tables.items()}
        self.sparse = sparse
        self.wd = weight_decay
        self.cfg = {}
        for name in self.t:
            algo, rate = (bias_optimizer, bias_lr) if 'bias' in name else (optimizer, lr)
            if bias_optimizer is None:
                algo, rate = optimizer, lr
            self.cfg[name] = (algo, rate)
        self.state = {name: {'step': 0,
                             'm': np.zeros_like(a), 'v': np.zeros_like(a),
                             'sum': np.zeros_like(a)} for name, a in self.t.items()}

    def step(self, users, pos, neg_BK, kind, **loss_kwargs):
        out = calculate_loss(self.t, users, pos, neg_BK, kind, **loss_kwargs)
        for name, (rows, vals) in out['grads'].items():
            algo, lr = self.cfg[name]
            st = self.state[name]
            st['step'] += 1
            p = self.t[name]
            if algo == 'sparse_adam':
                sparse_adam_step(p, st['m'], st['v'], st['step'], rows, vals, lr)
            elif algo == 'adam':
                adam_step(p, st['m'], st['v'], st['step'], densify((rows, vals), p.shape), lr,
                          weight_decay=self.wd)
            elif algo == 'adagrad':
                if self.sparse:
                    adagrad_step(p, st['sum'], st['step'], lr, rows=rows, g=vals)
                else:
                    adagrad_step(p, st['sum'], st['step'], lr,
                                 g_dense=densify((rows, vals), p.shape), weight_decay=self.wd)
            elif algo == 'sgd':
                if self.sparse:
                    sgd_step(p, lr, rows=rows, g=vals)
                else:
                    sgd_step(p, lr, g_dense=densify((rows, vals), p.shape), weight_decay=self.wd)
            else:
                raise ValueError(f'{algo} is not a valid optimizer!')
        return out


# --------------------------------------------------------------------------------------------
# A13-A17  evaluation  (collie/metrics.py)
# --------------------------------------------------------------------------------------------
def embedding_similarity(embeddings, idx):
    """base_pipeline.py:854-864: rows / L2 norm, then the matmul row of ``idx`` (fp32)."""
    e = np.asarray(embeddings, np.float32)
    e = e / np.sqrt((e * e).sum(1, dtype=np.float32))[:, None]
    return (e[[idx], :] @ e.T).squeeze().astype(np.float32)


def get_preds(tables, user_ids, n_items, y_range=None):
    """metrics.py:20-107: every (user, item) pair through ``forward``; [Bu, N]."""
    user_ids = np.asarray(user_ids).astype(np.int64)
    users = np.repeat(user_ids, n_items)
    items = np.tile(np.arange(n_items, dtype=np.int64), len(user_ids))
    return mf_forward(tables, users, items, y_range).reshape(-1, n_items)


def topk_ids(preds, k):
    """Descending score, ascending id on ties (the build's refinement of ``Tensor.topk``)."""
    n = preds.shape[1]
    ids = np.arange(n)
    out = np.empty((preds.shape[0], k), np.int64)
    for i in range(preds.shape[0]):
        order = np.lexsort((ids, -preds[i].astype(np.float64)))
        out[i] = order[:k]
    return out


def _labels(targets_csr, user_ids, ids):
    """metrics.py:135-141."""
    return np.asarray((targets_csr[np.asarray(user_ids)[:, None], ids] > 0).astype('double').toarray())


def mapk_per_user(targets_csr, user_ids, preds, k=10):
    """metrics.py:167-196 (float64 math, NaN -> 0)."""
    if k > preds.shape[1]:
        raise ValueError(f'Ensure ``k`` ({k}) is less than the number of items ({preds.shape[1]})')
    ids = topk_ids(preds, k)
    acc = _labels(targets_csr, user_ids, ids).astype(np.int32)
    weights = 1.0 / np.arange(1, k + 1, dtype=np.float64)
    denom = np.minimum(k, targets_csr[np.asarray(user_ids)].getnnz(axis=1))
    with np.errstate(divide='ignore', invalid='ignore'):
        res = (acc * acc.cumsum(axis=1) * weights).sum(axis=1) / denom
    res[np.isnan(res)] = 0
    return res, ids


def mapk(targets, user_ids, preds, k=10):
    return float(mapk_per_user(targets, user_ids, preds, k)[0].mean())


def mrr_per_user(targets_csr, user_ids, preds):
    """metrics.py:224-238: reciprocal rank of the first relevant item in the full order."""
    n = preds.shape[1]
    order = topk_ids(preds, n)
    labeled = _labels(targets_csr, user_ids, order)
    rr = np.zeros(preds.shape[0], F32)
    for i in range(preds.shape[0]):
        hits = np.nonzero(labeled[i] > 0)[0]
        if hits.size:
            rr[i] = F32(1.0) / F32(hits[0] + 1)
    return rr


def mrr(targets, user_ids, preds, k=None):
    return float(mrr_per_user(targets, user_ids, preds).mean(dtype=F32))


def auc_per_user(targets_csr, user_ids, preds):
    """metrics.py:266-280: tie-averaged binary AUROC of sigmoid(preds) (fp32) per user."""
    out = np.zeros(preds.shape[0], np.float64)
    for i, u in enumerate(user_ids):
        y = np.asarray(targets_csr[u].toarray()).reshape(-1) > 0
        p = _sigmoid32(preds[i])
        pos = p[y]
        negs = p[~y]
        if pos.size == 0 or negs.size == 0:
            out[i] = 0.0
            continue
        negs_sorted = np.sort(negs)
        lt = np.searchsorted(negs_sorted, pos, side='left')
        le = np.searchsorted(negs_sorted, pos, side='right')
        out[i] = (lt.sum() + 0.5 * (le - lt).sum()) / (pos.size * negs.size)
    return out


def auc(targets, user_ids, preds, k=None):
    return float(auc_per_user(targets, user_ids, preds).mean())


def evaluate_in_batches(metric_list, test_coo, tables, k=10, batch_size=20, y_range=None):
    """metrics.py:367-395: sorted unique test users, consecutive slices, size-weighted mean."""
    test_users = np.unique(test_coo.row)
    targets = test_coo.tocsr()
    n_items = test_coo.shape[1]
    if len(test_users) < batch_size:
        batch_size = len(test_users)
    acc = [0] * len(metric_list)
    for i in range(int(np.ceil(len(test_users) / batch_size))):
        user_range = test_users[i * batch_size:(i + 1) * batch_size]
        preds = get_preds(tables, user_range, n_items, y_range)
        for j, metric in enumerate(metric_list):
            acc[j] += metric(targets, user_range, preds, k) * len(user_range)
    scores = [a / len(test_users) for a in acc]
    return scores[0] if len(scores) == 1 else scores
