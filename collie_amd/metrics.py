"""
``evaluate_in_batches`` and the ``mapk`` / ``mrr`` / ``auc`` metric callables
(reference: ``collie/metrics.py``), same names, arguments, return shapes and errors.

Fast route (taken when the model is this package's ``MatrixFactorizationModel`` and every metric
is one of the three callables below): the catalog is scored by the exact-fp32 MFMA kernels of
``mf_eval.hip``; the ``[users, items]`` score matrix of ``get_preds`` (metrics.py:102-107) is never
written, the top-k ids never leave the GPU until the per-user AP / RR / AUC are final, and there
is no scipy round trip.  The per-user values are then aggregated exactly as the reference does:
consecutive slices of ``batch_size`` sorted unique test users, batch mean times batch length,
divided by the number of test users (metrics.py:367-386).

Generic route (any other model object or metric callable): ``get_preds`` materialises the score
matrix through ``model(users, items)`` and each metric callable receives it, as in the reference.
The callables below then reduce that dense matrix -- any catalogue size, any ``k`` -- with the
``collie_preds_*`` HIP kernels (radix-select top-k, streaming rank counts), so no stock torch
arithmetic is involved there either.  ``k`` above the MFMA kernel's 16 register-resident
candidates costs one more scoring pass per 16 entries on the fast route.
"""
import contextlib
import warnings
from functools import partial
from typing import Any, Callable, Iterable, List, Optional

import numpy as np
import torch
from scipy.sparse import csr_matrix

from collie_amd import _lib, ops
from collie_amd.interactions import Interactions

# users scored per launch: 256 user blocks of the MFMA kernel, so that few item chunks per block are
# needed to fill the chip (long item streams per block = rare top-k insertions, few candidate lists)
USERS_PER_LAUNCH = 65536


def _get_user_item_pairs(user_ids, n_items: int, device):
    """``metrics.py:20-74``."""
    if isinstance(n_items, np.int64):
        n_items = n_items.item()
    users = torch.tensor(np.asarray(user_ids), dtype=torch.int64, device=device
                         ).repeat_interleave(n_items)
    items = torch.arange(0, n_items, device=device).repeat(len(user_ids))
    return users, items


def get_preds(model, user_ids, n_items: int, device) -> torch.Tensor:
    """``metrics.py:77-107``: [n_users, n_items] scores through ``model(users, items)``."""
    if isinstance(n_items, np.int64):
        n_items = n_items.item()
    grid = getattr(model, 'score_catalogue', None)
    if grid is not None:
        # a model that can score the users x catalogue grid without expanding it pair by pair
        # (HybridPretrainedModel: the item side of its head once per item)
        scores = grid(user_ids, n_items)
        if scores is not None:
            return scores
    user, item = _get_user_item_pairs(user_ids, n_items, device)
    with torch.no_grad():
        predicted_scores = model(user, item)
    return predicted_scores.view(-1, n_items)


def _unique_users(test_interactions):
    """``np.unique(test_interactions.mat.row)`` (metrics.py:367) by counting instead of sorting."""
    rows = test_interactions.mat.row
    if len(rows) == 0:
        return np.unique(rows)
    return np.flatnonzero(np.bincount(rows, minlength=test_interactions.num_users)).astype(rows.dtype)


def _csr_slice(targets: csr_matrix, user_ids, device):
    sub = targets[np.asarray(user_ids)]
    sub.sort_indices()
    indptr = torch.from_numpy(sub.indptr.astype(np.int64)).to(device)
    indices = torch.from_numpy(sub.indices.astype(np.int64)).to(device)
    return indptr, indices


def _per_user(tables, user_slots, indptr, indices, n_items, k, y_range, want, ws=None,
              item_range=None, split=None):
    """Per-user (ap, rr, auc) device tensors for one launch-sized group of users.  ``indptr`` may
    be a callable returning ``(indptr, indices)``: MAP@k alone needs the CSR rows only after the
    scoring pass, so the host slices them while the GPU scores."""
    want_ap, want_rr, want_auc = want
    dev = user_slots.device
    n = user_slots.shape[0]
    lo_i, hi_i = item_range if item_range is not None else (0, n_items)
    ws = ws or ops.EvalWorkspace(dev)
    topk_ids = pos = counts = None

    def topk():     # split: the item table's bf16 x 3 image (opt-in scoring pass, MAP@k alone)
        if split is not None:
            return ops.eval_topk_bf16x3(tables, split, user_slots, lo_i, hi_i, k, y_range, ws)[1]
        return ops.eval_topk(tables, user_slots, lo_i, hi_i, k, y_range, ws)[1]

    if callable(indptr):
        if want_ap and not (want_rr or want_auc):
            topk_ids = topk()
            want_ap = False             # done; the launch is in flight while the CSR is sliced
        indptr, indices = indptr()
        if topk_ids is not None:
            return ops.eval_metrics(topk_ids, k, n, n_items, indptr, indices, None, None, y_range,
                                    True, False, False)
    if want_rr or want_auc:
        pos, lo, hi, best_s, best_id = ops.eval_pos_scores(tables, user_slots, indptr, indices,
                                                           y_range, True)
        counts = torch.zeros(n, 3, dtype=torch.int64, device=dev)
        if want_ap:     # top-k and rank counts share one pass over the catalogue
            _, topk_ids = ops.eval_topk_rank_counts(tables, user_slots, indptr, lo, hi, best_s,
                                                    best_id, lo_i, hi_i, k, y_range, counts, ws)
        else:
            ops.eval_rank_counts(tables, user_slots, indptr, lo, hi, best_s, best_id, lo_i, hi_i,
                                 counts)
    elif want_ap:
        topk_ids = topk()
    return ops.eval_metrics(topk_ids, k, n, n_items, indptr, indices, pos, counts, y_range,
                            want_ap, want_rr, want_auc)


def _check_k(k, n_items):
    if k > n_items:
        raise ValueError(f'Ensure ``k`` ({k}) is less than the number of items ({n_items}):')


def _dense_metric(which, targets, user_ids, preds, k):
    """One metric over a materialised ``preds [users, items]`` (any catalogue size, any ``k``):
    the ``collie_preds_*`` kernels stream the rows once per pass; nothing is sorted."""
    _lib.require_hip(preds, 'preds')
    n, N = preds.shape
    indptr, indices = _csr_slice(targets, user_ids, preds.device)
    if which == 'ap':
        _, ids = ops.preds_topk(preds, k)
        return ops.eval_metrics(ids, k, n, N, indptr, indices, None, None, None,
                                True, False, False)
    pos, lo, hi, best_s, best_id = ops.preds_pos_scores(preds, indptr, indices)
    counts = torch.zeros(n, 3, dtype=torch.int64, device=preds.device)
    ops.preds_rank_counts(preds, indptr, lo, hi, best_s, best_id, counts)
    return ops.eval_metrics(None, 1, n, N, indptr, indices, pos, counts, None,
                            False, which == 'rr', which == 'auc')


def mapk(targets: csr_matrix, user_ids, preds, k: int = 10) -> float:
    """``metrics.py:144-198``: mean average precision at k (float64 math, NaN -> 0)."""
    _check_k(k, preds.shape[1])
    ap, _, _ = _dense_metric('ap', targets, user_ids, preds, k)
    return ap.cpu().numpy().mean().item()


def mrr(targets: csr_matrix, user_ids, preds, k: Optional[Any] = None) -> float:
    """``metrics.py:201-240``: mean reciprocal rank of the first relevant item (fp32 mean)."""
    _, rr, _ = _dense_metric('rr', targets, user_ids, preds, 1)
    return rr.cpu().numpy().mean(dtype=np.float32).item()


def auc(targets: csr_matrix, user_ids, preds, k: Optional[Any] = None) -> float:
    """``metrics.py:243-282``: mean per-user AUROC of ``sigmoid(preds)`` (tie-averaged)."""
    _, _, a = _dense_metric('auc', targets, user_ids, preds, 1)
    return a.cpu().numpy().astype(np.float32).mean(dtype=np.float32).item()


_FAST = {mapk: 'ap', mrr: 'rr', auc: 'auc'}


def _batch_mean(name, values):
    """The reduction each reference metric applies to one batch of users."""
    if name == 'ap':
        return float(values.mean())                                   # float64 (:195-198)
    if name == 'rr':
        return float(values.astype(np.float32).mean(dtype=np.float32))  # fp32 (:237-240)
    return float(values.astype(np.float32).mean(dtype=np.float32))      # fp32 tensors (:280-282)


def _weighted_batch_sum(name, values, batch_size):
    """``sum_b metric(batch b) * len(batch b)`` over consecutive batches of ``batch_size`` users,
    i.e. the accumulation of ``metrics.py:379-384``, without a Python iteration per batch: row means
    of the [batches, batch_size] view (numpy reduces each contiguous row exactly as it reduces the
    1-d slice ``_batch_mean`` sees), scaled in float64 and added left to right (``cumsum``) like the
    reference's ``+=``.  Bit-identical to the loop (tested)."""
    n = len(values)
    nb = n // batch_size
    total = 0.0
    if nb > 0:
        head = values[:nb * batch_size]
        if name == 'ap':
            means = head.reshape(nb, batch_size).mean(axis=1)
        else:
            means = head.astype(np.float32).reshape(nb, batch_size).mean(axis=1, dtype=np.float32)
        total = float(np.cumsum(means.astype(np.float64) * batch_size)[-1])
    if nb * batch_size < n:
        tail = values[nb * batch_size:]
        total += _batch_mean(name, tail) * len(tail)
    return total


def evaluate_in_batches(
    metric_list: Iterable[Callable[..., float]],
    test_interactions: Interactions,
    model,
    k: int = 10,
    batch_size: int = 20,
    logger=None,
    verbose: bool = True,
) -> List[float]:
    """``metrics.py:285-395``."""
    if not isinstance(test_interactions, Interactions):
        raise ValueError(
            '``test_interactions`` must be of type ``Interactions``, not '
            f'{type(test_interactions)}. Try using ``explicit_evaluate_in_batches`` instead.')
    metric_list = list(metric_list)
    device = _get_evaluate_in_batches_device(model=model)
    model.to(device)
    model._move_any_external_data_to_device()

    test_users = _unique_users(test_interactions)
    n_items = test_interactions.num_items
    if len(test_users) < batch_size:
        batch_size = len(test_users)

    from collie_amd.model import MatrixFactorizationModel
    fast = (isinstance(model, MatrixFactorizationModel) and model._mf_eval_fast
            and all(m in _FAST for m in metric_list)
            and model.hparams.embedding_dim <= 256)

    if fast:
        _check_k(k, n_items)
        names = [_FAST[m] for m in metric_list]
        want = ('ap' in names, 'rr' in names, 'auc' in names)
        was_training = model.training
        model.eval()
        tables = model._tables()
        y = model.hparams.get('y_range')
        y = tuple(y) if y is not None else None
        ws = ops.EvalWorkspace(torch.device(device) if not isinstance(device, torch.device) else device)
        # opt-in (COLLIE_EVAL_BF16X3=1 or model.eval_bf16x3 = True): MAP@k alone ranks the catalogue
        # on bf16 MFMA (three-way split) and re-scores 16 candidates per user exactly
        split = None
        if (want == (True, False, False) and _bf16x3_wanted(model)
                and ops.eval_bf16x3_ok(model.hparams.embedding_dim, k)):
            split = ops.eval_split_items(model.item_embeddings.weight.data)
        per_user = {n: [] for n in ('ap', 'rr', 'auc')}
        csr = []                        # test_interactions.mat.tocsr(), built on first use

        def csr_rows(group, dev):
            if not csr:
                csr.append(test_interactions.mat.tocsr())
            return _csr_slice(csr[0], group, dev)

        with torch.no_grad():
            for s in range(0, len(test_users), USERS_PER_LAUNCH):
                group = test_users[s:s + USERS_PER_LAUNCH]
                slots = torch.from_numpy(group.astype(np.int64)).to(tables_device(model))
                ap, rr, au = _per_user(tables, slots, partial(csr_rows, group, slots.device), None,
                                       n_items, k, y, want, ws, split=split)
                for n, v in (('ap', ap), ('rr', rr), ('auc', au)):
                    if v is not None:
                        per_user[n].append(v)
        host = {n: torch.cat(v).cpu().numpy() for n, v in per_user.items() if v}
        if was_training:
            model.train()
        accumulators = [_weighted_batch_sum(name, host[name], batch_size) for name in names]
    else:
        targets = test_interactions.mat.tocsr()
        accumulators = [0] * len(metric_list)
        # nothing changes the model inside this loop: a model whose get_preds route has a
        # per-catalogue part (HybridPretrainedModel.score_catalogue) keeps it between batches
        scope = getattr(model, 'catalogue_scope', None)
        with (scope() if scope is not None else contextlib.nullcontext()):
            for i in range(int(np.ceil(len(test_users) / batch_size))):
                user_range = test_users[i * batch_size:(i + 1) * batch_size]
                preds = get_preds(model, user_range, n_items, device)
                for j, metric in enumerate(metric_list):
                    score = metric(targets=targets, user_ids=user_range, preds=preds, k=k)
                    accumulators[j] += (score * len(user_range))

    all_scores = [acc_score / len(test_users) for acc_score in accumulators]
    if logger is not None:
        _log_metrics(model=model, logger=logger, metric_list=metric_list, all_scores=all_scores,
                     verbose=verbose)
    return all_scores[0] if len(all_scores) == 1 else all_scores


def _bf16x3_wanted(model):
    import os
    return bool(getattr(model, 'eval_bf16x3', False)) or os.environ.get('COLLIE_EVAL_BF16X3') == '1'


def tables_device(model):
    return model.user_embeddings.weight.device


def _get_evaluate_in_batches_device(model):
    """``metrics.py:505-521``."""
    device = getattr(model, 'device', None)
    if torch.cuda.is_available() and str(device) == 'cpu':
        warnings.warn('CUDA available but model device is set to CPU - is this desired?')
    if device is None:
        if torch.cuda.is_available():
            warnings.warn('``model.device`` attribute is ``None``. Since GPU is available, putting '
                          'model on GPU.')
            device = 'cuda:0'
        else:
            device = 'cpu'
    return device


def _log_metrics(model, logger, metric_list, all_scores, verbose):
    """``metrics.py:524-543``."""
    try:
        step = model.hparams.get('num_epochs_completed')
    except AttributeError:
        step = None
    try:
        metrics_dict = dict(zip([x.__name__ for x in metric_list], all_scores))
    except AttributeError:
        metrics_dict = dict(zip([type(x).__name__ for x in metric_list], all_scores))
    if verbose:
        print(f'Logging metrics {metrics_dict} to ``logger``...')
    logger.log_metrics(metrics=metrics_dict, step=step)
