"""
Row-sharded training and catalog-sharded evaluation of the matrix-factorization path over the GPUs
of one node (the reference has no multi-GPU path for these tables, so the semantics
below are those of ONE ``MatrixFactorizationModel`` trained on the concatenation of the ranks'
batches: ``base_pipeline.py:603-634`` for the loss, ``torch.optim.SparseAdam`` / ``Adagrad`` / ``SGD``
on the coalesced sparse gradient for the update, ``metrics.py:285-395`` for evaluation).

Layout.  One process per GPU.  Global row ``r`` of a table lives on rank ``r mod G`` at local row
``r div G`` (round robin: balanced under Zipf-distributed ids).  A rank keeps its user rows and its
item rows in ONE combined table ``emb [U_loc + N_loc, D]`` (user rows first) with a parallel
``bias [U_loc + N_loc]`` and parallel optimizer state, so that a request for rows of both kinds is one
gather and the owner-side optimizer step is one ``collie_sparse_rows_update`` call.

Training step on a rank (``train_step``), for its slice of the global batch:
  1. unique user ids and unique item ids of the slice (positives and all K negatives), ordered by
     (owner, kind, id): that order IS the compact row numbering of the step;
     (``collie_shard_plan``: stamp table + prefix sum, no sort);
  2. all-to-all of the per-peer counts, then of the requested row numbers;
  3. owners gather the rows; all-to-all of embedding rows and bias values back.  The requester now
     holds compact tables ``[n, D]`` / ``[n]`` in its compact numbering;
  4. the fused forward kernel (``collie_mf_train_forward``) runs on the compact tables and compact
     ids; the segment kernels reduce the staged rows into ``[n, D]`` / ``[n]`` gradient buffers
     (``collie_mf_train_update`` with SGD, lr = -1, on zero-filled buffers: ``0 + 1 * g``), scaled by
     ``B_local / B_global`` so that the sum over ranks is the gradient of the global-batch loss;
  5. all-to-all of the gradient rows back to the owners along the same routes (every looked-up row
     travels, zero-gradient rows included: SparseAdam decays their moments);
  6. owners coalesce what the requesters sent (rank order, deterministic) and update their rows.
There is no float atomic and every row has exactly one writer.  The loss returned is the
global-batch loss: all-reduce of ``loss_local * B_local / B_global``.

Evaluation (``evaluate_in_batches``): every rank scores the same group of users against ITS item
rows (``id_stride = G, id_offset = rank`` in the eval kernels), emits its local top-k with global
ids, the scores of the positives it owns and the rank counters over its rows; an all-gather of
G x k candidates per user followed by ``collie_mf_eval_merge_topk``, and two sum all-reduces
(positive scores: exact, every other rank contributes 0.0; integer counters) finish the metrics.

The index plumbing (``torch.unique``, stable ``argsort``, ``index_select``) and the collectives
are torch calls; all floating-point arithmetic is in the HIP library through ``HipEngine``.  The
engine is a constructor argument only so that the world_size-2 ``gloo`` tests can drive the same
exchange code on CPU tensors with the oracle as the arithmetic (tests/sharded_cpu_engine.py);
the product never imports the oracle and ``HipEngine`` refuses CPU tensors.
"""
import math
from typing import Iterable, Optional

import numpy as np
import torch
import torch.distributed as dist

from collie_amd import _lib, ops
from collie_amd import loss as loss_mod

_SPARSE_ALGOS = ('sparse_adam', 'adagrad', 'sgd')
_I31 = 0x7fffffff


# ---- counter-based initialisation -------------------------------------------------------------
def _mix64(z):
    """splitmix64 finaliser on int64 tensors (wrapping arithmetic, logical shifts by masking)."""
    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)
    z = (z ^ lsr(z, 30)) * -4658895280553007687        # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * -7723592293110705685        # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def counter_normal(rows: torch.Tensor, dim: int, seed: int, stream: int, std: float,
                   chunk_rows: int = 1 << 18) -> torch.Tensor:
    """
    ``[len(rows), dim]`` float32 draws of N(0, std^2) that depend only on (seed, stream, global row,
    column): the tables a model starts from are the same for every G and every rank layout
    (replaces the ``torch.manual_seed`` + ``nn.Embedding.weight.normal_`` init of
    ``collie/model/base/layers.py:6-17``, which is tied to one process's generator).
    Box-Muller in float64 on a splitmix64 hash, rounded to float32 at the end.
    """
    dev = rows.device
    out = torch.empty(rows.shape[0], dim, dtype=torch.float32, device=dev)
    cols = torch.arange(dim, dtype=torch.int64, device=dev)
    base = _mix64(torch.tensor([seed * 2 + 1], dtype=torch.int64, device=dev)
                  * -7046029254386353131 + stream)      # 0x9E3779B97F4A7C15
    for s in range(0, rows.shape[0], chunk_rows):
        r = rows[s:s + chunk_rows].to(torch.int64)
        ctr = (r[:, None] * dim + cols[None, :]) * -7046029254386353131 + base
        z = _mix64(ctr)
        u1 = (((z >> 40) & 0xffffff).to(torch.float64) + 0.5) / 16777216.0
        u2 = (((z >> 16) & 0xffffff).to(torch.float64) + 0.5) / 16777216.0
        n = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos((2.0 * math.pi) * u2)
        out[s:s + chunk_rows] = (n * std).to(torch.float32)
    return out


# ---- arithmetic engine ---------------------------------------------------------------------------
class HipEngine:
    """Every floating-point operation of the sharded path, through ``libcollie_hip.so``."""

    name = 'hip'

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.CollieHipError('HipEngine computes on a HIP device only (no CPU fallback)')
        _lib.load()
        self._train_ws = ops.TrainWorkspace(self.device)
        self._rows_ws = None
        self._eval_ws = ops.EvalWorkspace(self.device)
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._acc = ops.optim_cfg('sgd', 1, -1.0)
        self._planner = None
        self.gradient_row_mask = None

    # requester-side plan of one step (collie_shard_plan)
    def plan(self, model, users, pos_items, neg_items):
        if self._planner is None:
            self._planner = ops.ShardPlanner(model.hparams['num_users'], model.hparams['num_items'],
                                             model.comm.world, self.device)
        return self._planner.plan(users, pos_items, neg_items)

    # owner side of the lookup exchange: (emb[rows], bias[rows]) in one launch
    def serve_rows(self, emb, bias, rows):
        return ops.shard_serve_rows(emb, bias, rows, self._status)

    # compact tables + compact ids -> (loss[1], grad rows [n, D], grad bias [n]), both * scale
    def train_grads(self, cemb, cbias, cu, cp, cn, kind, num_items, codes, weights, scale):
        n, D = cemb.shape
        B, K = cn.shape
        cb = cbias.view(n, 1)
        tables = _lib.make_tables(cemb, cemb, cb, cb)       # one table, two id ranges
        batch = ops.make_batch(cu, cp, cn)
        buf = self._train_ws.ensure(B, K, D, n, n)
        step_id = self._train_ws.next_step()
        cfg = ops.make_loss_cfg(kind, num_items, codes, weights)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        ops.train_forward(tables, batch, cfg, buf, True, step_id, loss, None, None, self._status)
        mark = getattr(self, 'phase_mark', None)
        if mark is not None:
            mark('segment_reduce')      # everything before this mark is K1 (+ its loss finalize)
        ops.train_build_segments(tables, batch, buf)
        gemb = torch.zeros(n, D, dtype=torch.float32, device=self.device)
        gbias = torch.zeros(n, dtype=torch.float32, device=self.device)
        gb = gbias.view(n, 1)
        gt = _lib.make_tables(gemb, gemb, gb, gb)
        both = {_lib.SIDE_USER: (self._acc, None, None, None),
                _lib.SIDE_ITEM: (self._acc, None, None, None)}
        ops.train_update(gt, batch, buf, False, False, step_id, both, scale)
        ops.train_update(gt, batch, buf, True, False, step_id, both, scale)
        # rows that own gradient entries: what the compressed gradient exchange sends home
        self.gradient_row_mask = torch.zeros(n, dtype=torch.uint8, device=self.device)
        ops.train_gradient_rows(tables, batch, buf, self.gradient_row_mask, self.gradient_row_mask)
        return loss, gemb, gbias

    # optimizer.step() of one combined shard table on an uncoalesced COO gradient
    def rows_update(self, table, rows, vals, algo, step, lr, state, hyper):
        cfg = ops.optim_cfg(algo, step, lr, hyper.get('beta1', 0.9), hyper.get('beta2', 0.999),
                            hyper['eps'], 0.0, hyper.get('lr_decay', 0.0))
        s1 = state.get('exp_avg', state.get('sum'))
        s2 = state.get('exp_avg_sq')
        self._rows_ws = ops.sparse_rows_update(table, rows, vals, cfg, s1, s2, self._status,
                                               self._rows_ws)

    def rows_update_pair(self, emb, bias, rows, gemb, gbias, algos, step, lrs, states, hypers,
                         val_index=None):
        """Embedding rows and their bias rows share the row list: one coalesce, two updates.
        val_index: row of gemb / gbias that belongs to each entry of ``rows`` (-1: zero gradient)."""
        if val_index is not None and gemb.shape[0] == 0:
            gemb = torch.zeros(1, emb.shape[1], dtype=torch.float32, device=emb.device)
            gbias = torch.zeros(1, dtype=torch.float32, device=emb.device)
        cfgs = [ops.optim_cfg(a, step, lr, h.get('beta1', 0.9), h.get('beta2', 0.999), h['eps'], 0.0,
                              h.get('lr_decay', 0.0)) for a, lr, h in zip(algos, lrs, hypers)]
        st = [(s.get('exp_avg', s.get('sum')), s.get('exp_avg_sq')) for s in states]
        self._rows_ws = ops.sparse_rows_update_pair(emb, bias, rows, gemb, gbias, cfgs[0], cfgs[1],
                                                    st[0], st[1], self._status, self._rows_ws,
                                                    val_index=val_index)

    # ---- catalog-sharded evaluation --------------------------------------------------------------
    def _eval_tables(self, utile, ubias, iemb, ibias):
        return _lib.make_tables(utile, iemb, ubias.view(-1, 1), ibias.view(-1, 1))

    def shard_topk(self, utile, ubias, iemb, ibias, stride, offset, k):
        slots = torch.arange(utile.shape[0], dtype=torch.int64, device=self.device)
        t = self._eval_tables(utile, ubias, iemb, ibias)
        return ops.eval_topk(t, slots, 0, iemb.shape[0], k, None, self._eval_ws, stride, offset)

    def merge_topk(self, cand_scores, cand_ids, k):
        return ops.eval_merge_topk(cand_scores, cand_ids, k)[1]

    def shard_pos_scores(self, utile, ubias, iemb, ibias, stride, offset, indptr, indices):
        slots = torch.arange(utile.shape[0], dtype=torch.int64, device=self.device)
        t = self._eval_tables(utile, ubias, iemb, ibias)
        return ops.eval_pos_scores(t, slots, indptr, indices, None, False, stride, offset)[0]

    def thresholds(self, utile, ubias, iemb, ibias, indptr, indices, pos, y_range):
        slots = torch.arange(utile.shape[0], dtype=torch.int64, device=self.device)
        t = self._eval_tables(utile, ubias, iemb, ibias)
        return ops.eval_pos_scores(t, slots, indptr, indices, y_range, True, given_scores=pos)[1:]

    def shard_rank_counts(self, utile, ubias, iemb, ibias, stride, offset, indptr, thr):
        n = utile.shape[0]
        slots = torch.arange(n, dtype=torch.int64, device=self.device)
        t = self._eval_tables(utile, ubias, iemb, ibias)
        counts = torch.zeros(n, 3, dtype=torch.int64, device=self.device)
        lo, hi, best_s, best_id = thr
        ops.eval_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, 0, iemb.shape[0], counts,
                             stride, offset)
        return counts

    def shard_topk_rank_counts(self, utile, ubias, iemb, ibias, stride, offset, k, indptr, thr):
        """``shard_topk`` and ``shard_rank_counts`` in one pass over the shard."""
        n = utile.shape[0]
        slots = torch.arange(n, dtype=torch.int64, device=self.device)
        t = self._eval_tables(utile, ubias, iemb, ibias)
        counts = torch.zeros(n, 3, dtype=torch.int64, device=self.device)
        lo, hi, best_s, best_id = thr
        top = ops.eval_topk_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, 0,
                                        iemb.shape[0], k, None, counts, self._eval_ws, stride,
                                        offset)
        return top, counts

    def metrics(self, topk_ids, k, n_users, n_items, indptr, indices, pos, counts, y_range, want):
        return ops.eval_metrics(topk_ids, k, n_users, n_items, indptr, indices, pos, counts,
                                y_range, *want)


# ---- collectives ---------------------------------------------------------------------------------
def _cumsum(values):
    total = 0
    for v in values:
        total += v
        yield total


class _Comm:
    """
    The few collectives the path needs; world size 1 needs no process group at all.  Backend
    ``nccl`` (RCCL over xGMI) moves device buffers directly; under ``gloo`` device tensors are staged
    through the host (tests with several ranks on one GPU, hosts without RCCL).
    """

    def __init__(self, group=None):
        self.group = group
        self.staged = False
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
            self.staged = dist.get_backend(group) == 'gloo'
        else:
            self.world, self.rank = 1, 0

    def all_to_all(self, inp, in_splits, out_splits):
        """Rows of ``inp`` split by ``in_splits`` go to the peers; returns the rows received."""
        shape = (int(sum(out_splits)),) + tuple(inp.shape[1:])
        out = torch.empty(shape, dtype=inp.dtype, device=inp.device)
        if self.world == 1:
            out.copy_(inp)
            return out
        if self.staged and inp.is_cuda:
            host = torch.empty(shape, dtype=inp.dtype)
            dist.all_to_all_single(host, inp.cpu().contiguous(), list(out_splits), list(in_splits),
                                   group=self.group)
            return out.copy_(host)
        dist.all_to_all_single(out, inp.contiguous(), list(out_splits), list(in_splits),
                               group=self.group)
        return out

    def all_reduce(self, t):
        if self.world > 1:
            if self.staged and t.is_cuda:
                host = t.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(host)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def gather_to_host(self, t, dst=0):
        """Every rank's ``t`` (same shape) as a list of HOST tensors on rank ``dst``; None on the
        other ranks.  Only ``dst`` ever holds more than its own piece."""
        if self.world == 1:
            return [t.cpu()]
        src = t.cpu() if (self.staged and t.is_cuda) else t.contiguous()
        out = [torch.empty_like(src) for _ in range(self.world)] if self.rank == dst else None
        dist.gather(src, out, dst=dst, group=self.group)
        return [o.cpu() for o in out] if out is not None else None

    def all_gather(self, t):
        if self.world == 1:
            return [t]
        src = t.cpu() if (self.staged and t.is_cuda) else t.contiguous()
        out = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(out, src, group=self.group)
        return [o.to(t.device) for o in out]


def shard_batch(batch, rank, world):
    """The contiguous slice of a global ``((users, pos), neg)`` batch that ``rank`` trains on."""
    (users, pos), neg = batch
    B = users.shape[0]
    lo = (B * rank) // world
    hi = (B * (rank + 1)) // world
    return (users[lo:hi], pos[lo:hi]), neg[lo:hi]


def plan_torch(users, pos_items, neg_items, U, N, G):
    """
    Step 1 of the module docstring with torch index ops (any device; what the gloo tests run).
    Same outputs, bit for bit, as ``collie_shard_plan`` (``HipEngine.plan``).
    """
    dev = users.device
    B, K = neg_items.shape
    uu, inv_u = torch.unique(users.long(), return_inverse=True)
    ui, inv_i = torch.unique(torch.cat([pos_items.long(), neg_items.long().reshape(-1)]),
                             return_inverse=True)
    bad = (uu[0] < 0) | (uu[-1] >= U) | (ui[0] < 0) | (ui[-1] >= N)
    uu, ui = uu.clamp(0, U - 1), ui.clamp(0, N - 1)
    nu = uu.shape[0]
    n = nu + ui.shape[0]
    ids = torch.cat([uu, ui])
    kind = torch.zeros(n, dtype=torch.int64, device=dev)
    kind[nu:] = 1
    owner = ids % G
    route = owner * 2 + kind                     # ids ascend inside a route
    order = torch.argsort(route, stable=True)
    compact = torch.empty(n, dtype=torch.int64, device=dev)
    compact[order] = torch.arange(n, dtype=torch.int64, device=dev)
    u_loc = (U - owner + G - 1) // G             # the owner's user rows come first
    send_rows = (ids // G + kind * u_loc)[order]
    counts = torch.zeros(2 * G + 2, dtype=torch.int64, device=dev)
    counts[:2 * G] = torch.bincount(route, minlength=2 * G)
    counts[2 * G] = n
    counts[2 * G + 1] = bad.to(torch.int64)
    cu = compact[inv_u].to(torch.int32)
    ci = compact[nu + inv_i]
    return (cu, ci[:B].to(torch.int32), ci[B:].reshape(B, K).contiguous(), send_rows, counts)


def _resolve_loss(loss, num_negative_samples):
    """``base_pipeline.py:277-340`` for the implicit losses: adaptive whenever K > 1."""
    many = num_negative_samples > 1
    if loss == 'warp':
        if not many:
            raise ValueError('Cannot use WARP loss with a single negative sample!')
        return 'warp'
    if 'bpr' in loss:
        return 'adaptive_bpr' if many else 'bpr'
    if 'hinge' in loss or loss == 'adaptive':
        return 'adaptive_hinge' if many else 'hinge'
    raise ValueError('{} is not a valid loss function.'.format(loss))


class ShardedMatrixFactorizationModel:
    """
    ``MatrixFactorizationModel`` whose four tables are row-sharded over the ranks of ``group``.

    Hyper-parameter names are the reference's (``matrix_factorization.py:73-94``); the optimizers
    are the sparse ones (``'sparse_adam'``, ``'adagrad'``, ``'sgd'``): dense Adam updates every
    row every step and stays a single-GPU configuration.
    """

    def __init__(self, num_users: int, num_items: int, num_negative_samples: int,
                 embedding_dim: int = 30, lr: float = 1e-3, bias_lr: Optional[float] = 1e-2,
                 optimizer: str = 'sparse_adam', bias_optimizer: Optional[str] = 'sgd',
                 loss: str = 'hinge', metadata_for_loss=None, metadata_for_loss_weights=None,
                 y_range=None, seed: int = 0, device=None, group=None, engine=None,
                 init: bool = True):
        self.comm = _Comm(group)
        G, rank = self.comm.world, self.comm.rank
        self.device = torch.device(device if device is not None else 'cuda')
        self.engine = engine if engine is not None else HipEngine(self.device)
        if bias_optimizer == 'infer' or bias_optimizer is None:
            bias_optimizer = optimizer
        if bias_lr == 'infer' or bias_lr is None:
            bias_lr = lr
        for algo in (optimizer, bias_optimizer):
            if algo not in _SPARSE_ALGOS:
                raise ValueError(f'{algo} is not a valid optimizer for row-sharded tables '
                                 f'(one of {_SPARSE_ALGOS})')
        self.hparams = dict(num_users=int(num_users), num_items=int(num_items),
                            embedding_dim=int(embedding_dim), lr=lr, bias_lr=bias_lr,
                            optimizer=optimizer, bias_optimizer=bias_optimizer, loss=loss,
                            sparse=True, y_range=y_range, seed=seed,
                            num_negative_samples=int(num_negative_samples),
                            metadata_for_loss_weights=metadata_for_loss_weights)
        self._loss_kind = _resolve_loss(loss, num_negative_samples)
        self.U_loc = (num_users - rank + G - 1) // G
        self.N_loc = (num_items - rank + G - 1) // G
        D = embedding_dim
        rows = self.U_loc + self.N_loc
        dev = self.device
        self.emb = torch.empty(rows, D, dtype=torch.float32, device=dev)
        self.bias = torch.zeros(rows, dtype=torch.float32, device=dev)     # ZeroEmbedding
        if init:                                                           # ScaledEmbedding
            gu = torch.arange(rank, num_users, G, dtype=torch.int64, device=dev)
            gi = torch.arange(rank, num_items, G, dtype=torch.int64, device=dev)
            self.emb[:self.U_loc] = counter_normal(gu, D, seed, 0, 1.0 / (2.5 * D))
            self.emb[self.U_loc:] = counter_normal(gi, D, seed, 1, 1.0 / (2.5 * D))
        self.state = {'emb': self._new_state(optimizer, self.emb),
                      'bias': self._new_state(bias_optimizer, self.bias)}
        self.steps = 0
        self._codes, self._weights = loss_mod.metadata_to_codes(
            metadata_for_loss, metadata_for_loss_weights, dev)
        self.last_exchange = {}
        # send home only the gradient rows that are not zero (see _send_nonzero_gradients); with one
        # rank nothing travels, so the dense hand-over stays
        self.compress_gradients = self.comm.world > 1
        self.last_gradient_rows = None
        self._ahead = None          # (batch tensors, plan, ready event) of plan_ahead()
        self._plan_stream = None

    @staticmethod
    def _new_state(algo, table):
        if algo == 'sparse_adam':
            return {'exp_avg': torch.zeros_like(table), 'exp_avg_sq': torch.zeros_like(table)}
        if algo == 'adagrad':
            return {'sum': torch.zeros_like(table)}
        return {}

    _HYPER = {'sparse_adam': {'beta1': 0.9, 'beta2': 0.999, 'eps': 1e-8},
              'adagrad': {'eps': 1e-10, 'lr_decay': 0.0}, 'sgd': {'eps': 0.0}}

    # ---- views in the reference's parameter names (local rows) ---------------------------------
    @property
    def user_embeddings(self):
        return self.emb[:self.U_loc]

    @property
    def item_embeddings(self):
        return self.emb[self.U_loc:]

    @property
    def user_biases(self):
        return self.bias[:self.U_loc]

    @property
    def item_biases(self):
        return self.bias[self.U_loc:]

    # ---- full <-> sharded ------------------------------------------------------------------------
    def load_full_state_dict(self, state_dict):
        """Take this rank's rows out of a reference-format (unsharded) ``state_dict``."""
        G, rank = self.comm.world, self.comm.rank
        dev = self.device
        self.emb[:self.U_loc] = state_dict['user_embeddings.weight'][rank::G].to(dev)
        self.emb[self.U_loc:] = state_dict['item_embeddings.weight'][rank::G].to(dev)
        self.bias[:self.U_loc] = state_dict['user_biases.weight'][rank::G, 0].to(dev)
        self.bias[self.U_loc:] = state_dict['item_biases.weight'][rank::G, 0].to(dev)

    @classmethod
    def from_model(cls, model, num_negative_samples, seed=0, device=None, group=None, engine=None):
        """Shard a single-GPU ``MatrixFactorizationModel`` (same hyper-parameters, same tables)."""
        hp = model.hparams
        self = cls(hp['num_users'], hp['num_items'], num_negative_samples,
                   embedding_dim=hp['embedding_dim'], lr=hp['lr'], bias_lr=hp['bias_lr'],
                   optimizer=model.optimizer, bias_optimizer=model.bias_optimizer,
                   loss=model.loss, metadata_for_loss=hp.get('metadata_for_loss'),
                   metadata_for_loss_weights=hp.get('metadata_for_loss_weights'),
                   y_range=hp.get('y_range'), seed=seed, device=device, group=group,
                   engine=engine, init=False)
        self.load_full_state_dict(model.state_dict())
        return self

    def _gather_rows(self, local, n_global):
        """All ranks' shards of one table, interleaved back into global row order."""
        G = self.comm.world
        per = (n_global + G - 1) // G
        pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
        parts = self.comm.all_gather(pad)
        full = torch.empty((n_global,) + tuple(local.shape[1:]), dtype=local.dtype,
                           device=local.device)
        for r, part in enumerate(parts):
            n_r = (n_global - r + G - 1) // G
            full[r::G] = part[:n_r]
        return full

    def state_dict(self):
        """Reference-format ``state_dict`` with the full tables on EVERY rank's device: for
        moderate sizes and tests; ``save_model`` / ``state_dict_to_rank0`` stream instead."""
        U, N = self.hparams['num_users'], self.hparams['num_items']
        return {
            'user_biases.weight': self._gather_rows(self.user_biases, U).reshape(U, 1),
            'item_biases.weight': self._gather_rows(self.item_biases, N).reshape(N, 1),
            'user_embeddings.weight': self._gather_rows(self.user_embeddings, U),
            'item_embeddings.weight': self._gather_rows(self.item_embeddings, N),
        }

    def _table_to_rank0_host(self, local, n_global, chunk_rows):
        """One table in global row order as a HOST tensor on rank 0 (None elsewhere), streamed in
        chunks of ``chunk_rows`` local rows: no rank ever holds another rank's shard on its GPU and
        only rank 0's host memory holds the full table (100 M x 128 floats = 51 GB)."""
        G, rank = self.comm.world, self.comm.rank
        per = (n_global + G - 1) // G
        full = None
        if rank == 0:
            full = torch.empty((n_global,) + tuple(local.shape[1:]), dtype=local.dtype)
        for lo in range(0, per, chunk_rows):
            hi = min(lo + chunk_rows, per)
            piece = torch.zeros((hi - lo,) + tuple(local.shape[1:]), dtype=local.dtype,
                                device=local.device)
            have = max(0, min(hi, local.shape[0]) - lo)
            if have:
                piece[:have] = local[lo:lo + have]
            parts = self.comm.gather_to_host(piece, dst=0)
            if parts is not None:
                for r, part in enumerate(parts):
                    n_r = (n_global - r + G - 1) // G            # rows rank r owns
                    take = max(0, min(hi, n_r) - lo)
                    if take:
                        full[lo * G + r:(lo + take - 1) * G + r + 1:G] = part[:take]
        return full

    def state_dict_to_rank0(self, chunk_rows: int = 1 << 20):
        """Reference-format ``state_dict`` on rank 0's HOST only (None on the other ranks)."""
        U, N = self.hparams['num_users'], self.hparams['num_items']
        out = {}
        for name, local, n, shape in (
                ('user_biases.weight', self.user_biases, U, (U, 1)),
                ('item_biases.weight', self.item_biases, N, (N, 1)),
                ('user_embeddings.weight', self.user_embeddings, U, None),
                ('item_embeddings.weight', self.item_embeddings, N, None)):
            full = self._table_to_rank0_host(local, n, chunk_rows)
            if full is not None:
                out[name] = full.reshape(shape) if shape is not None else full
        return out if self.comm.rank == 0 else None

    def save_model(self, filename='model.pth', chunk_rows: int = 1 << 20):
        """``base_pipeline.py:880-900`` file format, written by rank 0.  The tables are streamed to
        rank 0's host memory chunk by chunk (``state_dict_to_rank0``): no GPU ever holds more than
        its own shard plus one chunk.  Collective: every rank must call it."""
        sd = self.state_dict_to_rank0(chunk_rows)
        if self.comm.rank == 0:
            hp = {k: v for k, v in self.hparams.items() if k not in ('seed', 'num_negative_samples')}
            hp.update(dropout_p=0.0, weight_decay=0.0, metadata_for_loss=None,
                      load_model_path=None, map_location=None, num_epochs_completed=0,
                      _is_implicit=True)
            torch.save({'state_dict': sd, 'hparams': hp}, str(filename))

    # ---- training ----------------------------------------------------------------------------------
    def profile_phases(self, enable=True):
        """Record a device event at every phase boundary of ``train_step`` (cheap: no sync);
        ``phase_times()`` turns them into mean milliseconds per phase.  The only way to see where a
        multi-GPU step spends its time from inside the benchmark (rocprofv3 aside)."""
        self._phase_events = [] if enable else None
        self.engine.phase_mark = self._phase_mark if enable else None

    def _phase_mark(self, name):
        ev = getattr(self, '_phase_events', None)
        if ev is None or not torch.cuda.is_available() or self.device.type != 'cuda':
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append((name, e))

    def phase_times(self):
        """Mean ms per phase over the recorded steps (synchronises)."""
        ev = getattr(self, '_phase_events', None)
        if not ev:
            return {}
        torch.cuda.synchronize()
        sums, steps = {}, 0
        for (name, a), (_, b) in zip(ev[:-1], ev[1:]):
            if name is None:
                continue
            sums[name] = sums.get(name, 0.0) + a.elapsed_time(b)
            steps += name == 'plan'
        return {k: v / max(steps, 1) for k, v in sums.items()}

    def _plan(self, users, pos_items, neg_items):
        """Steps 1-2 of the module docstring: compact numbering + the request exchange."""
        comm, G = self.comm, self.comm.world
        dev = users.device
        B, K = neg_items.shape
        planner = getattr(self.engine, 'plan', None)
        if planner is not None:
            cu, cp, cn, send_rows, counts = planner(self, users, pos_items, neg_items)
        else:
            cu, cp, cn, send_rows, counts = plan_torch(
                users, pos_items, neg_items, self.hparams['num_users'],
                self.hparams['num_items'], G)
        # one small all-to-all: [users, items, B_local, id range flag] per peer
        meta = torch.empty(G, 4, dtype=torch.int64, device=dev)
        meta[:, :2] = counts[:2 * G].reshape(G, 2)
        meta[:, 2] = B
        meta[:, 3] = counts[2 * G + 1]
        got = comm.all_to_all(meta, [1] * G, [1] * G)
        # The split sizes of the row exchanges must be known to the host: the one read-back of the
        # step.  With plan_ahead() it happens on a side stream while the previous step computes.
        meta_h, got_h = torch.stack([meta, got]).cpu().tolist()
        if any(row[3] for row in got_h):
            raise IndexError('index out of range in self: a user or item id is outside the tables')
        send_splits = [r[0] + r[1] for r in meta_h]
        recv_splits = [r[0] + r[1] for r in got_h]
        n = sum(send_splits)
        B_global = sum(r[2] for r in got_h)
        # rows of the owner's combined table (user rows first, item rows behind them)
        crow = comm.all_to_all(send_rows[:n], send_splits, recv_splits)
        return dict(n=n, B=B, K=K, B_global=B_global, crow=crow, send_splits=send_splits,
                    recv_splits=recv_splits, cu=cu, cp=cp, cn=cn)

    def plan_ahead(self, batch) -> None:
        """
        Optional: plan (steps 1-2) the NEXT ``train_step``'s batch now, on a side stream.  Call it
        right after ``train_step(t)`` returns with the batch of step t + 1: the planner kernels, the
        count exchange and the host read-back of the split sizes then overlap step t's kernels, and
        ``train_step(t + 1)`` enqueues its work without the host ever waiting for the main stream
        (without it the host waits for step t to drain before it can size step t + 1's exchanges:
        the GPU idles for the launch latency of a whole step).  Collective: every rank must call it,
        with its own slice, in the same order as the ``train_step`` calls.  The batch tensors must be
        complete when this is called (a loader's output, or produced before the last host
        synchronisation): the side stream does not wait for work still queued on the main stream --
        waiting for it would serialise the plan behind step t again.
        """
        (users, pos_items), neg_items = batch
        if neg_items.dim() == 1:
            neg_items = neg_items.reshape(-1, 1)
        if self.device.type != 'cuda':
            self._ahead = ((users, pos_items, neg_items), self._plan(users, pos_items, neg_items),
                           None)
            return
        if self._plan_stream is None:
            self._plan_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        side = self._plan_stream
        with torch.cuda.stream(side):
            p = self._plan(users, pos_items, neg_items)
            ready = torch.cuda.Event()
            ready.record(side)
        for t in (p['crow'], p['cu'], p['cp'], p['cn']):
            t.record_stream(main)       # allocated on the side stream, consumed on the main one
        self._ahead = ((users, pos_items, neg_items), p, ready)

    def _take_plan(self, users, pos_items, neg_items):
        ahead, self._ahead = self._ahead, None
        if ahead is not None:
            same = all(a.data_ptr() == b.data_ptr() and a.shape == b.shape
                       for a, b in zip(ahead[0], (users, pos_items, neg_items)))
            _, plan, ready = ahead
            if same:
                if ready is not None:
                    torch.cuda.current_stream(self.device).wait_event(ready)
                return plan
            raise RuntimeError('plan_ahead() was called with a different batch than the next '
                               'train_step()')
        return self._plan(users, pos_items, neg_items)

    def train_step(self, batch) -> torch.Tensor:
        """
        ``calculate_loss`` + ``backward`` + both ``optimizer.step()`` calls for this rank's slice
        ``((users, pos_items), neg_items)`` of the global batch; returns the global-batch loss
        (0-d tensor, identical on every rank).  Collective: every rank must call it.
        """
        (users, pos_items), neg_items = batch
        if neg_items.dim() == 1:
            neg_items = neg_items.reshape(-1, 1)
        comm, eng, dev = self.comm, self.engine, self.device
        D = self.hparams['embedding_dim']
        mark = self._phase_mark
        mark('plan')
        p = self._take_plan(users, pos_items, neg_items)
        n, crow = p['n'], p['crow']
        # 3. owners serve the rows
        mark('lookup_exchange')
        serve = getattr(eng, 'serve_rows', None)
        if serve is not None:
            served_e, served_b = serve(self.emb, self.bias, crow)
        else:
            served_e, served_b = self.emb.index_select(0, crow), self.bias.index_select(0, crow)
        cemb = comm.all_to_all(served_e, p['recv_splits'], p['send_splits'])
        cbias = comm.all_to_all(served_b, p['recv_splits'], p['send_splits'])
        # 4. fused forward + segment reduce on the compact tables
        mark('forward_backward')
        codes = []
        for c in self._codes or []:          # category codes of the compact rows (items only)
            cc = torch.zeros(n, dtype=torch.int32, device=dev)
            cc[p['cp'].long()] = c[pos_items.long()]
            cc[p['cn'].reshape(-1).long()] = c[neg_items.reshape(-1).long()]
            codes.append(cc)
        scale = torch.full((1,), p['B'] / p['B_global'], dtype=torch.float32, device=dev)
        loss, gemb, gbias = eng.train_grads(cemb, cbias, p['cu'], p['cp'], p['cn'],
                                            self._loss_kind, self.hparams['num_items'], codes,
                                            self._weights, scale)
        # 5. gradients home
        mark('gradient_exchange')
        val_index = None
        if self.compress_gradients:
            grows, gb, val_index = self._send_nonzero_gradients(p, gemb, gbias, crow)
        else:
            grows = comm.all_to_all(gemb, p['send_splits'], p['recv_splits'])
            gb = comm.all_to_all(gbias, p['send_splits'], p['recv_splits'])
        # 6. owners update
        mark('owner_update')
        self.steps += 1
        hp = self.hparams
        pair = hasattr(eng, 'rows_update_pair') and D >= 2
        if val_index is not None and not pair:
            # engines without the indexed update: the received rows back in request order
            full = torch.zeros(crow.shape[0], D, dtype=torch.float32, device=dev)
            fullb = torch.zeros(crow.shape[0], dtype=torch.float32, device=dev)
            at = torch.nonzero(val_index >= 0).reshape(-1)
            full[at] = grows[val_index[at].long()]
            fullb[at] = gb[val_index[at].long()]
            grows, gb, val_index = full, fullb, None
        if crow.shape[0] > 0 and pair:
            eng.rows_update_pair(self.emb, self.bias, crow, grows, gb,
                                 (hp['optimizer'], hp['bias_optimizer']), self.steps,
                                 (hp['lr'], hp['bias_lr']), (self.state['emb'], self.state['bias']),
                                 (self._HYPER[hp['optimizer']], self._HYPER[hp['bias_optimizer']]),
                                 val_index=val_index)
        elif crow.shape[0] > 0:
            eng.rows_update(self.emb, crow, grows, hp['optimizer'], self.steps, hp['lr'],
                            self.state['emb'], self._HYPER[hp['optimizer']])
            eng.rows_update(self.bias, crow, gb, hp['bias_optimizer'], self.steps, hp['bias_lr'],
                            self.state['bias'], self._HYPER[hp['bias_optimizer']])
        total = comm.all_reduce(loss * scale)
        mark(None)
        self.last_exchange = dict(rows_requested=n, rows_served=int(crow.shape[0]),
                                  bytes_each_way=n * (D + 1) * 4)
        return total.reshape(())

    def _send_nonzero_gradients(self, p, gemb, gbias, crow):
        """
        Step 5 with only the gradient rows that are not zero on the wire.  Every loss gives at most
        one negative of a sample a gradient, so of the (2 + K) rows a sample looks up only three
        can carry one: with K = 10 three quarters of the dense exchange would be zeros.  The owner
        already knows which rows were looked up (the request lists of step 2: those are the rows
        SparseAdam must touch), so a requester sends, per owner, the positions inside its request
        list whose gradient is not zero, and those rows.  Costs one more count exchange and host
        read-back per step.  Returns (rows [mv, D], bias values [mv], val_index [len(crow)] int32:
        the received row of every served row, -1 = zero gradient).
        """
        comm, G, dev = self.comm, self.comm.world, gemb.device
        n = gemb.shape[0]
        mask = getattr(self.engine, 'gradient_row_mask', None)
        if mask is not None:        # rows with gradient entries (a superset of the non-zero rows)
            nz = mask.bool()
        else:
            nz = (gemb != 0).any(dim=1) | (gbias != 0)
        bounds = torch.tensor([0] + list(_cumsum(p['send_splits'])), dtype=torch.int64, device=dev)
        csum = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nz, 0, out=csum[1:])
        cnt_send = csum[bounds[1:]] - csum[bounds[:-1]]
        cnt_recv = comm.all_to_all(cnt_send, [1] * G, [1] * G)
        send_c, recv_c = torch.stack([cnt_send, cnt_recv]).cpu().tolist()
        idx = torch.nonzero(nz).reshape(-1)               # ascending = grouped by owner
        owner = torch.bucketize(idx, bounds[1:], right=True)
        pos = idx - bounds[owner]                         # position inside that owner's request list
        rpos = comm.all_to_all(pos, send_c, recv_c)
        grows = comm.all_to_all(gemb.index_select(0, idx), send_c, recv_c)
        gb = comm.all_to_all(gbias.index_select(0, idx), send_c, recv_c)
        # served row of every received gradient row: offset of the peer's request list + position
        roff = torch.tensor([0] + list(_cumsum(p['recv_splits']))[:-1], dtype=torch.int64, device=dev)
        peer = torch.repeat_interleave(torch.arange(G, device=dev),
                                       torch.tensor(recv_c, dtype=torch.int64, device=dev))
        served = roff[peer] + rpos
        val_index = torch.full((crow.shape[0],), -1, dtype=torch.int32, device=dev)
        val_index[served] = torch.arange(served.shape[0], dtype=torch.int32, device=dev)
        self.last_gradient_rows = int(idx.shape[0])
        return grows, gb, val_index

    # ---- evaluation --------------------------------------------------------------------------------
    def _user_tile(self, uids):
        """Embedding rows + biases of a group of users on every rank (owners fill, sum all-reduce)."""
        G, rank = self.comm.world, self.comm.rank
        n = uids.shape[0]
        tile = torch.zeros(n, self.hparams['embedding_dim'], dtype=torch.float32,
                           device=self.device)
        tb = torch.zeros(n, dtype=torch.float32, device=self.device)
        own = torch.nonzero((uids % G) == rank).reshape(-1)
        loc = uids[own] // G
        tile[own] = self.emb.index_select(0, loc)
        tb[own] = self.bias.index_select(0, loc)
        self.comm.all_reduce(tile)
        self.comm.all_reduce(tb)
        return tile, tb

    def per_user_metrics(self, uids, indptr, indices, k, want):
        """(ap, rr, auc) per user of one group (device tensors, None where not wanted)."""
        comm, eng = self.comm, self.engine
        G, rank = comm.world, comm.rank
        n = uids.shape[0]
        N = self.hparams['num_items']
        y = self.hparams.get('y_range')
        y = tuple(y) if y is not None else None
        want_ap, want_rr, want_auc = want
        utile, ubias = self._user_tile(uids)
        iemb, ibias = self.item_embeddings, self.item_biases
        topk_ids = pos = counts = shard_top = None
        if callable(indptr):    # lazy CSR rows: MAP@k alone needs them only after the scoring pass
            if want_ap and not (want_rr or want_auc):
                shard_top = eng.shard_topk(utile, ubias, iemb, ibias, G, rank, k)
            indptr, indices = indptr()
        if want_rr or want_auc:
            pos = eng.shard_pos_scores(utile, ubias, iemb, ibias, G, rank, indptr, indices)
            comm.all_reduce(pos)
            thr = eng.thresholds(utile, ubias, iemb, ibias, indptr, indices, pos, y)
            fused = getattr(eng, 'shard_topk_rank_counts', None)
            if want_ap and fused is not None:   # one pass over the shard for both
                shard_top, counts = fused(utile, ubias, iemb, ibias, G, rank, k, indptr, thr)
            else:
                counts = eng.shard_rank_counts(utile, ubias, iemb, ibias, G, rank, indptr, thr)
            comm.all_reduce(counts)
        if want_ap:
            cap = ops.eval_topk_capacity() if eng.name == 'hip' else 16
            s, i = shard_top if shard_top is not None else eng.shard_topk(
                utile, ubias, iemb, ibias, G, rank, k)
            parts_s, parts_i = comm.all_gather(s), comm.all_gather(i.to(torch.int32))
            cs = torch.full((n, G, cap), float('-inf'), dtype=torch.float32, device=self.device)
            ci = torch.full((n, G, cap), _I31, dtype=torch.int32, device=self.device)
            for r in range(G):
                cs[:, r, :k] = parts_s[r]
                ci[:, r, :k] = parts_i[r]
            topk_ids = eng.merge_topk(cs, ci, k)
        return eng.metrics(topk_ids, k, n, N, indptr, indices, pos, counts, y, want)

    def evaluate_in_batches(self, metric_list: Iterable, test_interactions, k: int = 10,
                            batch_size: int = 20, users_per_launch: int = 65536):
        """
        ``collie.metrics.evaluate_in_batches`` (``metrics.py:285-395``) with the catalog sharded
        over the ranks; every rank returns the same numbers.  ``metric_list`` holds the
        ``mapk`` / ``mrr`` / ``auc`` callables of ``collie_amd.metrics``.
        """
        from collie_amd import metrics as M
        metric_list = list(metric_list)
        names = []
        for m in metric_list:
            if m not in M._FAST:
                raise ValueError('the sharded evaluator computes mapk, mrr and auc only')
            names.append(M._FAST[m])
        want = ('ap' in names, 'rr' in names, 'auc' in names)
        n_items = test_interactions.num_items
        if n_items != self.hparams['num_items']:
            raise ValueError('test interactions and model disagree on the number of items')
        if k > n_items:
            raise ValueError(f'Ensure ``k`` ({k}) is less than the number of items ({n_items}):')
        # the smallest shard over ALL ranks (shards differ by one row): every rank raises or none
        # does, so no rank is left waiting in a collective
        min_shard = n_items // self.comm.world
        if want[0] and (k > 16 or k > min_shard):
            raise ValueError(f'k={k} must not exceed 16 or the smallest item shard ({min_shard})')
        test_users = M._unique_users(test_interactions)
        csr = []

        def targets():                  # built on first use, i.e. under the first scoring launch
            if not csr:
                csr.append(test_interactions.mat.tocsr())
            return csr[0]

        if len(test_users) < batch_size:
            batch_size = len(test_users)
        per_user = {nm: [] for nm in ('ap', 'rr', 'auc')}
        for s in range(0, len(test_users), users_per_launch):
            group = test_users[s:s + users_per_launch]
            uids = torch.from_numpy(group.astype(np.int64)).to(self.device)
            ap, rr, au = self.per_user_metrics(
                uids, lambda g=group: M._csr_slice(targets(), g, self.device), None, k, want)
            for nm, v in (('ap', ap), ('rr', rr), ('auc', au)):
                if v is not None:
                    per_user[nm].append(v)
        host = {nm: torch.cat(v).cpu().numpy() for nm, v in per_user.items() if v}
        acc = [M._weighted_batch_sum(nm, host[nm], batch_size) for nm in names]
        scores = [a / len(test_users) for a in acc]
        return scores[0] if len(scores) == 1 else scores
