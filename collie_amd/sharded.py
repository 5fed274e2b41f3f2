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
gather and the owner-side optimizer step is one ``collie_shard_owner_update`` call.

Training step on a rank (``train_step``), for its slice of the global batch:
  1. distinct user rows and item rows of the slice (positives and all K negatives), ordered by
     (owner, kind, id): that order IS the compact row numbering of the step
     (``collie_shard_plan``: marks + popcount ranks when the batch is dense in the tables, one
     stable radix sort of the (2 + K) B ids otherwise);
  2. all-to-all of the per-peer counts (the step's first host read: ``train_step(next_batch=)``
     queues both inside the previous step, ``plan_ahead`` on a side stream after it), then of the
     requested row numbers;
  3. owners write the requested rows as the peers' blocks of ONE lookup buffer -- embedding rows,
     then the bias values packed into rows of the same width (``collie_shard_serve_rows``, which
     also stamps the served item rows) -- and one all-to-all with row splits brings them back: the
     received buffer IS the requester's compact embedding table (compact numbers skip the bias
     rows between two owners' blocks);
  4. the fused forward kernel (``collie_mf_train_forward``) runs on the compact table and compact
     ids; ``collie_shard_export_grads`` coalesces its staged gradient rows (users, positives and the
     one negative per sample the loss selected: at most 3 of the 2 + K rows a sample looks up) and
     writes them, scaled by ``B_local / B_global``, straight into the owners' blocks of the gradient
     buffer (rows, bias gradients, owner row numbers);
  5. all-to-all of the per-owner gradient counts (second host read; the update of local user rows
     and the next step's plan are queued behind its copy, so the GPU works while the host waits),
     then ONE all-to-all of the gradient buffer;
  6. owners coalesce what the requesters sent (stable sort by row: rank order, deterministic) and
     update: rows with an entry get their summed gradient, and -- SparseAdam -- every item row
     stamped in step 3 without an entry has its moments decayed (looked-up
     rows are touched even with a zero gradient) (``collie_shard_owner_update``).
There is no float atomic and every row has exactly one writer.  The loss returned is the
global-batch loss: all-reduce of ``loss_local * B_local / B_global``.

Evaluation (``evaluate_in_batches``): every rank scores the same group of users against ITS item
rows (``id_stride = G, id_offset = rank`` in the eval kernels), emits its local top-k with global
ids, the scores of the positives it owns and the rank counters over its rows; an all-gather of
G x k candidates per user followed by ``collie_mf_eval_merge_topk``, and two sum all-reduces
(positive scores: exact, every other rank contributes 0.0; integer counters) finish the metrics.

The collectives are torch calls; the planning, packing and all floating-point arithmetic are in
the HIP library through ``HipEngine``.  The
engine is a constructor argument only so that the world_size-2 ``gloo`` tests can drive the same
exchange code on CPU tensors with the oracle as the arithmetic (tests/sharded_cpu_engine.py);
the product never imports the oracle and ``HipEngine`` refuses CPU tensors.
"""
import math
import os
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
        self._export_ws = None
        self._owner_ws = None
        self._eval_ws = ops.EvalWorkspace(self.device)
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._planner = None
        self._touch = None          # owner-side stamps of one model (this engine serves one model)
        self._row_seg = None

    # requester-side plan of one step (collie_shard_plan)
    def plan(self, model, users, pos_items, neg_items, users_local=False):
        if self._planner is None:
            hp = model.hparams
            self._planner = ops.ShardPlanner(hp['num_users'], hp['num_items'], model.comm.world,
                                             hp['embedding_dim'], self.device)
        return self._planner.plan(users, pos_items, neg_items,
                                  model.comm.rank if users_local else -1)

    def _owner_state(self, model):
        if self._touch is None:
            self._touch = torch.zeros(max(model.N_loc, 1), dtype=torch.uint8, device=self.device)
            self._row_seg = torch.zeros(model.emb.shape[0], dtype=torch.int32, device=self.device)
        return self._touch, self._row_seg

    # owner side of the lookup exchange: the peers' blocks of the lookup buffer in one launch
    def serve(self, model, rows, peer_counts, step):
        touch, _ = self._owner_state(model)
        return ops.shard_serve_rows(model.emb, model.bias, rows, peer_counts, touch, model.U_loc,
                                    step, self._status)

    # received lookup buffer + compact ids -> (loss[1], gradient wire buffer, per-owner counts)
    def forward_export(self, model, cw, lookup_counts, send_rows, cu, cp, cn, kind, num_items, codes,
                       weights, scale, users_local=False, step=None):
        R, D = cw.shape
        B, K = cn.shape
        cbias = ops.shard_unpack_bias(cw, lookup_counts).view(R, 1)
        if users_local:     # user rows straight out of this rank's shard, item rows compact
            nu = model.U_loc
            tables = _lib.make_tables(model.emb[:nu], cw, model.bias[:nu].view(nu, 1), cbias)
        else:
            nu = R
            tables = _lib.make_tables(cw, cw, cbias, cbias)     # one table, two id ranges
        batch = ops.make_batch(cu, cp, cn)
        buf = self._train_ws.ensure(B, K, D, nu, R)
        step_id = self._train_ws.next_step()
        cfg = ops.make_loss_cfg(kind, num_items, codes, weights)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        ops.train_forward(tables, batch, cfg, buf, 1 | ops.NO_STAMPS, step_id, loss, None, None,
                          self._status)
        mark = getattr(self, 'phase_mark', None)
        if mark is not None:
            mark('gradient_export')     # everything before this mark is K1 (+ its loss finalize)
        wire, counts, self._export_ws = ops.shard_export_grads(
            tables, batch, buf, lookup_counts, send_rows, scale, self._export_ws, users_local)
        if not users_local:
            return loss, wire, counts

        def local_user_rows():
            # the user rows never left: their update runs on the segments the export just built
            # (the user-bias gradient of a pairwise loss is exactly zero: nothing to do there).
            # Handed back as a callable: train_step launches it AFTER the copy of the gradient counts
            # to the host has been queued, so that the host's wait for the counts ends while this
            # kernel still runs and the GPU never idles for the read.
            hp = model.hparams
            h = model._HYPER[hp['optimizer']]
            ocfg = ops.optim_cfg(hp['optimizer'], step, hp['lr'], h.get('beta1', 0.9),
                                 h.get('beta2', 0.999), h['eps'], 0.0, h.get('lr_decay', 0.0))
            st = model.state['emb']
            s1, s2 = st.get('exp_avg', st.get('sum')), st.get('exp_avg_sq')
            ops.train_update(tables, batch, buf, 0, False, step_id,
                             {_lib.SIDE_USER: (ocfg, s1[:nu] if s1 is not None else None,
                                               s2[:nu] if s2 is not None else None, None)}, scale)
        return loss, wire, counts, local_user_rows

    # optimizer.step() of the combined shard table on the gradient blocks the peers sent
    def owner_update(self, model, wire, peer_counts, step):
        hp = model.hparams
        touch, row_seg = self._owner_state(model)
        cfgs = []
        for algo, lr in ((hp['optimizer'], hp['lr']), (hp['bias_optimizer'], hp['bias_lr'])):
            h = model._HYPER[algo]
            cfgs.append(ops.optim_cfg(algo, step, lr, h.get('beta1', 0.9), h.get('beta2', 0.999),
                                      h['eps'], 0.0, h.get('lr_decay', 0.0)))
        st = [(s_.get('exp_avg', s_.get('sum')), s_.get('exp_avg_sq'))
              for s_ in (model.state['emb'], model.state['bias'])]
        self._owner_ws = ops.shard_owner_update(model.emb, model.bias, model.U_loc, wire,
                                                peer_counts, touch, row_seg, cfgs[0], cfgs[1],
                                                st[0], st[1], self._status, self._owner_ws)

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
        # one rank: nothing travels, the "received" buffer is the one sent.  COLLIE_SHARDED_COLLECTIVES=1
        # keeps the collective calls anyway (tools/sharded_nccl_selftest.py: argument dtypes,
        # contiguity and split lists of every call against RCCL on one GPU)
        self.local = self.world == 1 and not (
            dist.is_available() and dist.is_initialized()
            and os.environ.get('COLLIE_SHARDED_COLLECTIVES') == '1')

    def all_to_all(self, inp, in_splits, out_splits):
        """Rows of ``inp`` split by ``in_splits`` go to the peers; returns the rows received."""
        if self.local:
            return inp
        shape = (int(sum(out_splits)),) + tuple(inp.shape[1:])
        out = torch.empty(shape, dtype=inp.dtype, device=inp.device)
        if self.staged and inp.is_cuda:
            host = torch.empty(shape, dtype=inp.dtype)
            dist.all_to_all_single(host, inp.cpu().contiguous(), list(out_splits), list(in_splits),
                                   group=self.group)
            return out.copy_(host)
        dist.all_to_all_single(out, inp.contiguous(), list(out_splits), list(in_splits),
                               group=self.group)
        return out

    def all_reduce(self, t):
        if not self.local:
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
        if self.local:
            return [t.cpu()]
        src = t.cpu() if (self.staged and t.is_cuda) else t.contiguous()
        out = [torch.empty_like(src) for _ in range(self.world)] if self.rank == dst else None
        dist.gather(src, out, dst=dst, group=self.group)
        return [o.cpu() for o in out] if out is not None else None

    def all_gather(self, t):
        if self.local:
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


def route_batch_to_user_owners(batch, group=None):
    """
    Redistributes the ranks' slices of a global batch so that every sample sits on the rank that
    owns its user row (``user % world``): the slice this function returns can be trained with
    ``train_step(..., users_local=True)``.  Index plumbing only (96 bytes per sample at K = 10):
    one count exchange (a host read) and one all-to-all per id tensor.  Samples keep their order
    inside (source rank, original position); the loss and the updates of a step do not depend on
    which rank computes a sample, only the floating-point summation order of a row's gradient does.
    """
    comm = _Comm(group)
    (users, pos_items), neg_items = batch
    if comm.world == 1:
        return batch
    G = comm.world
    owner = users.long() % G
    order = torch.argsort(owner, stable=True)
    counts = torch.bincount(owner, minlength=G)
    got = comm.all_to_all(counts, [1] * G, [1] * G)
    send_c, recv_c = torch.stack([counts, got]).cpu().tolist()
    out = [comm.all_to_all(t.index_select(0, order).contiguous(), send_c, recv_c)
           for t in (users, pos_items, neg_items)]
    return (out[0], out[1]), out[2]


def wire_rows(count, dim, gradient=False):
    """Rows of one peer's block of an exchange buffer (csrc/shard_wire.h): ``count`` embedding (or
    gradient) rows, ``ceil(count / dim)`` rows of bias values and, in the gradient direction, as
    many rows again of owner row numbers."""
    return count + (2 if gradient else 1) * ((count + dim - 1) // dim)


def plan_torch(users, pos_items, neg_items, U, N, G, dim, local_rank=-1):
    """
    Step 1 of the module docstring with torch index ops (any device; what the gloo tests run).
    Same outputs, bit for bit, as ``collie_shard_plan`` (``HipEngine.plan``).  ``local_rank >= 0``:
    this rank owns every user of the batch, only the items are planned.
    """
    dev = users.device
    B, K = neg_items.shape
    local_users = None
    if local_rank >= 0:
        ul = users.long().clamp(0, U - 1)
        foreign = ((users.long() != ul) | (ul % G != local_rank))
        local_users = torch.where(foreign, torch.zeros_like(ul), ul // G).to(torch.int32)
        users = users[:0]
    uu, inv_u = torch.unique(users.long(), return_inverse=True)
    ui, inv_i = torch.unique(torch.cat([pos_items.long(), neg_items.long().reshape(-1)]),
                             return_inverse=True)
    bad = (ui[0] < 0) | (ui[-1] >= N)
    if uu.shape[0]:
        bad = bad | (uu[0] < 0) | (uu[-1] >= U)
    if local_users is not None:
        bad = bad | foreign.any()
    # an id outside the tables stands for the nearest valid row (and is reported): merge it with
    # that row if the batch also holds it
    uu, ui = uu.clamp(0, U - 1), ui.clamp(0, N - 1)
    uu, inv_uu = torch.unique(uu, return_inverse=True)
    ui, inv_ii = torch.unique(ui, return_inverse=True)
    inv_u, inv_i = inv_uu[inv_u], inv_ii[inv_i]
    nu = uu.shape[0]
    n = nu + ui.shape[0]
    ids = torch.cat([uu, ui])
    kind = torch.zeros(n, dtype=torch.int64, device=dev)
    kind[nu:] = 1
    owner = ids % G
    route = owner * 2 + kind                     # ids ascend inside a route
    order = torch.argsort(route, stable=True)
    seg = torch.empty(n, dtype=torch.int64, device=dev)
    seg[order] = torch.arange(n, dtype=torch.int64, device=dev)
    u_loc = (U - owner + G - 1) // G             # the owner's user rows come first
    send_rows = (ids // G + kind * u_loc)[order]
    counts = torch.zeros(2 * G + 2, dtype=torch.int64, device=dev)
    counts[:2 * G] = torch.bincount(route, minlength=2 * G)
    counts[2 * G] = n
    counts[2 * G + 1] = bad.to(torch.int64)
    # compact row = row of the lookup buffer: skip the bias rows of the owners before this one
    per_owner = counts[:2 * G].reshape(G, 2).sum(1)
    pad = (per_owner + dim - 1) // dim
    shift = torch.cumsum(pad, 0) - pad
    compact = seg + shift[owner]
    cu = compact[inv_u].to(torch.int32) if local_users is None else local_users
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
        self._ahead = None          # (batch tensors, plan, ready event) of plan_ahead()
        self._plan_stream = None
        self._pinned = {}           # small page-locked host buffers of the two count reads

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

    def _host_read(self, name, dev_tensor):
        """Queue the copy of a small device tensor to the host; returns a function that waits for
        THAT copy (not for the stream) and returns the values as nested lists."""
        if dev_tensor.device.type != 'cuda':
            return dev_tensor.tolist
        pin = self._pinned.get(name)
        if pin is None or pin.shape != dev_tensor.shape:
            pin = self._pinned[name] = torch.empty(dev_tensor.shape, dtype=dev_tensor.dtype,
                                                   pin_memory=True)
        pin.copy_(dev_tensor, non_blocking=True)
        done = torch.cuda.Event()
        done.record()

        def read():
            done.synchronize()
            return pin.tolist()
        return read

    def _plan_start(self, users, pos_items, neg_items, users_local=False):
        """Steps 1-2 of the module docstring up to the host read: compact numbering, the exchange
        of the counts, and their copy to the host queued (``_plan_finish`` waits for it)."""
        comm, G = self.comm, self.comm.world
        dev = users.device
        B, K = neg_items.shape
        planner = getattr(self.engine, 'plan', None)
        if B == 0:
            # a rank without samples in this step (a routed batch can leave one empty) requests
            # nothing; it still serves rows, receives gradients and joins every collective
            cu = cp = torch.empty(0, dtype=torch.int32, device=dev)
            cn = torch.empty(0, K, dtype=torch.int64, device=dev)
            send_rows = torch.empty(0, dtype=torch.int64, device=dev)
            counts = torch.zeros(2 * G + 2, dtype=torch.int64, device=dev)
        elif planner is not None:
            cu, cp, cn, send_rows, counts = planner(self, users, pos_items, neg_items, users_local)
        else:
            cu, cp, cn, send_rows, counts = plan_torch(
                users, pos_items, neg_items, self.hparams['num_users'],
                self.hparams['num_items'], G, self.hparams['embedding_dim'],
                comm.rank if users_local else -1)
        # one small all-to-all: [users, items, B_local, id range flag] per peer
        meta = torch.empty(G, 4, dtype=torch.int64, device=dev)
        meta[:, :2] = counts[:2 * G].reshape(G, 2)
        meta[:, 2] = B
        meta[:, 3] = counts[2 * G + 1]
        got = comm.all_to_all(meta, [1] * G, [1] * G)
        # The split sizes of the row exchanges must be known to the host: the first read-back of
        # the step.  Planned inside the previous step (train_step(next_batch=...)) or on a side
        # stream (plan_ahead()), it completes while the previous step still computes.
        return dict(B=B, K=K, cu=cu, cp=cp, cn=cn, send_rows=send_rows, users_local=users_local,
                    read=self._host_read('plan', torch.stack([meta, got])))

    def _plan_finish(self, p):
        """The host read of a started plan + the exchange of the requested row numbers."""
        read = p.pop('read', None)
        if read is None:
            return p
        meta_h, got_h = read()
        if any(row[3] for row in got_h):
            raise IndexError('index out of range in self: a user or item id is outside the tables'
                             + (' (or, with users_local, a user is owned by another rank)'
                                if p['users_local'] else ''))
        send_counts = [r[0] + r[1] for r in meta_h]      # rows this rank requests from each peer
        recv_counts = [r[0] + r[1] for r in got_h]       # rows each peer requests from this rank
        n = sum(send_counts)
        # rows of the owner's combined table (user rows first, item rows behind them)
        crow = self.comm.all_to_all(p['send_rows'][:n], send_counts, recv_counts)
        p.update(n=n, B_global=sum(r[2] for r in got_h), crow=crow, send_counts=send_counts,
                 recv_counts=recv_counts)
        return p

    def _plan(self, users, pos_items, neg_items, users_local=False):
        """Steps 1-2 of the module docstring: compact numbering + the request exchange."""
        return self._plan_finish(self._plan_start(users, pos_items, neg_items, users_local))

    def plan_ahead(self, batch, users_local=False) -> None:
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
            self._ahead = ((users, pos_items, neg_items),
                           self._plan(users, pos_items, neg_items, users_local), None)
            return
        if self._plan_stream is None:
            self._plan_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        side = self._plan_stream
        with torch.cuda.stream(side):
            p = self._plan(users, pos_items, neg_items, users_local)
            ready = torch.cuda.Event()
            ready.record(side)
        for t in (p['crow'], p['cu'], p['cp'], p['cn'], p['send_rows']):
            t.record_stream(main)       # allocated on the side stream, consumed on the main one
        self._ahead = ((users, pos_items, neg_items), p, ready)

    def _take_plan(self, users, pos_items, neg_items, users_local=False):
        ahead, self._ahead = self._ahead, None
        if ahead is not None:
            same = all(a.data_ptr() == b.data_ptr() and a.shape == b.shape
                       for a, b in zip(ahead[0], (users, pos_items, neg_items)))
            same = same and ahead[1]['users_local'] == users_local
            _, plan, ready = ahead
            if same:
                if ready is not None:
                    torch.cuda.current_stream(self.device).wait_event(ready)
                return self._plan_finish(plan)      # (a plan started by train_step(next_batch=))
            raise RuntimeError('plan_ahead() was called with a different batch than the next '
                               'train_step()')
        return self._plan(users, pos_items, neg_items, users_local)

    def train_step(self, batch, users_local=False, next_batch=None) -> torch.Tensor:
        """
        ``calculate_loss`` + ``backward`` + both ``optimizer.step()`` calls for this rank's slice
        ``((users, pos_items), neg_items)`` of the global batch; returns the global-batch loss
        (0-d tensor, identical on every rank).  Collective: every rank must call it.

        ``users_local=True``: the caller guarantees that this rank owns every user of its slice
        (``user % world == rank``: a loader that hands each rank the interactions of its own users,
        or ``route_batch_to_user_owners``).  User rows are then read and updated in place -- one
        of the 2 + K rows a sample looks up and one of its at most 3 gradient rows never travel
        -- and only item rows are exchanged.  Checked on the device: a foreign user raises.

        ``next_batch``: this rank's slice of the NEXT step, if the caller has it already (a loader
        that runs one batch ahead).  Its plan -- which depends on ids only -- is then queued INSIDE
        this step, on the same stream, in front of the owner update: the host's read of its counts
        completes while the owner update still runs, the next ``train_step`` (which must be called
        with that batch and the same ``users_local``) finds its plan ready, and the GPU never waits
        for the host at a step boundary.  Unlike ``plan_ahead()`` nothing runs concurrently, so the
        bandwidth-bound kernels of the step are not slowed down by the planner.  Collective like the
        step itself: every rank passes a next batch, or none does.
        """
        (users, pos_items), neg_items = batch
        if neg_items.dim() == 1:
            neg_items = neg_items.reshape(-1, 1)
        comm, eng, dev = self.comm, self.engine, self.device
        G = comm.world
        D = self.hparams['embedding_dim']
        mark = self._phase_mark
        mark('plan')
        p = self._take_plan(users, pos_items, neg_items, users_local)
        step = self.steps + 1
        # 3. owners serve the rows: one buffer, one all-to-all
        mark('lookup_exchange')
        served = eng.serve(self, p['crow'], p['recv_counts'], step)
        cw = comm.all_to_all(served, [wire_rows(c, D) for c in p['recv_counts']],
                             [wire_rows(c, D) for c in p['send_counts']])
        # 4. fused forward on the compact table + gradient rows into the owners' blocks
        mark('forward')
        scale = torch.full((1,), p['B'] / max(p['B_global'], 1), dtype=torch.float32, device=dev)
        local_rows = None
        if p['B'] == 0:
            loss = torch.zeros(1, dtype=torch.float32, device=dev)
            gwire = torch.empty(0, D, dtype=torch.float32, device=dev)
            gcounts = torch.zeros(G + 1, dtype=torch.int64, device=dev)
        else:
            loss, gwire, gcounts, *rest = self._forward_export(p, cw, pos_items, neg_items, scale,
                                                               users_local, step)
            local_rows = rest[0] if rest else None
        # 5. gradients home: how many rows each owner gets depends on the forward pass (which
        # negative each sample selected), so the counts are exchanged and read back first -- the
        # step's second host read.  Its copy is queued BEFORE the work that does not depend on it
        # (the update of the local user rows, the plan of the next batch), and the host waits for
        # the copy alone.
        mark('gradient_exchange')
        cnt_send = gcounts[:G]
        cnt_recv = comm.all_to_all(cnt_send, [1] * G, [1] * G)
        read_counts = self._host_read('grad', torch.stack([cnt_send, cnt_recv]))
        if local_rows is not None:
            mark('local_user_rows')
            local_rows()
        if next_batch is not None:
            mark('plan_next')
            (nu_, np_), nn_ = next_batch
            if nn_.dim() == 1:
                nn_ = nn_.reshape(-1, 1)
            self._ahead = ((nu_, np_, nn_), self._plan_start(nu_, np_, nn_, users_local), None)
        mark('gradient_exchange')
        send_c, recv_c = read_counts()
        gsplit_send = [wire_rows(c, D, True) for c in send_c]
        gsplit_recv = [wire_rows(c, D, True) for c in recv_c]
        grecv = comm.all_to_all(gwire[:sum(gsplit_send)], gsplit_send, gsplit_recv)
        # 6. owners update
        mark('owner_update')
        eng.owner_update(self, grecv, recv_c, step)
        self.steps = step
        total = comm.all_reduce(loss * scale)
        mark(None)
        self.last_exchange = dict(rows_requested=p['n'], rows_served=int(p['crow'].shape[0]),
                                  gradient_rows_sent=int(sum(send_c)),
                                  bytes_in=int(cw.shape[0]) * D * 4,
                                  bytes_out=int(sum(gsplit_send)) * D * 4)
        return total.reshape(())

    def _forward_export(self, p, cw, pos_items, neg_items, scale, users_local, step):
        """Step 4 of the module docstring for a non-empty slice."""
        dev, eng = self.device, self.engine
        codes = []
        for c in self._codes or []:          # category codes of the compact rows (items only)
            cc = torch.zeros(cw.shape[0], dtype=torch.int32, device=dev)
            cc[p['cp'].long()] = c[pos_items.long()]
            cc[p['cn'].reshape(-1).long()] = c[neg_items.reshape(-1).long()]
            codes.append(cc)
        return eng.forward_export(
            self, cw, p['send_counts'], p['send_rows'], p['cu'], p['cp'], p['cn'], self._loss_kind,
            self.hparams['num_items'], codes, self._weights, scale, users_local, step)

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
        lk = min(k, iemb.shape[0])      # a shard smaller than k contributes all of its items
        if callable(indptr):    # lazy CSR rows: MAP@k alone needs them only after the scoring pass
            if want_ap and not (want_rr or want_auc):
                shard_top = eng.shard_topk(utile, ubias, iemb, ibias, G, rank, lk)
            indptr, indices = indptr()
        if want_rr or want_auc:
            pos = eng.shard_pos_scores(utile, ubias, iemb, ibias, G, rank, indptr, indices)
            comm.all_reduce(pos)
            thr = eng.thresholds(utile, ubias, iemb, ibias, indptr, indices, pos, y)
            fused = getattr(eng, 'shard_topk_rank_counts', None)
            if want_ap and fused is not None:   # one pass over the shard for both
                shard_top, counts = fused(utile, ubias, iemb, ibias, G, rank, lk, indptr, thr)
            else:
                counts = eng.shard_rank_counts(utile, ubias, iemb, ibias, G, rank, indptr, thr)
            comm.all_reduce(counts)
        if want_ap:
            # every shard's own top-k, all gathered: G lists of k candidates per user, merged by
            # (score desc, id asc); any k <= num_items (metrics.py:170-175)
            s, i = shard_top if shard_top is not None else eng.shard_topk(
                utile, ubias, iemb, ibias, G, rank, lk)
            i = i.to(torch.int32)
            if lk < k:
                s = torch.nn.functional.pad(s, (0, k - lk), value=float('-inf'))
                i = torch.nn.functional.pad(i, (0, k - lk), value=_I31)
            parts_s, parts_i = comm.all_gather(s.contiguous()), comm.all_gather(i.contiguous())
            cs = torch.stack(parts_s, 1).contiguous()
            ci = torch.stack(parts_i, 1).contiguous()
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
