"""
Thin tensor-level wrappers over the C ABI: they only marshal torch tensors (device pointers,
current HIP stream) into ``libcollie_hip.so`` calls.  No arithmetic happens in this file.
"""
import ctypes
from ctypes import byref, c_void_p

import torch

from collie_amd import _lib
from collie_amd._lib import (Batch, DenseOperand, LossCfg, OptimCfg, check, idx_dtype, load,
                             make_tables, ptr, require_hip, stream_ptr)


def make_batch(users, pos_items, neg_items):
    """((users[B], pos[B]), neg[B, K]) exactly as the loaders emit it; no copies, no casts."""
    for name, t in (('users', users), ('pos_items', pos_items), ('neg_items', neg_items)):
        require_hip(t, name)
    if users.dim() != 1 or pos_items.dim() != 1 or neg_items.dim() != 2:
        raise ValueError('expected users[B], pos_items[B], neg_items[B, K]')
    B = users.shape[0]
    if pos_items.shape[0] != B or neg_items.shape[0] != B:
        raise ValueError('users, pos_items and neg_items disagree on the batch size')
    if users.stride(0) != 1:
        users = users.contiguous()
    if pos_items.stride(0) != 1:
        pos_items = pos_items.contiguous()
    batch = Batch(users.data_ptr(), pos_items.data_ptr(), neg_items.data_ptr(), idx_dtype(users),
                  idx_dtype(pos_items), idx_dtype(neg_items), neg_items.shape[1], B,
                  neg_items.stride(0), neg_items.stride(1))
    batch._keepalive = (users, pos_items, neg_items)
    return batch


def make_loss_cfg(kind, num_items, metadata_codes=None, metadata_weights=None):
    cfg = LossCfg()
    cfg.kind = _lib.LOSS_KINDS[kind] if isinstance(kind, str) else int(kind)
    cfg.num_items = int(num_items)
    metadata_codes = metadata_codes or []
    metadata_weights = metadata_weights or []
    if len(metadata_codes) > _lib.MAX_META:
        raise ValueError(f'at most {_lib.MAX_META} metadata arrays are supported')
    cfg.n_meta = len(metadata_codes)
    for i, (codes, w) in enumerate(zip(metadata_codes, metadata_weights)):
        require_hip(codes, 'metadata')
        assert codes.dtype == torch.int32 and codes.is_contiguous()
        cfg.meta[i] = codes.data_ptr()
        cfg.meta_weight[i] = float(w)
    cfg._keepalive = list(metadata_codes)
    return cfg


def train_workspace_bytes(batch_size, num_neg, dim, num_users, num_items):
    return int(load().collie_mf_train_workspace_bytes(batch_size, num_neg, dim, num_users,
                                                      num_items))


class TrainWorkspace:
    """
    Device scratch of ONE model: zero-filled at creation (it holds per-item step stamps), grown
    on demand, reused every step.  ``next_step()`` hands out the step ids the stamps compare to.
    """

    def __init__(self, device, start_step=0):
        self.device = device
        self.buf = None
        # Step ids only ever grow (also across buffer re-creation: a fresh zero-filled buffer is
        # valid with any id >= 1).  Optimizer-owned ``_row_stamp`` arrays outlive the buffer and
        # compare against the same ids, so a restart at 1 would make stale stamps look current.
        self.step_id = int(start_step)

    def ensure(self, batch_size, num_neg, dim, num_users, num_items):
        key = (batch_size, num_neg, dim, num_users, num_items)
        if key != getattr(self, '_key', None):
            self._need = train_workspace_bytes(*key)
            self._key = key
        need = self._need
        if need == 0:
            raise ValueError('empty batch')
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.zeros(need, dtype=torch.uint8, device=self.device)
        return self.buf

    def next_step(self):
        self.step_id += 1
        if self.step_id >= 2 ** 31 - 1:
            self.buf.zero_()
            self.step_id = 1
        return self.step_id


def train_forward(tables, batch, loss_cfg, ws_buf, need_grad, step_id, loss_out, s_pos_out=None,
                  s_neg_out=None, status=None):
    check(load().collie_mf_train_forward(
        byref(tables), byref(batch), byref(loss_cfg), ptr(ws_buf), ws_buf.numel(),
        int(need_grad), step_id, ptr(loss_out), ptr(s_pos_out), ptr(s_neg_out),
        ptr(status), stream_ptr()), 'collie_mf_train_forward')


EXPLICIT_LOSSES = {'mse': 0, 'mae': 1}


def explicit_forward(tables, users, items, ratings, kind, y_range, ws_buf, need_grad, loss_out,
                     preds_out=None, status=None):
    """Explicit branch of calculate_loss: MSE / MAE of forward(users, items) vs ratings, with the
    backward staged in ``ws_buf`` (``collie_mf_explicit_forward``)."""
    has_y = y_range is not None
    check(load().collie_mf_explicit_forward(
        byref(tables), ptr(users), idx_dtype(users), ptr(items), idx_dtype(items), ptr(ratings),
        users.shape[0], EXPLICIT_LOSSES[kind], int(has_y), float(y_range[0]) if has_y else 0.0,
        float(y_range[1]) if has_y else 0.0, ptr(ws_buf), ws_buf.numel(), 1 if need_grad else 0,
        ptr(loss_out), ptr(preds_out), ptr(status), stream_ptr()), 'collie_mf_explicit_forward')


def train_build_segments(tables, batch, ws_buf):
    check(load().collie_mf_train_build_segments(
        byref(tables), byref(batch), ptr(ws_buf), ws_buf.numel(), stream_ptr()),
        'collie_mf_train_build_segments')


def _ptr_array(tensors):
    arr = (c_void_p * 2)()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def train_update(tables, batch, ws_buf, is_bias, dense, step_id, per_side, grad_scale=None):
    """
    per_side: {side: (OptimCfg, state1, state2, row_stamp)} for the tables this optimizer owns
    (side = _lib.SIDE_USER / _lib.SIDE_ITEM); both sides are updated by one launch.
    """
    cfgs = (OptimCfg * 2)()
    s1, s2, st = [None, None], [None, None], [None, None]
    mask = 0
    for side, (cfg, a, b, c) in per_side.items():
        cfgs[side] = cfg
        s1[side], s2[side], st[side] = a, b, c
        mask |= 1 << side
    check(load().collie_mf_train_update(
        byref(tables), byref(batch), ptr(ws_buf), ws_buf.numel(), mask, int(is_bias),
        1 if dense else 0, step_id, cfgs, _ptr_array(s1), _ptr_array(s2), _ptr_array(st),
        ptr(grad_scale), stream_ptr()), 'collie_mf_train_update')


def train_update_needs_segments(tables, batch, dense):
    """False where ``train_update`` coalesces inside its own launch (small sparse batches)."""
    rc = load().collie_mf_train_update_needs_segments(byref(tables), byref(batch),
                                                      1 if dense else 0)
    check(min(rc, 0), 'collie_mf_train_update_needs_segments')
    return rc != 0


def train_update_all(tables, batch, ws_buf, bias_mode, dense, step_id, per_table, grad_scale=None):
    """per_table: four entries (user_emb, item_emb, user_bias, item_bias), each ``None`` or
    ``(OptimCfg, state1, state2, row_stamp)`` (``collie_mf_train_update_all``)."""
    cfgs = (OptimCfg * 4)()
    s1, s2, st = (c_void_p * 4)(), (c_void_p * 4)(), (c_void_p * 4)()
    for i, entry in enumerate(per_table):
        if entry is None:
            cfgs[i].algo = -1
            continue
        cfg, a, b, c = entry
        cfgs[i] = cfg
        s1[i] = a.data_ptr() if a is not None else None
        s2[i] = b.data_ptr() if b is not None else None
        st[i] = c.data_ptr() if c is not None else None
    check(load().collie_mf_train_update_all(
        byref(tables), byref(batch), ptr(ws_buf), ws_buf.numel(), int(bias_mode),
        1 if dense else 0, step_id, cfgs, s1, s2, st, ptr(grad_scale), stream_ptr()),
        'collie_mf_train_update_all')


STEP_DEVICE = -1           # COLLIE_STEP_DEVICE
STEP_SCHEDULE_MAX = 1024   # COLLIE_STEP_SCHEDULE_MAX


def train_schedule_steps(tables, batch, ws_buf, first_step_id, n_steps, cfgs):
    """cfgs: four OptimCfg (user_emb, item_emb, user_bias, item_bias; ``None`` = not updated) as
    they stand at the first scheduled step (``collie_mf_train_schedule_steps``)."""
    arr = (OptimCfg * 4)()
    for i, c in enumerate(cfgs):
        if c is None:
            arr[i].algo = -1
        else:
            arr[i] = c
    check(load().collie_mf_train_schedule_steps(
        byref(tables), byref(batch), ptr(ws_buf), ws_buf.numel(), int(first_step_id), int(n_steps),
        arr, stream_ptr()), 'collie_mf_train_schedule_steps')


DEFER_LOSS = 2             # COLLIE_DEFER_LOSS
NO_STAMPS = 4              # COLLIE_NO_STAMPS
SMALL_COALESCE_MAX = 4096  # csrc/train_ws.h kSmallMax: coalesces this small cut no pieces


def batch_copy(src, dst):
    check(load().collie_mf_batch_copy(byref(src), byref(dst), stream_ptr()), 'collie_mf_batch_copy')


def train_advance_step(tables, batch, ws_buf, loss_out=None, status=None):
    check(load().collie_mf_train_advance_step(
        byref(tables), byref(batch), ptr(ws_buf), ws_buf.numel(), ptr(loss_out), ptr(status),
        stream_ptr()), 'collie_mf_train_advance_step')


def train_export_grad(tables, batch, ws_buf, side, is_bias, step_id):
    """
    Coalesced gradient of one table as torch's ``grad.coalesce()`` would hold it: (rows int64 [n],
    values [n, D or 1]) over every looked-up row, zero-gradient rows included.  Synchronises;
    meant for tests and for callers that want ``.grad``.
    """
    B = batch.batch
    cap = 2 * B
    D = 1 if is_bias else tables.dim
    dev = ws_buf.device
    rows = torch.empty(cap, dtype=torch.int64, device=dev)
    vals = torch.empty(cap, D, dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    touched = None
    if side == _lib.SIDE_ITEM:
        touched = torch.zeros(tables.num_items, dtype=torch.uint8, device=dev)
    check(load().collie_mf_train_export_grad(
        byref(tables), byref(batch), ptr(ws_buf), ws_buf.numel(), side, 1 if is_bias else 0,
        step_id, ptr(rows), ptr(vals), ptr(count), ptr(touched), stream_ptr()),
        'collie_mf_train_export_grad')
    n = int(count.item())
    rows, vals = rows[:n], vals[:n]
    if touched is None:
        return rows, vals
    all_rows = torch.nonzero(touched).reshape(-1)
    full = torch.zeros(all_rows.shape[0], D, dtype=torch.float32, device=dev)
    full[torch.searchsorted(all_rows, rows)] = vals
    return all_rows, full


def optim_cfg(algo, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, lr_decay=0.0):
    return OptimCfg(_lib.OPT_ALGOS[algo] if isinstance(algo, str) else int(algo), int(step),
                    float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                    float(lr_decay))


def mf_score(tables, users, items, y_range=None, status=None, dropout=None):
    """``dropout = (p, seed, call)``: training-mode forward with nn.Dropout on both rows."""
    require_hip(users, 'users')
    require_hip(items, 'items')
    users = users.contiguous()
    items = items.contiguous()
    if users.shape != items.shape or users.dim() != 1:
        raise ValueError('users and items must be 1-d tensors of the same length')
    out = torch.empty(users.shape[0], dtype=torch.float32, device=users.device)
    has_y = y_range is not None
    if dropout is not None:
        p, seed, call = dropout
        check(load().collie_mf_score_dropout(
            byref(tables), ptr(users), idx_dtype(users), ptr(items), idx_dtype(items),
            users.shape[0], int(has_y), float(y_range[0]) if has_y else 0.0,
            float(y_range[1]) if has_y else 0.0, float(p), seed & 0xFFFFFFFFFFFFFFFF,
            call & 0xFFFFFFFFFFFFFFFF, ptr(out), ptr(status), stream_ptr()),
            'collie_mf_score_dropout')
        return out
    check(load().collie_mf_score(
        byref(tables), ptr(users), idx_dtype(users), ptr(items), idx_dtype(items),
        users.shape[0], int(has_y), float(y_range[0]) if has_y else 0.0,
        float(y_range[1]) if has_y else 0.0, ptr(out), ptr(status), stream_ptr()),
        'collie_mf_score')
    return out


def mf_score_backward(tables, users, items, grad_out, y_range=None, status=None, dropout=None):
    users = users.contiguous()
    items = items.contiguous()
    grad_out = grad_out.contiguous().float()
    M, D = users.shape[0], tables.dim
    dev = users.device
    gu = torch.empty(M, D, dtype=torch.float32, device=dev)
    gi = torch.empty(M, D, dtype=torch.float32, device=dev)
    gb = torch.empty(M, 1, dtype=torch.float32, device=dev)
    has_y = y_range is not None
    if dropout is not None:
        p, seed, call = dropout
        check(load().collie_mf_score_dropout_backward(
            byref(tables), ptr(users), idx_dtype(users), ptr(items), idx_dtype(items), M,
            int(has_y), float(y_range[0]) if has_y else 0.0, float(y_range[1]) if has_y else 0.0,
            float(p), seed & 0xFFFFFFFFFFFFFFFF, call & 0xFFFFFFFFFFFFFFFF, ptr(grad_out), ptr(gu),
            ptr(gi), ptr(gb), ptr(status), stream_ptr()), 'collie_mf_score_dropout_backward')
        return gu, gi, gb
    check(load().collie_mf_score_backward(
        byref(tables), ptr(users), idx_dtype(users), ptr(items), idx_dtype(items), M, int(has_y),
        float(y_range[0]) if has_y else 0.0, float(y_range[1]) if has_y else 0.0, ptr(grad_out),
        ptr(gu), ptr(gi), ptr(gb), ptr(status), stream_ptr()), 'collie_mf_score_backward')
    return gu, gi, gb


def embedding_similarity(emb, idx):
    """Cosine similarity of every row of ``emb`` with row ``idx`` (``collie_embedding_similarity``)."""
    require_hip(emb, 'embeddings')
    emb = emb.contiguous().float()
    out = torch.empty(emb.shape[0], dtype=torch.float32, device=emb.device)
    check(load().collie_embedding_similarity(ptr(emb), emb.shape[0], emb.shape[1], int(idx), ptr(out),
                                             stream_ptr()), 'collie_embedding_similarity')
    return out


def loss_forward_backward(loss_cfg, s_pos, s_neg_KB, pos_items=None, neg_items_KB=None,
                          want_grad=True):
    """Stand-alone loss on score tensors.  s_neg_KB is [K, B].  Returns (loss[1], g_pos, g_neg)."""
    require_hip(s_pos, 'positive_scores')
    require_hip(s_neg_KB, 'negative_scores')
    s_pos = s_pos.contiguous().float()
    s_neg_KB = s_neg_KB.contiguous().float()
    K, B = s_neg_KB.shape
    dev = s_pos.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    g_pos = torch.empty(B, dtype=torch.float32, device=dev) if want_grad else None
    g_neg = torch.empty(K, B, dtype=torch.float32, device=dev) if want_grad else None
    partials = torch.empty(2 * int(load().collie_loss_partials(B)), dtype=torch.float32, device=dev)
    if pos_items is not None:
        pos_items = pos_items.contiguous().long()
        neg_items_KB = neg_items_KB.contiguous().long()
    check(load().collie_loss_forward_backward(
        byref(loss_cfg), ptr(s_pos), ptr(s_neg_KB), B, K, ptr(pos_items), ptr(neg_items_KB),
        ptr(loss), ptr(g_pos), ptr(g_neg), ptr(partials), stream_ptr()),
        'collie_loss_forward_backward')
    return loss, g_pos, g_neg




# ---- hybrid head: dense layers --------------------------------------------------------------------
LEAKY_SLOPE = 0.01      # F.leaky_relu default (hybrid_pretrained_matrix_factorization.py:357,389)


DENSE_AUTO, DENSE_REGISTER_STAGED = 0, 1


def dense_set_kernels(family):
    """Kernel family of the dense entry points (comparison runs and tests; see the header)."""
    check(load().collie_dense_set_kernels(int(family)), 'collie_dense_set_kernels')


def dense_operand(*segments):
    """Operand of a dense layer = column-wise concatenation of ``segments``; a segment is
    ``(src [rows, width], idx [m] or None, act_ref [m, width] or None)`` (see include/collie_hip.h)."""
    if not 1 <= len(segments) <= _lib.DENSE_MAX_SEGS:
        raise ValueError(f'a dense operand has 1..{_lib.DENSE_MAX_SEGS} segments')
    op = DenseOperand()
    op.n_seg = len(segments)
    keep = []
    width = 0
    for i, (src, idx, act) in enumerate(segments):
        require_hip(src, 'dense operand')
        if src.dim() == 1:
            src = src.reshape(-1, 1)
        if src.dtype != torch.float32 or src.stride(1) != 1:
            src = src.float().contiguous()
        seg = op.seg[i]
        seg.src = src.data_ptr()
        seg.ld = src.stride(0) if src.shape[0] > 1 else src.shape[1]
        seg.rows = src.shape[0]
        seg.width = src.shape[1]
        if idx is not None:
            idx = idx.contiguous()
            seg.idx = idx.data_ptr()
            seg.idx_dtype = idx_dtype(idx)
        if act is not None:
            if act.dim() == 1:
                act = act.reshape(-1, 1)
            assert act.dtype == torch.float32 and act.stride(1) == 1 and act.shape[1] == src.shape[1]
            seg.act_ref = act.data_ptr()
            seg.act_ld = act.stride(0) if act.shape[0] > 1 else act.shape[1]
        width += src.shape[1]
        keep.append((src, idx, act))
    op._keepalive = keep
    op._width = width
    return op


def dense_forward(x_op, m, weight, bias, leaky, transposed=False, out=None, row_bias=None,
                  status=None, col_range=None, addend=None):
    """``F.leaky_relu(nn.Linear)`` (``leaky``) or plain ``nn.Linear`` over an operand;
    ``transposed=True`` multiplies by ``weight.T`` instead (data backward ``dZ @ W``).
    ``row_bias = (users, user_bias_table, items, item_bias_table)`` adds the two bias-table rows.
    ``col_range`` without ``transposed``: the layer only multiplies W's input columns c0:c1.
    ``addend = (P [period, n_out], period)``: ``+ P[row mod period]`` before the activation."""
    require_hip(weight, 'weight')
    assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.dim() == 2
    w_ptr = ptr(weight)
    if transposed:
        # dZ @ W[:, c0:c1]: the "output features" of this product are W's input columns
        c0, c1 = col_range if col_range is not None else (0, weight.shape[1])
        n_out, n_in = c1 - c0, weight.shape[0]
        ld_out, ld_in = 1, weight.shape[1]
        w_ptr = c_void_p(weight.data_ptr() + 4 * c0)
    else:
        c0, c1 = col_range if col_range is not None else (0, weight.shape[1])
        n_out, n_in = weight.shape[0], c1 - c0
        ld_out, ld_in = weight.shape[1], 1
        w_ptr = c_void_p(weight.data_ptr() + 4 * c0)
    if out is None:
        out = torch.empty(m, n_out, dtype=torch.float32, device=weight.device)
    users = items = ub = ib = None
    nu = ni = 0
    if row_bias is not None:
        users, ub, items, ib = row_bias
        users, items = users.contiguous(), items.contiguous()
        nu, ni = ub.shape[0], ib.shape[0]
    add_p, add_ld, add_period = None, 0, 0
    if addend is not None:
        add_p, add_period = addend
        assert (add_p.dtype == torch.float32 and add_p.stride(1) == 1
                and add_p.shape == (add_period, n_out))
        add_ld = add_p.stride(0)
    check(load().collie_dense_forward_add(
        byref(x_op), m, w_ptr, ld_out, ld_in, ptr(bias), n_in, n_out, 1 if leaky else 0,
        LEAKY_SLOPE, ptr(users), idx_dtype(users) if users is not None else 0, ptr(ub), nu,
        ptr(items), idx_dtype(items) if items is not None else 0, ptr(ib), ni, ptr(add_p), add_ld,
        add_period, ptr(out), out.stride(0) if out.shape[0] > 1 else out.shape[1], ptr(status),
        stream_ptr()), 'collie_dense_forward_add')
    return out


def active_rows(g, batch, num_neg, users, items):
    """``collie_active_rows``: (act [2B] int64, g [2B], users [2B], items [2B]) -- the positive and
    the one negative per sample whose score gradient can differ from zero."""
    require_hip(g, 'score gradient')
    assert g.dtype == torch.float32 and g.is_contiguous() and g.numel() == (1 + num_neg) * batch
    assert users.dtype == torch.int64 and items.dtype == torch.int64
    assert users.is_contiguous() and items.is_contiguous() and users.numel() == items.numel() == g.numel()
    dev = g.device
    act = torch.empty(2 * batch, dtype=torch.int64, device=dev)
    g_out = torch.empty(2 * batch, dtype=torch.float32, device=dev)
    u_out = torch.empty(2 * batch, dtype=torch.int64, device=dev)
    i_out = torch.empty(2 * batch, dtype=torch.int64, device=dev)
    check(load().collie_active_rows(ptr(g), batch, num_neg, ptr(users), ptr(items), ptr(act),
                                    ptr(g_out), ptr(u_out), ptr(i_out), stream_ptr()),
          'collie_active_rows')
    return act, g_out, u_out, i_out


def pair_ids(users, pos_items, neg_items):
    """``collie_pair_ids``: (all_users, all_items), each [(1 + K) B] int64, in ``calculate_loss``'s
    order (the B positives, then negative k of every sample)."""
    batch = make_batch(users, pos_items, neg_items)
    B, K = neg_items.shape
    all_users = torch.empty((1 + K) * B, dtype=torch.int64, device=users.device)
    all_items = torch.empty((1 + K) * B, dtype=torch.int64, device=users.device)
    check(load().collie_pair_ids(byref(batch), ptr(all_users), ptr(all_items), stream_ptr()),
          'collie_pair_ids')
    return all_users, all_items


def gather_rows(tensors, idx, status=None):
    """``[t.index_select(0, idx) for t in tensors]`` for 2-d float32 tensors with the same number of
    rows, four per launch (``collie_gather_rows``)."""
    assert idx.dtype == torch.int64 and idx.is_contiguous()
    outs = []
    m_out = idx.shape[0]
    for i0 in range(0, len(tensors), 4):
        group = tensors[i0:i0 + 4]
        n = len(group)
        m_src = group[0].shape[0]
        res = []
        for t in group:
            require_hip(t, 'gather source')
            assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] == m_src
            res.append(torch.empty(m_out, t.shape[1], dtype=torch.float32, device=t.device))
        srcs = (c_void_p * n)(*[t.data_ptr() for t in group])
        lds = (ctypes.c_int64 * n)(*[t.stride(0) if t.shape[0] > 1 else t.shape[1] for t in group])
        widths = (ctypes.c_int32 * n)(*[t.shape[1] for t in group])
        outp = (c_void_p * n)(*[o.data_ptr() for o in res])
        check(load().collie_gather_rows(n, srcs, lds, widths, outp, ptr(idx), m_out, m_src,
                                        ptr(status), stream_ptr()), 'collie_gather_rows')
        outs.extend(res)
    return outs


def dense_outer_sum(outer, inner, bias, leaky, out=None):
    """``act((outer[j // period] + bias) + inner[j % period])`` for ``j < len(outer) * period``,
    ``period = len(inner)`` (the first combined layer on the users x catalogue grid)."""
    require_hip(outer, 'outer')
    require_hip(inner, 'inner')
    assert outer.dtype == inner.dtype == torch.float32 and outer.dim() == inner.dim() == 2
    assert outer.stride(1) == 1 and inner.stride(1) == 1 and outer.shape[1] == inner.shape[1]
    n_outer, n = outer.shape
    period = inner.shape[0]
    if out is None:
        out = torch.empty(n_outer * period, n, dtype=torch.float32, device=outer.device)
    assert out.shape == (n_outer * period, n) and out.stride(1) == 1 and out.dtype == torch.float32
    check(load().collie_dense_outer_sum(
        ptr(outer), outer.stride(0) if n_outer > 1 else n, n_outer, ptr(inner),
        inner.stride(0) if period > 1 else n, period, ptr(bias), n, 1 if leaky else 0, LEAKY_SLOPE,
        ptr(out), out.stride(0) if out.shape[0] > 1 else n, stream_ptr()), 'collie_dense_outer_sum')
    return out


def dense_grid_tail(user_share, item_share, b1, w2, b2, w3, b3, w4, b4, users, user_bias, item0,
                    item_bias, out, status=None):
    """Scores ``out[u, i]`` of the users x items grid through the default head's tail
    (128 -> 64 -> 32 -> 1) from the first layer's two shares; see ``collie_dense_grid_tail``."""
    for t in (user_share, item_share, w2, w3, w4, out):
        require_hip(t, 'dense_grid_tail')
        assert t.dtype == torch.float32
    n_users, period = user_share.shape[0], item_share.shape[0]
    assert user_share.shape[1] == item_share.shape[1] == 128
    assert user_share.stride(1) == 1 and item_share.stride(1) == 1
    assert w2.shape == (64, 128) and w3.shape == (32, 64) and w4.numel() == 32
    assert w2.is_contiguous() and w3.is_contiguous() and w4.is_contiguous()
    assert out.shape == (n_users, period) and out.stride(1) == 1
    users = users.contiguous()
    check(load().collie_dense_grid_tail(
        ptr(user_share), user_share.stride(0) if n_users > 1 else 128, n_users, ptr(item_share),
        item_share.stride(0) if period > 1 else 128, period, ptr(b1), ptr(w2), ptr(b2), ptr(w3),
        ptr(b3), ptr(w4), ptr(b4), LEAKY_SLOPE, ptr(users), idx_dtype(users), ptr(user_bias),
        user_bias.shape[0], item0, ptr(item_bias), item_bias.shape[0], ptr(out),
        out.stride(0) if n_users > 1 else period, ptr(status), stream_ptr()),
        'collie_dense_grid_tail')
    return out


def dense_weight_grad(dz_op, x_op, m, n_out, n_in, want_bias=True, workspace=None, status=None):
    """(dW [n_out, n_in], db [n_out] or None, workspace) of ``nn.Linear`` from the two operands."""
    dev = torch.device('cuda', torch.cuda.current_device())
    need = int(load().collie_dense_weight_grad_workspace_bytes(m, n_out, n_in))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    dw = torch.empty(n_out, n_in, dtype=torch.float32, device=dev)
    db = torch.empty(n_out, dtype=torch.float32, device=dev) if want_bias else None
    check(load().collie_dense_weight_grad(
        byref(dz_op), byref(x_op), m, n_out, n_in, LEAKY_SLOPE, ptr(dw), ptr(db), ptr(workspace),
        workspace.numel(), ptr(status), stream_ptr()), 'collie_dense_weight_grad')
    return dw, db, workspace


def dense_optim_step(param, grad, cfg, state1=None, state2=None, grad_scale=None):
    """``optimizer.step()`` of Adam / SGD / Adagrad on one dense parameter (in place)."""
    require_hip(param, 'parameter')
    assert param.is_contiguous() and grad.is_contiguous() and grad.dtype == torch.float32
    check(load().collie_dense_optim_step(
        ptr(param), ptr(grad), param.numel(), byref(cfg), ptr(state1), ptr(state2), ptr(grad_scale),
        stream_ptr()), 'collie_dense_optim_step')


def dense_optim_step_multi(params, grads, cfg, state1=None, state2=None):
    """``dense_optim_step`` for several parameters of one optimizer group, 16 per launch."""
    for i0 in range(0, len(params), 16):
        ps, gs = params[i0:i0 + 16], grads[i0:i0 + 16]
        n = len(ps)
        for p_, g_ in zip(ps, gs):
            require_hip(p_, 'parameter')
            assert p_.is_contiguous() and g_.is_contiguous() and g_.dtype == torch.float32 \
                and g_.numel() == p_.numel()

        def arr(ts):
            if ts is None:
                return None
            return (c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts[i0:i0 + 16]])
        check(load().collie_dense_optim_step_multi(
            n, arr(params), arr(grads), (ctypes.c_int64 * n)(*[p_.numel() for p_ in ps]),
            byref(cfg), arr(state1), arr(state2), stream_ptr()), 'collie_dense_optim_step_multi')


def dense_rows_update(table, rows, vals, cfg, state1=None, state2=None, row_stamp=None,
                      status=None, workspace=None):
    """Dense-gradient optimizers on a table whose gradient is an uncoalesced COO list."""
    require_hip(table, 'table')
    rows = rows.contiguous()
    vals = vals.contiguous().float()
    m = rows.shape[0]
    dim = table.shape[1] if table.dim() == 2 else 1
    need = int(load().collie_sparse_rows_workspace_bytes(m, dim))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=table.device)
    check(load().collie_dense_rows_update(
        ptr(table), table.shape[0], dim, ptr(rows), idx_dtype(rows), ptr(vals), m, byref(cfg),
        ptr(state1), ptr(state2), ptr(row_stamp), ptr(workspace), workspace.numel(), ptr(status),
        stream_ptr()), 'collie_dense_rows_update')
    return workspace


# ---- device-side batch assembly + negative sampling ------------------------------------------------
def sample_batch(rows, cols, order, start, batch, indptr, indices, num_items, num_neg, max_tries,
                 seed, position_base):
    """((users int32 [B], pos int32 [B]), neg int64 [B, K]) for dataset entries
    order[start:start + batch], produced on the device (``collie_sample_batch``)."""
    require_hip(rows, 'rows')
    dev = rows.device
    users = torch.empty(batch, dtype=torch.int32, device=dev)
    pos = torch.empty(batch, dtype=torch.int32, device=dev)
    neg = torch.empty(batch, num_neg, dtype=torch.int64, device=dev)
    check(load().collie_sample_batch(
        ptr(rows), ptr(cols), rows.shape[0], ptr(order), start, batch, ptr(indptr), ptr(indices),
        num_items, num_neg, max_tries, seed & 0xFFFFFFFFFFFFFFFF,
        position_base & 0xFFFFFFFFFFFFFFFF, ptr(users), ptr(pos), ptr(neg), stream_ptr()),
        'collie_sample_batch')
    return (users, pos), neg


# ---- row-sharded training (collie_shard_*) --------------------------------------------------------
def _i64_array(values):
    arr = (ctypes.c_int64 * len(values))(*[int(v) for v in values])
    return arr


def shard_wire_rows(counts, dim, gradient):
    """Rows of an exchange buffer whose peer blocks hold ``counts[p]`` entries (shard_wire.h)."""
    rows = int(load().collie_shard_wire_rows(_i64_array(counts), len(counts), dim,
                                             1 if gradient else 0))
    if rows < 0:
        raise _lib.CollieHipError('collie_shard_wire_rows: bad arguments')
    return rows


class ShardPlanner:
    """Scratch of the requester-side plan + the per-step plan call (``collie_shard_plan``)."""

    def __init__(self, num_users, num_items, world, dim, device):
        self.num_users, self.num_items = int(num_users), int(num_items)
        self.world, self.dim = int(world), int(dim)
        self.device = device
        self.buf = None

    def plan(self, users, pos_items, neg_items, local_rank=-1):
        """local_rank >= 0: this rank owns every user of the batch (only the items are planned)."""
        batch = make_batch(users, pos_items, neg_items)
        B, K = neg_items.shape
        dev = users.device
        need = int(load().collie_shard_plan_workspace_bytes(B, K, self.world, self.num_users,
                                                            self.num_items, local_rank))
        if need == 0:
            raise _lib.CollieHipError('collie_shard_plan_workspace_bytes: bad sizes')
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        cu = torch.empty(B, dtype=torch.int32, device=dev)
        cp = torch.empty(B, dtype=torch.int32, device=dev)
        cn = torch.empty(B, K, dtype=torch.int64, device=dev)
        cap = min((2 + K) * B, self.num_users + self.num_items)
        send_rows = torch.empty(cap, dtype=torch.int64, device=dev)
        counts = torch.empty(2 * self.world + 2, dtype=torch.int64, device=dev)
        check(load().collie_shard_plan(
            byref(batch), self.num_users, self.num_items, self.world, self.dim, local_rank,
            ptr(self.buf), self.buf.numel(), ptr(cu), ptr(cp), ptr(cn), ptr(send_rows), cap,
            ptr(counts),
            stream_ptr()), 'collie_shard_plan')
        return cu, cp, cn, send_rows, counts


def shard_serve_rows(emb, bias, rows, peer_counts, item_touch, item_base, step_id, status=None):
    """Owner side of the lookup exchange: the requested rows as the peers' blocks of the lookup
    buffer ``[rows, dim]`` (``collie_shard_serve_rows``); stamps the served item rows."""
    require_hip(emb, 'table')
    dim = emb.shape[1]
    out_rows = shard_wire_rows(peer_counts, dim, False)
    out = torch.empty(out_rows, dim, dtype=torch.float32, device=emb.device)
    rows = rows.contiguous()
    assert rows.dtype == torch.int64 and rows.shape[0] == sum(peer_counts)
    check(load().collie_shard_serve_rows(
        ptr(emb), ptr(bias), emb.shape[0], dim, ptr(rows), _i64_array(peer_counts),
        len(peer_counts), ptr(out), out_rows, ptr(item_touch), item_base, step_id, ptr(status),
        stream_ptr()), 'collie_shard_serve_rows')
    return out


def shard_unpack_bias(buf, peer_counts):
    """Bias values of a received lookup buffer, indexed like its rows."""
    out = torch.empty(buf.shape[0], dtype=torch.float32, device=buf.device)
    check(load().collie_shard_unpack_bias(
        ptr(buf), buf.shape[0], buf.shape[1], _i64_array(peer_counts), len(peer_counts), ptr(out),
        stream_ptr()), 'collie_shard_unpack_bias')
    return out


def shard_export_grads(tables, batch, train_ws_buf, lookup_counts, send_rows, grad_scale, workspace,
                       users_local=False):
    """``collie_shard_export_grads``: (wire [rows, dim], peer_counts [world + 1] device int64,
    workspace)."""
    B, D, world = batch.batch, tables.dim, len(lookup_counts)
    dev = train_ws_buf.device
    wire_rows = 3 * B + 2 * ((3 * B + D - 1) // D + world)
    wire = torch.empty(wire_rows, D, dtype=torch.float32, device=dev)
    counts = torch.empty(world + 1, dtype=torch.int64, device=dev)
    need = int(load().collie_shard_export_workspace_bytes(B))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    check(load().collie_shard_export_grads(
        byref(tables), byref(batch), ptr(train_ws_buf), train_ws_buf.numel(), world,
        1 if users_local else 0, _i64_array(lookup_counts), ptr(send_rows), ptr(grad_scale), ptr(wire), wire_rows,
        ptr(counts), ptr(workspace), workspace.numel(), stream_ptr()), 'collie_shard_export_grads')
    return wire, counts, workspace


def shard_owner_update(emb, bias, item_base, wire, peer_counts, item_touch, row_seg, cfg, bias_cfg,
                       state, bias_state, status=None, workspace=None):
    """``collie_shard_owner_update``; returns the (possibly grown) workspace."""
    require_hip(emb, 'table')
    dim = emb.shape[1]
    entries = int(sum(peer_counts))
    wire_rows = wire.shape[0] if wire is not None else 0
    need = int(load().collie_shard_owner_workspace_bytes(entries, wire_rows, dim))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=emb.device)
    check(load().collie_shard_owner_update(
        ptr(emb), ptr(bias), emb.shape[0], dim, item_base, ptr(wire), wire_rows,
        _i64_array(peer_counts), len(peer_counts), ptr(item_touch), ptr(row_seg), byref(cfg),
        byref(bias_cfg), ptr(state[0]), ptr(state[1]), ptr(bias_state[0]), ptr(bias_state[1]),
        ptr(workspace), workspace.numel(), ptr(status), stream_ptr()),
        'collie_shard_owner_update')
    return workspace


# ---- evaluation -----------------------------------------------------------------------------------
def eval_topk_capacity():
    return int(load().collie_mf_eval_topk_capacity())


def _y(y_range):
    has_y = y_range is not None
    return int(has_y), float(y_range[0]) if has_y else 0.0, float(y_range[1]) if has_y else 0.0


class EvalWorkspace:
    def __init__(self, device):
        self.device = device
        self.buf = None

    def ensure(self, n_users, n_items_shard):
        need = int(load().collie_mf_eval_workspace_bytes(n_users, n_items_shard))
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.buf


def eval_topk(tables, user_ids, item_begin, item_end, k, y_range, ws, id_stride=1, id_offset=0):
    """(scores [n, k] float32, ids [n, k] int64): descending score, ascending id on ties."""
    require_hip(user_ids, 'user_ids')
    assert user_ids.dtype == torch.int64 and user_ids.is_contiguous()
    n = user_ids.shape[0]
    dev = user_ids.device
    buf = ws.ensure(n, item_end - item_begin)
    scores = torch.empty(n, k, dtype=torch.float32, device=dev)
    ids = torch.empty(n, k, dtype=torch.int64, device=dev)
    has_y, lo, hi = _y(y_range)
    check(load().collie_mf_eval_topk(byref(tables), ptr(user_ids), n, item_begin, item_end,
                                     id_stride, id_offset, k, has_y, lo, hi, ptr(scores),
                                     ptr(ids), ptr(buf), buf.numel(), stream_ptr()),
          'collie_mf_eval_topk')
    return scores, ids


def eval_bf16x3_ok(dim, k):
    """Can the opt-in bf16 x 3 scoring pass take this shape (``collie_mf_eval_bf16x3_ok``)?"""
    return bool(load().collie_mf_eval_bf16x3_ok(int(dim), int(k)))


def eval_split_items(item_emb):
    """The item table's bf16 x 3 image ``[rows, 3, dim]`` for ``eval_topk_bf16x3``; valid until
    the table changes."""
    require_hip(item_emb, 'item_embeddings')
    assert item_emb.dtype == torch.float32 and item_emb.is_contiguous()
    rows, dim = item_emb.shape
    out = torch.empty(rows, 3, dim, dtype=torch.bfloat16, device=item_emb.device)
    check(load().collie_mf_eval_split_bf16x3(ptr(item_emb), rows, dim, ptr(out), stream_ptr()),
          'collie_mf_eval_split_bf16x3')
    return out


def eval_topk_bf16x3(tables, item_split, user_ids, item_begin, item_end, k, y_range, ws,
                     id_stride=1, id_offset=0):
    """``eval_topk`` with the catalogue pass on bf16 MFMA (three-way split, exact re-scoring of
    16 candidates per user): same scores, ids equal unless > 16 - k items tie the k-th to ~1e-7."""
    require_hip(user_ids, 'user_ids')
    assert user_ids.dtype == torch.int64 and user_ids.is_contiguous()
    assert item_split.dtype == torch.bfloat16 and item_split.is_contiguous()
    n = user_ids.shape[0]
    dev = user_ids.device
    need = int(load().collie_mf_eval_bf16x3_workspace_bytes(n, item_end - item_begin))
    if ws.buf is None or ws.buf.numel() < need:
        ws.buf = torch.empty(need, dtype=torch.uint8, device=dev)
    scores = torch.empty(n, k, dtype=torch.float32, device=dev)
    ids = torch.empty(n, k, dtype=torch.int64, device=dev)
    has_y, lo, hi = _y(y_range)
    check(load().collie_mf_eval_topk_bf16x3(
        byref(tables), ptr(item_split), ptr(user_ids), n, item_begin, item_end, id_stride,
        id_offset, k, has_y, lo, hi, ptr(scores), ptr(ids), ptr(ws.buf), ws.buf.numel(),
        stream_ptr()), 'collie_mf_eval_topk_bf16x3')
    return scores, ids


def eval_topk_static(tables, user_ids, item_begin, item_end, k, y_range, scores, ids,
                     id_stride=1, id_offset=0):
    """``eval_topk`` into caller-provided ``scores`` / ``ids`` [n, k] without any allocation: the
    candidate lists live in the library's static device buffer (``collie_mf_eval_topk_static``;
    at most 1024 users per call, calls on one GPU must not overlap)."""
    require_hip(user_ids, 'user_ids')
    n = user_ids.shape[0]
    if user_ids.dtype != torch.int64 or not user_ids.is_contiguous():
        raise ValueError('user_ids must be a contiguous int64 tensor')
    for name, t, dt in (('scores', scores, torch.float32), ('ids', ids, torch.int64)):
        if (t.dtype != dt or tuple(t.shape) != (n, k) or not t.is_contiguous()
                or t.device != user_ids.device):
            raise ValueError(f'{name} must be a contiguous {dt} tensor of shape ({n}, {k}) on '
                             f'{user_ids.device}')
    has_y, lo, hi = _y(y_range)
    check(load().collie_mf_eval_topk_static(
        byref(tables), ptr(user_ids), user_ids.shape[0], item_begin, item_end, id_stride, id_offset,
        k, has_y, lo, hi, ptr(scores), ptr(ids), stream_ptr()), 'collie_mf_eval_topk_static')
    return scores, ids


def eval_pos_scores(tables, user_ids, indptr, indices, y_range, want_thresholds, id_stride=1,
                    id_offset=0, given_scores=None):
    n = user_ids.shape[0]
    nnz = indices.shape[0]
    dev = user_ids.device
    pos = given_scores if given_scores is not None else torch.empty(
        max(nnz, 1), dtype=torch.float32, device=dev)
    lo = hi = best_s = best_id = None
    if want_thresholds:
        lo = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
        hi = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
        best_s = torch.empty(n, dtype=torch.float32, device=dev)
        best_id = torch.empty(n, dtype=torch.int64, device=dev)
    has_y, ylo, yhi = _y(y_range)
    check(load().collie_mf_eval_pos_scores(
        byref(tables), ptr(user_ids), n, ptr(indptr), ptr(indices), nnz, has_y, ylo, yhi,
        0 if given_scores is not None else 1, id_stride, id_offset, ptr(pos), ptr(lo), ptr(hi),
        ptr(best_s), ptr(best_id), stream_ptr()),
        'collie_mf_eval_pos_scores')
    return pos, lo, hi, best_s, best_id


def eval_rank_counts(tables, user_ids, indptr, lo, hi, best_s, best_id, item_begin, item_end,
                     counts, id_stride=1, id_offset=0):
    check(load().collie_mf_eval_rank_counts(
        byref(tables), ptr(user_ids), user_ids.shape[0], ptr(indptr), ptr(lo), ptr(hi),
        ptr(best_s), ptr(best_id), item_begin, item_end, id_stride, id_offset, ptr(counts),
        stream_ptr()),
        'collie_mf_eval_rank_counts')


def eval_topk_rank_counts(tables, user_ids, indptr, lo, hi, best_s, best_id, item_begin, item_end,
                          k, y_range, counts, ws, id_stride=1, id_offset=0):
    """``eval_topk`` and ``eval_rank_counts`` in one pass over the items (one scoring GEMM)."""
    require_hip(user_ids, 'user_ids')
    assert user_ids.dtype == torch.int64 and user_ids.is_contiguous()
    n = user_ids.shape[0]
    dev = user_ids.device
    buf = ws.ensure(n, item_end - item_begin)
    scores = torch.empty(n, k, dtype=torch.float32, device=dev)
    ids = torch.empty(n, k, dtype=torch.int64, device=dev)
    has_y, ylo, yhi = _y(y_range)
    check(load().collie_mf_eval_topk_rank_counts(
        byref(tables), ptr(user_ids), n, item_begin, item_end, id_stride, id_offset, k, has_y, ylo,
        yhi, ptr(indptr), ptr(lo), ptr(hi), ptr(best_s), ptr(best_id), ptr(scores), ptr(ids),
        ptr(counts), ptr(buf), buf.numel(), stream_ptr()), 'collie_mf_eval_topk_rank_counts')
    return scores, ids


def eval_metrics(topk_ids, k, n_users, n_items, indptr, indices, pos_scores, counts, y_range,
                 want_ap, want_rr, want_auc):
    dev = indptr.device
    ap = torch.empty(n_users, dtype=torch.float64, device=dev) if want_ap else None
    rr = torch.empty(n_users, dtype=torch.float32, device=dev) if want_rr else None
    auc = torch.empty(n_users, dtype=torch.float64, device=dev) if want_auc else None
    has_y, ylo, yhi = _y(y_range)
    check(load().collie_mf_eval_metrics(
        ptr(topk_ids), k, n_users, n_items, ptr(indptr), ptr(indices), ptr(pos_scores),
        ptr(counts), has_y, ylo, yhi, ptr(ap), ptr(rr), ptr(auc), stream_ptr()),
        'collie_mf_eval_metrics')
    return ap, rr, auc


# ---- metric reductions over a materialised score matrix (generic route) ---------------------------
def _preds_arg(preds):
    require_hip(preds, 'preds')
    if preds.dim() != 2:
        raise ValueError('preds must be a [users, items] matrix')
    if preds.dtype != torch.float32 or preds.stride(1) != 1:
        preds = preds.float().contiguous()
    ld = preds.stride(0) if preds.shape[0] > 1 else preds.shape[1]
    return preds, ld


def preds_topk(preds, k):
    """``preds.topk(k, dim=1)`` with ties in ascending id order (``collie_preds_topk``)."""
    preds, ld = _preds_arg(preds)
    n, N = preds.shape
    scores = torch.empty(n, k, dtype=torch.float32, device=preds.device)
    ids = torch.empty(n, k, dtype=torch.int64, device=preds.device)
    check(load().collie_preds_topk(ptr(preds), n, N, ld, k, ptr(scores), ptr(ids), stream_ptr()),
          'collie_preds_topk')
    return scores, ids


def preds_pos_scores(preds, indptr, indices, status=None):
    preds, ld = _preds_arg(preds)
    n, N = preds.shape
    nnz = indices.shape[0]
    dev = preds.device
    pos = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    lo = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    hi = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
    best_s = torch.empty(n, dtype=torch.float32, device=dev)
    best_id = torch.empty(n, dtype=torch.int64, device=dev)
    check(load().collie_preds_pos_scores(ptr(preds), n, N, ld, ptr(indptr), ptr(indices), nnz,
                                         ptr(pos), ptr(lo), ptr(hi), ptr(best_s), ptr(best_id),
                                         ptr(status), stream_ptr()), 'collie_preds_pos_scores')
    return pos, lo, hi, best_s, best_id


def preds_rank_counts(preds, indptr, lo, hi, best_s, best_id, counts):
    preds, ld = _preds_arg(preds)
    n, N = preds.shape
    check(load().collie_preds_rank_counts(ptr(preds), n, N, ld, ptr(indptr), ptr(lo), ptr(hi),
                                          ptr(best_s), ptr(best_id), ptr(counts), stream_ptr()),
          'collie_preds_rank_counts')


def eval_merge_topk(cand_scores, cand_ids, k):
    n, nslots, slot_len = cand_scores.shape
    assert cand_ids.dtype == torch.int32 and cand_ids.shape == cand_scores.shape
    dev = cand_scores.device
    scores = torch.empty(n, k, dtype=torch.float32, device=dev)
    ids = torch.empty(n, k, dtype=torch.int64, device=dev)
    check(load().collie_mf_eval_merge_topk(ptr(cand_scores.contiguous()),
                                           ptr(cand_ids.contiguous()), n, nslots, slot_len, k,
                                           ptr(scores),
                                           ptr(ids), stream_ptr()), 'collie_mf_eval_merge_topk')
    return scores, ids
