"""
GPU tests (``-m gpu``) at BASELINE.json's FULL sizes, through properties that need no CPU oracle
run at that size (the oracle-vs-kernel comparisons live in the other ``test_gpu_*`` files at sizes
the oracle finishes in seconds):

* configs[1] (138,493 x 26,744 x 64, B = 1,048,576 positives, K = 10, adaptive BPR, SparseAdam +
  SGD bias): the loss is additive over sample chunks ((sum l + sum l^2) / B, ``bpr.py:89``), a
  step is bit-reproducible, exactly the looked-up rows move, and every positive / user row does.
* configs[2] catalogue (1,000,000 items x 128): the top-k of the whole catalogue equals the merge
  of the top-k of its shards (ids bit-exact), and re-scoring the returned ids with a plain torch
  fp32 product reproduces the returned scores and torch's own top-k scores to 1e-5.
* configs[3] (100,000,000 users x 10,000,000 items x 128 on ONE GPU, B = 1,048,576, K = 10: the
  tables no cache holds, 169 GB with the SparseAdam moments): same properties as configs[1], with
  per-row fingerprints instead of table clones (a second copy of the tables does not fit).
* configs[4] (HybridPretrainedModel head, 10 M x 1 M x 128, 256-d item metadata, B = 65,536):
  the step is bit-reproducible, frozen tables do not move, the loss is chunk-additive; unfrozen,
  only looked-up embedding / bias rows move on the first Adam step; get_preds over the whole
  catalogue (grid route) equals model(users, items) on sampled pairs and the layer-by-layer route's
  metrics.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NAMES = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _c2_model(dev):
    from collie_amd.model import MatrixFactorizationModel

    class Shape:
        num_users, num_items, num_negative_samples = 138_493, 26_744, 10

    torch.manual_seed(0)
    with torch.device(dev):
        m = MatrixFactorizationModel(train=Shape(), embedding_dim=64, sparse=True, lr=1e-3,
                                     optimizer='sparse_adam', bias_optimizer='sgd', bias_lr=1e-2,
                                     loss='bpr', lr_scheduler_func=None)
    return m.to(dev).train()


def test_configs1_full_batch_properties(dev):
    U, N, K, B = 138_493, 26_744, 10, 1_048_576
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    users = torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32)
    pos = torch.randint(0, N, (B,), device=dev, generator=g, dtype=torch.int32)
    neg = torch.randint(0, N, (B, K), device=dev, generator=g)
    # leave some rows out of the batch on purpose: they must not move
    users[users >= U - 1000] = 0
    pos[pos >= N - 500] = 1
    neg[neg >= N - 500] = 2
    # a negative equal to its sample's positive would give that sample's user row a zero gradient
    neg = torch.where(neg == pos.long()[:, None], (neg + 1) % (N - 500), neg)
    batch = ((users, pos), neg)

    def one_step():
        m = _c2_model(dev)
        with torch.no_grad():           # non-zero item biases so the bias path is exercised
            m.item_biases.weight.copy_(torch.randn(N, 1, generator=g, device=dev) * 0.01)
        before = {n: getattr(m, n).weight.detach().clone() for n in NAMES}
        opts = m.configure_optimizers()
        for o in opts:
            o.zero_grad()
        loss = m.calculate_loss(batch)
        loss.backward()
        for i, o in enumerate(opts):
            m.optimizer_step(optimizer=o, optimizer_idx=i)
        m.check_status()
        return m, opts, before, loss.detach()

    g.manual_seed(11)
    m, opts, before, loss = one_step()

    # (1) the loss of the batch = the chunk losses recombined ((sum l + sum l^2) / B is additive)
    ref = _c2_model(dev)
    with torch.no_grad():
        for n in NAMES:
            getattr(ref, n).weight.copy_(before[n])
        total = 0.0
        for lo in range(0, B, 65_536):
            sl = slice(lo, lo + 65_536)
            total += float(ref.calculate_loss(((users[sl], pos[sl]), neg[sl]))) * 65_536
    assert abs(float(loss) - total / B) <= 1e-5 * abs(total / B)

    # (2) a step is bit-reproducible (fixed-order reductions, no float atomics)
    g.manual_seed(11)
    m2, opts2, _, loss2 = one_step()
    assert torch.equal(loss, loss2)
    for n in NAMES:
        assert torch.equal(getattr(m, n).weight, getattr(m2, n).weight), n

    # (3) which rows moved
    after = {n: getattr(m, n).weight.detach() for n in NAMES}
    looked_u = torch.zeros(U, dtype=torch.bool, device=dev)
    looked_u[users.long()] = True
    looked_i = torch.zeros(N, dtype=torch.bool, device=dev)
    looked_i[pos.long()] = True
    pos_rows = looked_i.clone()
    looked_i[neg.reshape(-1)] = True
    assert int((~looked_u).sum()) >= 1000 and int((~looked_i).sum()) >= 500
    moved_u = (after['user_embeddings'] != before['user_embeddings']).any(1)
    moved_i = (after['item_embeddings'] != before['item_embeddings']).any(1)
    # BPR's gradient coefficient is never exactly 0 here, so every sample's user row and positive
    # row receive a gradient; Adam's first step moves a row by lr * sign(g) wherever g != 0
    assert torch.equal(moved_u, looked_u)
    assert not bool((moved_i & ~looked_i).any())
    assert bool(moved_i[pos_rows].all())
    # rows of the selected negatives only: at most one per sample beyond the positives
    assert int(moved_i.sum()) <= int(pos_rows.sum()) + B
    # optimizer state: non-zero exactly on rows that moved (first step: m = (1 - beta1) g)
    emb_opt = opts[0]
    st_u = emb_opt.state[m.user_embeddings.weight]
    st_i = emb_opt.state[m.item_embeddings.weight]
    assert torch.equal((st_u['exp_avg'] != 0).any(1), moved_u)
    assert torch.equal((st_i['exp_avg'] != 0).any(1), moved_i)
    assert not bool((st_i['exp_avg_sq'] < 0).any())
    # biases: the user-bias gradient cancels (exact 0 here); item biases move only
    # on rows that got a non-zero gradient
    assert torch.equal(after['user_biases'], before['user_biases'])
    moved_b = (after['item_biases'] != before['item_biases']).reshape(-1)
    assert not bool((moved_b & ~moved_i).any())
    assert int(moved_b.sum()) > 0
    # first Adam step: |delta| <= lr (1 + 1e-3) on every element
    assert float((after['item_embeddings'] - before['item_embeddings']).abs().max()) <= 1e-3 * 1.001
    assert float((after['user_embeddings'] - before['user_embeddings']).abs().max()) <= 1e-3 * 1.001


def test_configs2_catalogue_topk_shard_merge_and_rescoring(dev):
    from collie_amd import _lib, ops
    N, U, D, k = 1_000_000, 4096, 128, 10
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    sd = 1.0 / (2.5 * D)
    ue = torch.randn(U, D, device=dev, generator=g) * sd * 30      # spread the scores
    ie = torch.randn(N, D, device=dev, generator=g) * sd * 30
    ub = torch.randn(U, 1, device=dev, generator=g) * 0.01
    ib = torch.randn(N, 1, device=dev, generator=g) * 0.01
    t = _lib.make_tables(ue, ie, ub, ib)
    ws = ops.EvalWorkspace(dev)
    n_eval = 1024
    slots = torch.randperm(U, device=dev, generator=g)[:n_eval].sort().values.contiguous()
    scores, ids = ops.eval_topk(t, slots, 0, N, k, None, ws)
    # order: descending score, ascending id on ties
    assert bool((scores[:, :-1] >= scores[:, 1:]).all())
    tie = scores[:, :-1] == scores[:, 1:]
    assert bool((ids[:, :-1][tie] < ids[:, 1:][tie]).all())
    assert int(ids.min()) >= 0 and int(ids.max()) < N

    # (1) shards + merge == whole catalogue, ids and scores bit for bit (8 contiguous shards)
    cap = ops.eval_topk_capacity()
    parts_s, parts_i = [], []
    bounds = [N * j // 8 for j in range(9)]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        s_, i_ = ops.eval_topk(t, slots, lo, hi, k, None, ws)
        cs = torch.full((n_eval, 1, cap), float('-inf'), device=dev)
        ci = torch.full((n_eval, 1, cap), 2 ** 31 - 1, dtype=torch.int32, device=dev)
        cs[:, 0, :k] = s_
        ci[:, 0, :k] = i_.to(torch.int32)
        parts_s.append(cs)
        parts_i.append(ci)
    ms, mi = ops.eval_merge_topk(torch.cat(parts_s, 1), torch.cat(parts_i, 1), k)
    assert torch.equal(mi, ids)
    assert torch.equal(ms, scores)

    # (2) plain torch fp32 on the same tables: re-scored ids and torch's own top-k scores, 1e-5
    full = ue[slots] @ ie.T + ub[slots] + ib.T                      # [n_eval, N]
    scale = float(full.abs().max())
    rescored = torch.gather(full, 1, ids)
    assert float((rescored - scores).abs().max()) <= 1e-5 * scale
    top = full.topk(k, dim=1).values
    assert float((top - scores).abs().max()) <= 1e-5 * scale
    # ids agree with torch's wherever torch's k-th and (k+1)-th scores are separated by more than
    # the rounding difference between the two summation orders
    top_k1 = full.topk(k + 1, dim=1)
    gaps = (top_k1.values[:, :-1] - top_k1.values[:, 1:]).min(dim=1).values
    clear = gaps > 4e-5 * scale
    assert int(clear.sum()) > n_eval // 2
    assert torch.equal(ids[clear], top_k1.indices[clear][:, :k])

    # (3) the MRR / AUC rank-count epilogues and the fused pass at this catalogue size: users with
    # 1, 15, 16 and 40 positives among them (the per-user LDS search tree holds 15), every other
    # user with three of its own top-10 items planted as positives (so that AP and RR are not ~0)
    import scipy.sparse as sp
    from collie_amd.metrics import _csr_slice
    rng = np.random.default_rng(9)
    n_pos = rng.integers(1, 13, n_eval)
    n_pos[:8] = (1, 15, 16, 40, 1, 15, 16, 40)
    ids_h = ids.cpu().numpy()
    slots_h = slots.cpu().numpy()
    rows, cols = [], []
    for u in range(n_eval):
        c = set(rng.choice(N, int(n_pos[u]), replace=False).tolist())
        if u % 2 == 0:
            c.update(int(x) for x in ids_h[u, [0, 3, 7]])
        rows += [slots_h[u]] * len(c)
        cols += sorted(c)
    csr = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, N)).tocsr()
    indptr, indices = _csr_slice(csr, slots_h, dev)
    pos, lo, hi, best_s, best_id = ops.eval_pos_scores(t, slots, indptr, indices, None, True)
    counts = torch.zeros(n_eval, 3, dtype=torch.int64, device=dev)
    s2, i2 = ops.eval_topk_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, 0, N, k, None,
                                       counts, ws)
    assert torch.equal(i2, ids) and torch.equal(s2, scores)      # fused pass == top-k pass
    only = torch.zeros(n_eval, 3, dtype=torch.int64, device=dev)
    ops.eval_rank_counts(t, slots, indptr, lo, hi, best_s, best_id, 0, N, only)
    assert torch.equal(only, counts)                              # ... == rank-count pass
    ap, rr, au = ops.eval_metrics(ids, k, n_eval, N, indptr, indices, pos, counts, None,
                                  True, True, True)
    # the positives' scores are the catalogue scores (same chain), to the rounding of part (2)
    ip = indptr.cpu().numpy()
    ix = indices.cpu().numpy()
    urow = torch.repeat_interleave(torch.arange(n_eval, device=dev), indptr[1:] - indptr[:-1])
    assert float((full[urow, indices] - pos[:indices.shape[0]]).abs().max()) <= 1e-5 * scale
    # counters against plain torch on `full`: an item within the rounding band of a threshold may
    # fall on either side, everything else must be counted exactly
    band = 1e-5 * scale
    item_ids = torch.arange(N, device=dev)
    want_rr = np.zeros(n_eval)
    want_auc = np.zeros(n_eval)
    for u in range(n_eval):
        a, b = ip[u], ip[u + 1]
        row = full[u]
        p_s = pos[a:b]
        p_i = indices[a:b]
        order = np.lexsort((p_i.cpu().numpy(), -p_s.cpu().numpy().astype(np.float64)))
        bs, bi = p_s[order[0]], p_i[order[0]]
        c0 = int(counts[u, 0])
        assert int((row > bs + band).sum()) <= c0 <= int((row >= bs - band).sum()), u
        lt_lo = int((row[None, :] < (p_s[:, None] - band)).sum())
        le_hi = int((row[None, :] <= (p_s[:, None] + band)).sum())
        assert lt_lo <= int(counts[u, 1]) <= int(counts[u, 2]) <= le_hi, u
        # torch-side formulas (metrics.py:224-240, 266-282) on torch's own scores
        first = int(((row > row[bi]) | ((row == row[bi]) & (item_ids < bi))).sum())
        want_rr[u] = 1.0 / (1.0 + first)
        sig = torch.sigmoid(row)
        psig = sig[p_i]
        is_pos = torch.zeros(N, dtype=torch.bool, device=dev)
        is_pos[p_i] = True
        neg = sig[~is_pos]
        lt = (neg[None, :] < psig[:, None]).sum().item()
        eq = (neg[None, :] == psig[:, None]).sum().item()
        want_auc[u] = (lt + 0.5 * eq) / (float(b - a) * neg.shape[0])
    np.testing.assert_allclose(rr.cpu().numpy(), want_rr, rtol=0, atol=1e-5)
    np.testing.assert_allclose(au.cpu().numpy(), want_auc, rtol=0, atol=1e-5)
    assert float(rr[::2].min()) == 1.0          # planted: the user's top item is a positive
    # AP@k (metrics.py:178-198) from the top-k ids verified above, in float64
    want_ap = np.zeros(n_eval)
    for u in range(n_eval):
        truth = set(ix[ip[u]:ip[u + 1]].tolist())
        hits, acc = 0, 0.0
        for j in range(k):
            if int(ids_h[u, j]) in truth:
                hits += 1
                acc += hits / (j + 1.0)
        want_ap[u] = acc / min(k, len(truth))
    np.testing.assert_allclose(ap.cpu().numpy(), want_ap, rtol=1e-12)
    assert float(ap[::2].min()) > 0.0


# ---- configs[3]: 100 M x 10 M x 128 on one GPU ---------------------------------------------------
def _fingerprint(w, probe):
    """One int64 per row: an exact integer hash of the row's bit patterns (sum of the 32-bit
    patterns times fixed odd 64-bit weights, wrapping) -- unchanged rows give identical values and a
    changed row collides with probability ~2^-60 (a float dot product with a probe vector did
    collide for ~14 of 10^6 rows that moved by 1e-3).  Row chunks bound the temporaries."""
    out = torch.empty(w.shape[0], dtype=torch.int64, device=w.device)
    step = 1 << 22
    for lo in range(0, w.shape[0], step):
        bits = w[lo:lo + step].view(torch.int32).to(torch.int64)
        torch.sum(bits * probe, dim=1, out=out[lo:lo + step])
    return out


def _row_any_nonzero(w):
    out = torch.empty(w.shape[0], dtype=torch.bool, device=w.device)
    step = 1 << 24
    for lo in range(0, w.shape[0], step):
        out[lo:lo + step] = (w[lo:lo + step] != 0).any(1)
    return out


def test_configs3_hbm_resident_tables_full_batch_properties(dev):
    from collie_amd.model import MatrixFactorizationModel
    U, N, D, K, B = 100_000_000, 10_000_000, 128, 10, 1_048_576
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 200 * 2 ** 30:
        pytest.skip(f'needs ~180 GB of free HBM, {free / 2 ** 30:.0f} GB available')

    class Shape:
        num_users, num_items, num_negative_samples = U, N, K

    with torch.device(dev):
        m = MatrixFactorizationModel(train=Shape(), embedding_dim=D, sparse=True, lr=1e-3,
                                     optimizer='sparse_adam', bias_optimizer='sgd', bias_lr=1e-2,
                                     loss='bpr', lr_scheduler_func=None)
    m = m.to(dev).train()
    assert m.user_embeddings.weight.device.type == 'cuda'
    opts = m.configure_optimizers()
    std = 1.0 / (2.5 * D)

    def reset():
        """Tables re-initialised in place from a fixed seed (ScaledEmbedding's N(0, 1/(2.5 D));
        scaled up so that scores are O(1)), optimizer state cleared."""
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        with torch.no_grad():
            m.user_embeddings.weight.normal_(0, std * 40, generator=g)
            m.item_embeddings.weight.normal_(0, std * 40, generator=g)
            m.user_biases.weight.zero_()
            m.item_biases.weight.normal_(0, 0.01, generator=g)
        for o in opts:
            for st in o.state.values():
                for key, v in st.items():
                    if torch.is_tensor(v):
                        v.zero_()
                    else:
                        st[key] = 0

    g = torch.Generator(device=dev)
    g.manual_seed(7)
    users = torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32)
    pos = torch.randint(0, N, (B,), device=dev, generator=g, dtype=torch.int32)
    neg = torch.randint(0, N, (B, K), device=dev, generator=g)
    neg = torch.where(neg == pos.long()[:, None], (neg + 1) % N, neg)
    batch = ((users, pos), neg)
    probe_u = torch.randint(-2 ** 62, 2 ** 62, (D,), device=dev, generator=g) * 2 + 1

    def one_step():
        reset()
        fp0 = (_fingerprint(m.user_embeddings.weight.detach(), probe_u),
               _fingerprint(m.item_embeddings.weight.detach(), probe_u))
        bias0 = m.item_biases.weight.detach().clone()
        with torch.no_grad():           # (1) chunk losses on the step's starting tables
            total = 0.0
            for lo in range(0, B, 131_072):
                sl = slice(lo, lo + 131_072)
                total += float(m.calculate_loss(((users[sl], pos[sl]), neg[sl]))) * 131_072
        for o in opts:
            o.zero_grad()
        loss = m.calculate_loss(batch)
        loss.backward()
        for i, o in enumerate(opts):
            m.optimizer_step(optimizer=o, optimizer_idx=i)
        m.check_status()
        fp1 = (_fingerprint(m.user_embeddings.weight.detach(), probe_u),
               _fingerprint(m.item_embeddings.weight.detach(), probe_u))
        return loss.detach().clone(), total / B, fp0, fp1, bias0

    loss, chunked, fp0, fp1, bias0 = one_step()
    # (1) loss = chunk losses recombined
    assert abs(float(loss) - chunked) <= 1e-5 * abs(chunked)
    assert 0.1 < float(loss) < 10.0

    # (3) exactly the looked-up rows move
    looked_u = torch.zeros(U, dtype=torch.bool, device=dev)
    looked_u[users.long()] = True
    looked_i = torch.zeros(N, dtype=torch.bool, device=dev)
    looked_i[pos.long()] = True
    pos_rows = looked_i.clone()
    looked_i[neg.reshape(-1)] = True
    moved_u = fp1[0] != fp0[0]
    moved_i = fp1[1] != fp0[1]
    assert int(looked_u.sum()) > 0.98 * B          # 1 M draws out of 100 M rows: few repeats
    assert torch.equal(moved_u, looked_u), (int((moved_u & ~looked_u).sum()),
                                            int((looked_u & ~moved_u).sum()))
    assert not bool((moved_i & ~looked_i).any())
    assert bool(moved_i[pos_rows].all())
    assert int(moved_i.sum()) <= int(pos_rows.sum()) + B
    # optimizer state: non-zero exactly on rows with a gradient; SparseAdam leaves every other row's
    # moments at their initial zeros, also for looked-up rows with a zero gradient
    emb_opt = opts[0]
    st_u = emb_opt.state[m.user_embeddings.weight]
    st_i = emb_opt.state[m.item_embeddings.weight]
    assert torch.equal(_row_any_nonzero(st_u['exp_avg']), moved_u)
    assert torch.equal(_row_any_nonzero(st_i['exp_avg']), moved_i)
    assert torch.equal(_row_any_nonzero(st_u['exp_avg_sq']), moved_u)
    # biases
    assert float(m.user_biases.weight.detach().abs().max()) == 0.0
    moved_b = (m.item_biases.weight.detach() != bias0).reshape(-1)
    assert not bool((moved_b & ~moved_i).any()) and int(moved_b.sum()) > 0
    # first Adam step: |delta| <= lr (1 + 1e-3), checked on the rows that moved
    rows = torch.nonzero(moved_u).reshape(-1)[:200_000]
    g2 = torch.Generator(device=dev)
    g2.manual_seed(5)
    del fp0, looked_u, looked_i, pos_rows
    after_rows = m.user_embeddings.weight.detach()[rows].clone()
    after_fp = fp1

    # (2) bit-reproducible: the same step again from the same start
    loss2, _, _, fp2, _ = one_step()
    assert torch.equal(loss, loss2)
    assert torch.equal(after_fp[0], fp2[0]) and torch.equal(after_fp[1], fp2[1])
    assert torch.equal(after_rows, m.user_embeddings.weight.detach()[rows])
    # |delta| of the first Adam step, against the re-initialised start values of those rows
    reset()
    delta = (after_rows - m.user_embeddings.weight.detach()[rows]).abs().max()
    assert 0 < float(delta) <= 1e-3 * 1.001


# ---- configs[4]: hybrid head at 10 M x 1 M, 256-d metadata ------------------------------------------
def _hybrid_c4(dev, freeze):
    from collie_amd.hybrid import HybridPretrainedModel
    from collie_amd.model import MatrixFactorizationModel
    U, N, D, K, MC = 10_000_000, 1_000_000, 128, 10, 256

    class Shape:
        num_users, num_items, num_negative_samples = U, N, K

    torch.manual_seed(0)
    with torch.device(dev):
        mf = MatrixFactorizationModel(train=Shape(), embedding_dim=D, lr_scheduler_func=None)
        with torch.no_grad():
            mf.user_embeddings.weight.mul_(30.0)
            mf.item_embeddings.weight.mul_(30.0)
            mf.item_biases.weight.normal_(0, 0.01)
        item_md = torch.randn(N, MC)
    model = HybridPretrainedModel(train=Shape(), trained_model=mf, item_metadata=item_md,
                                  item_metadata_layers_dims=[128], combined_layers_dims=[128, 64, 32],
                                  freeze_embeddings=freeze, loss='bpr', optimizer='adam', lr=1e-3,
                                  lr_scheduler_func=None).to(dev)
    return model.train(), mf


@pytest.mark.parametrize('freeze', [True, False])
def test_configs4_hybrid_head_full_size_properties(dev, freeze):
    U, N, K, B = 10_000_000, 1_000_000, 10, 65_536
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    users = torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32)
    pos = torch.randint(0, N, (B,), device=dev, generator=g, dtype=torch.int32)
    neg = torch.randint(0, N, (B, K), device=dev, generator=g)
    batch = ((users, pos), neg)
    runs = []
    for rep in range(2):
        model, mf = _hybrid_c4(dev, freeze)
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        if rep == 0:
            with torch.no_grad():       # chunk-additive loss on the starting parameters
                total = 0.0
                for lo in range(0, B, 8192):
                    sl = slice(lo, lo + 8192)
                    total += float(model.calculate_loss(((users[sl], pos[sl]), neg[sl]))) * 8192
        opt = model.configure_optimizers()
        opt.zero_grad()
        loss = model.calculate_loss(batch)
        loss.backward()
        opt.step()
        model.check_status()
        runs.append((loss.detach().clone(),
                     {n: p.detach().clone() for n, p in model.named_parameters()}))
        if rep == 0:
            assert abs(float(loss) - total / B) <= 1e-5 * abs(total / B)
            after = runs[0][1]
            looked_u = torch.zeros(U, dtype=torch.bool, device=dev)
            looked_u[users.long()] = True
            looked_i = torch.zeros(N, dtype=torch.bool, device=dev)
            looked_i[pos.long()] = True
            looked_i[neg.reshape(-1)] = True
            moved_any = False
            for n, p in after.items():
                changed = (p != before[n])
                is_table = n.startswith('embeddings.') or n.startswith('biases.')
                if is_table and 'embeddings' in n and freeze:
                    assert not bool(changed.any()), n           # frozen tables do not move
                elif is_table:
                    # Adam's first step with a zero gradient is the identity (m = v = 0): only
                    # looked-up rows may move
                    rows = changed.reshape(changed.shape[0], -1).any(1)
                    looked = looked_u if n.split('.')[1] == '0' else looked_i
                    assert not bool((rows & ~looked).any()), n
                    if 'embeddings' in n:
                        assert int(rows.sum()) > 0, n
                else:
                    moved_any |= bool(changed.any())            # the head's Linear layers train
            assert moved_any
            # the source MF model's tables are untouched either way (the hybrid model owns copies)
            assert torch.equal(mf.user_embeddings.weight, before['embeddings.0.weight'])
            del looked_u, looked_i
        del model, mf, opt, before
        torch.cuda.empty_cache()
    # bit-reproducible step
    assert torch.equal(runs[0][0], runs[1][0])
    for n in runs[0][1]:
        assert torch.equal(runs[0][1][n], runs[1][1][n]), n


def test_configs4_hybrid_catalogue_scoring_full_size(dev):
    """get_preds / evaluate_in_batches of the hybrid head over the whole 1 M-item catalogue
    (metrics.py:77-107, 367-386): the grid route (item share once per item, user share once per user,
    the tail in one launch) against model(users, items) on sampled pairs incl. the catalogue's edges,
    bit-reproducible, the same with and without the evaluate scope, and the metrics equal to the
    layer-by-layer route's."""
    from collie_amd.interactions import Interactions
    from collie_amd.metrics import auc, evaluate_in_batches, get_preds, mapk, mrr
    U, N = 10_000_000, 1_000_000
    model, _ = _hybrid_c4(dev, True)
    model.eval()
    user_ids = np.array([0, U - 1, 1_234_567, 42, 5_000_000])
    grid = get_preds(model, user_ids, N, dev)
    assert grid.shape == (len(user_ids), N) and bool(torch.isfinite(grid).all())
    rng = np.random.default_rng(0)
    rows = rng.integers(0, len(user_ids), 200_000)
    items = rng.integers(0, N, 200_000)
    items[:64] = np.arange(64)                      # first tile
    items[64:128] = N - 1 - np.arange(64)           # last tiles
    with torch.no_grad():
        pair = model(torch.tensor(user_ids[rows], device=dev), torch.tensor(items, device=dev))
    got = grid[torch.tensor(rows, device=dev), torch.tensor(items, device=dev)]
    scale = float(pair.abs().max())
    np.testing.assert_allclose(got.cpu().numpy(), pair.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)
    assert torch.equal(get_preds(model, user_ids, N, dev), grid)
    with model.catalogue_scope():
        assert torch.equal(get_preds(model, user_ids, N, dev), grid)
        assert torch.equal(get_preds(model, user_ids[:2], N, dev), grid[:2])

    test = Interactions(users=np.repeat(user_ids[:4] % 1000, 25), items=rng.integers(0, N, 100),
                        num_users=U, num_items=N, allow_missing_ids=True,
                        check_num_negative_samples_is_valid=False,
                        max_number_of_samples_to_consider=0)
    fused = evaluate_in_batches([mapk, mrr, auc], test, model, k=10, batch_size=3)
    model._FUSED_GRID_TAIL = False
    layered = evaluate_in_batches([mapk, mrr, auc], test, model, k=10, batch_size=3)
    np.testing.assert_allclose(fused, layered, rtol=1e-4, atol=1e-7)
    model.check_status()
