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
