"""
GPU tests (``-m gpu``) of the hybrid head: the MFMA dense-layer kernels
through the C ABI against a float64 numpy restatement / torch fp32 on the CPU, and
``HybridPretrainedModel`` against ``oracle/torch_path.py:TorchHybrid`` (the reference's torch op
sequence, ``hybrid_pretrained_matrix_factorization.py:227-464``) on identical batches.

Tolerances: 1e-5 relative to the largest magnitude of the compared tensor for single kernels
(fp32 GEMMs with a different summation order than the CPU's), 2e-4 for three-step trajectories.
Parity is pinned by torch 2.10 run here: the reference holds no known-answer vector for this model.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SLOPE = 0.01


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def close(a, b, rtol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30))


def _operand_case(rng, m, widths, gathered, act, dev):
    """Segments for ops.dense_operand + the float64 matrix they denote."""
    from collie_amd import ops
    segs, cols = [], []
    for w, g, a in zip(widths, gathered, act):
        if g:
            rows = int(rng.integers(m // 2 + 1, 2 * m + 5))
            src = rng.normal(size=(rows, w)).astype(np.float32)
            idx = rng.integers(0, rows, m).astype(np.int64 if w % 2 else np.int32)
            val = src[idx].astype(np.float64)
            idx_t = torch.tensor(idx, device=dev)
        else:
            src = rng.normal(size=(m, w)).astype(np.float32)
            val = src.astype(np.float64)
            idx_t = None
        act_t = None
        if a:
            ref = rng.normal(size=(m, w)).astype(np.float32)
            ref[rng.random((m, w)) < 0.1] = 0.0
            val = val * np.where(ref > 0, 1.0, np.float32(SLOPE).astype(np.float64))
            act_t = torch.tensor(ref, device=dev)
        segs.append((torch.tensor(src, device=dev), idx_t, act_t))
        cols.append(val)
    return ops.dense_operand(*segs), np.concatenate(cols, 1)


FWD_CASES = [
    # m, segment widths, gathered?, act mask?, n_out, leaky
    (1, [8], [False], [False], 4, True),
    (37, [5, 3], [True, False], [False, False], 33, True),
    (300, [16, 16, 24], [True, True, False], [False, False, False], 32, True),
    (513, [128, 128, 128], [True, True, False], [False, False, False], 128, True),
    (700, [256], [True], [False], 130, False),
    (260, [64], [False], [True], 128, False),
    (129, [7, 9, 1, 2], [True, True, False, True], [False, False, True, False], 1, False),
    # one output column over an identity operand: the memory-bound GEMV path
    (5000, [32], [False], [False], 1, False),
    (777, [256], [False], [False], 1, True),
    (64, [4], [False], [False], 1, False),
]


FWD_DMA_CASES = [
    # 128 output columns, 128-column segments: the LDS-DMA kernel
    (4096, [128, 128], [True, False], [False, False], 128, True),
    (1000, [256], [True], [False], 128, True),
    (64, [128], [False], [False], 128, False),
    (77, [128], [True], [False], 128, True),
    (30000, [128], [False], [True], 128, False),
    (129, [128], [True], [True], 128, False),
    # 64 / 32 output columns (tiles of 128 / 256 rows)
    (3000, [128], [False], [False], 64, True),
    (777, [64, 64], [True, False], [False, False], 64, True),
    (5000, [64], [False], [False], 32, True),
    (300, [32, 32], [True, True], [False, False], 32, False),
    # n_in = 64: 64-column images
    (3000, [64], [False], [True], 128, False),
    (100, [64], [True], [False], 128, True),
    # n_in = 384: the two waves of a column tile split K (one more rounding order: 1e-5, not bits)
    (5000, [128, 128, 128], [True, True, False], [False, False, False], 128, True),
    (64 * 300 + 5, [384], [False], [False], 128, False),
]


@pytest.mark.parametrize('case', FWD_DMA_CASES)
def test_dense_forward_dma_equals_register_kernel(dev, case, monkeypatch):
    """Same k order and pairing as the register-staged kernel: bit-identical outputs."""
    from collie_amd import ops
    m, widths, gathered, act, n_out, leaky = case
    rng = np.random.default_rng(m + 7)
    op, x = _operand_case(rng, m, widths, gathered, act, dev)
    n_in = x.shape[1]
    w = (rng.normal(size=(n_out, n_in)) / np.sqrt(n_in)).astype(np.float32)
    b = rng.normal(size=n_out).astype(np.float32)
    wt, bt = torch.tensor(w, device=dev), torch.tensor(b, device=dev)
    y = ops.dense_forward(op, m, wt, bt, leaky)
    # transposed weights (the data-gradient form): W read through swapped strides
    wt2 = torch.tensor(np.ascontiguousarray(w.T), device=dev)
    y2 = ops.dense_forward(op, m, wt2, None, False, transposed=True)
    ops.dense_set_kernels(ops.DENSE_REGISTER_STAGED)
    try:
        y0 = ops.dense_forward(op, m, wt, bt, leaky)
        y20 = ops.dense_forward(op, m, wt2, None, False, transposed=True)
    finally:
        ops.dense_set_kernels(ops.DENSE_AUTO)
    if n_in == 384:
        close(y.cpu().numpy(), y0.cpu().numpy())
        close(y2.cpu().numpy(), y20.cpu().numpy())
    else:
        assert torch.equal(y, y0) and torch.equal(y2, y20)
    # deterministic
    assert torch.equal(ops.dense_forward(op, m, wt, bt, leaky), y0 if n_in != 384 else
                       ops.dense_forward(op, m, wt, bt, leaky))
    ref = x @ w.astype(np.float64).T + b
    if leaky:
        ref = np.where(ref > 0, ref, ref * np.float64(np.float32(SLOPE)))
    close(y.cpu().numpy(), ref)


@pytest.mark.parametrize('case', [
    (4096, [128, 128], [True, False], 128, 512),     # LDS-DMA kernel, period divides the tiles
    (64 * 37 + 11, [128], [True], 128, 300),         # ... and does not; ragged last tile
    (5000, [128, 128, 128], [True, True, False], 128, 257),   # the split-K pair
    (600, [64], [False], 48, 256),                   # register-staged kernel
    (600, [5, 3], [True, False], 33, 299),           # generic kernel
])
def test_dense_forward_periodic_addend(dev, case):
    """collie_dense_forward_add: y = act(x W^T + b + P[m mod period])."""
    from collie_amd import ops
    m, widths, gathered, n_out, period = case
    rng = np.random.default_rng(m + period)
    op, x = _operand_case(rng, m, widths, gathered, [False] * len(widths), dev)
    n_in = x.shape[1]
    w = (rng.normal(size=(n_out, n_in)) / np.sqrt(n_in)).astype(np.float32)
    b = rng.normal(size=n_out).astype(np.float32)
    part = rng.normal(size=(period, n_out)).astype(np.float32)
    y = ops.dense_forward(op, m, torch.tensor(w, device=dev), torch.tensor(b, device=dev), True,
                          addend=(torch.tensor(part, device=dev), period))
    ref = x @ w.astype(np.float64).T + b + part[np.arange(m) % period]
    ref = np.where(ref > 0, ref, ref * np.float64(np.float32(SLOPE)))
    close(y.cpu().numpy(), ref)
    # a column range of W: the layer multiplies only W[:, c0:c1]
    wide = np.concatenate([rng.normal(size=(n_out, 8)).astype(np.float32), w], 1)
    y2 = ops.dense_forward(op, m, torch.tensor(wide, device=dev), torch.tensor(b, device=dev), True,
                           col_range=(8, 8 + n_in), addend=(torch.tensor(part, device=dev), period))
    close(y2.cpu().numpy(), ref)


def test_hybrid_user_share_of_first_layer_once_per_sample(dev, monkeypatch):
    """calculate_loss feeds users.repeat(1 + K): the user embedding's share of the first combined
    layer is computed once per sample and added in the epilogue; loss and every gradient must equal
    the plain evaluation of the layer."""
    from collie_amd.hybrid import HybridPretrainedModel
    U, N, D, K, B = 300, 200, 128, 5, 512
    rng = np.random.default_rng(8)
    inter, mf = _mf_model(dev, U, N, D, K, 6)
    item_md = torch.tensor(rng.normal(size=(N, 256)).astype(np.float32))
    model = HybridPretrainedModel(train=inter, trained_model=mf, item_metadata=item_md,
                                  item_metadata_layers_dims=[128], freeze_embeddings=False,
                                  loss='adaptive_bpr', lr_scheduler_func=None).to(dev)
    model.train()
    batch = ((torch.tensor(rng.integers(0, U, B), dtype=torch.int32, device=dev),
              torch.tensor(rng.integers(0, N, B), dtype=torch.int32, device=dev)),
             torch.tensor(rng.integers(0, N, (B, K)), device=dev))
    calls = []
    real = HybridPretrainedModel._head_forward

    def spy(self, users, items):
        calls.append(self._user_period_hint)
        return real(self, users, items)

    monkeypatch.setattr(HybridPretrainedModel, '_head_forward', spy)

    def run():
        model._staged_rows.clear()
        for p in model.parameters():
            p.grad = None
        loss = model.calculate_loss(batch)
        loss.backward()
        grads = {n: p.grad.cpu().numpy().copy() for n, p in model.named_parameters()
                 if p.grad is not None}
        rows = {n: torch.zeros_like(p).index_add_(0, model._staged_rows[id(p)][0].long(),
                                                  model._staged_rows[id(p)][1].reshape(
                                                      model._staged_rows[id(p)][0].shape[0],
                                                      p.shape[1]))
                .cpu().numpy() for n, p in model.named_parameters() if id(p) in model._staged_rows}
        return float(loss), grads, rows

    l1, g1, r1 = run()
    assert calls == [B]
    monkeypatch.setattr(HybridPretrainedModel, '_USER_SHARE_ONCE', False)
    l0, g0, r0 = run()
    assert calls == [B, None]
    assert abs(l1 - l0) <= 2e-6 * abs(l0)
    assert g1.keys() == g0.keys() and r1.keys() == r0.keys() and len(r0) == 4
    for k in g0:
        scale = max(np.abs(g0[k]).max(), 0.05)
        np.testing.assert_allclose(g1[k], g0[k], rtol=2e-5, atol=1e-4 * scale, err_msg=k)
    for k in r0:
        scale = max(np.abs(r0[k]).max(), 0.05)
        np.testing.assert_allclose(r1[k], r0[k], rtol=2e-5, atol=1e-4 * scale, err_msg=k)
    model.check_status()


def test_dense_forward_dma_flags_out_of_range_rows(dev):
    from collie_amd import ops
    src = torch.ones(10, 128, device=dev)
    idx = torch.arange(200, device=dev) % 10
    idx[150] = 10
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    w = torch.ones(128, 128, device=dev)
    ops.dense_forward(ops.dense_operand((src, idx, None)), 200, w, None, False, status=status)
    assert int(status.item()) == 1


@pytest.mark.parametrize('case', FWD_CASES)
def test_dense_forward_vs_float64(dev, case):
    from collie_amd import ops
    m, widths, gathered, act, n_out, leaky = case
    rng = np.random.default_rng(m + n_out)
    op, x = _operand_case(rng, m, widths, gathered, act, dev)
    n_in = x.shape[1]
    w = (rng.normal(size=(n_out, n_in)) / np.sqrt(n_in)).astype(np.float32)
    b = rng.normal(size=n_out).astype(np.float32)
    y = ops.dense_forward(op, m, torch.tensor(w, device=dev), torch.tensor(b, device=dev), leaky)
    ref = x @ w.astype(np.float64).T + b
    if leaky:
        ref = np.where(ref > 0, ref, ref * np.float64(np.float32(SLOPE)))
    close(y.cpu().numpy(), ref)


def test_dense_forward_transposed_slice_and_row_bias(dev):
    from collie_amd import ops
    rng = np.random.default_rng(5)
    m, n_out, n_in = 333, 48, 100
    op, dz = _operand_case(rng, m, [n_out], [False], [True], dev)
    w = rng.normal(size=(n_out, n_in)).astype(np.float32)
    wt = torch.tensor(w, device=dev)
    close(ops.dense_forward(op, m, wt, None, False, transposed=True).cpu().numpy(),
          dz @ w.astype(np.float64))
    close(ops.dense_forward(op, m, wt, None, False, transposed=True, col_range=(20, 61)).cpu().numpy(),
          dz @ w.astype(np.float64)[:, 20:61])
    # data gradient of a one-column layer (outer product path): dz [m, 1], W [1, 32]
    g_op, g = _operand_case(rng, m, [1], [False], [False], dev)
    w32 = rng.normal(size=(1, 32)).astype(np.float32)
    close(ops.dense_forward(g_op, m, torch.tensor(w32, device=dev), None, False,
                            transposed=True).cpu().numpy(), g @ w32.astype(np.float64))
    # final layer: Linear(.., 1) + user bias + item bias
    users = rng.integers(0, 50, m)
    items = rng.integers(0, 70, m).astype(np.int32)
    ub = rng.normal(size=(50, 1)).astype(np.float32)
    ib = rng.normal(size=(70, 1)).astype(np.float32)
    w1 = rng.normal(size=(1, n_out)).astype(np.float32)
    b1 = rng.normal(size=1).astype(np.float32)
    y = ops.dense_forward(op, m, torch.tensor(w1, device=dev), torch.tensor(b1, device=dev), False,
                          row_bias=(torch.tensor(users, device=dev), torch.tensor(ub, device=dev),
                                    torch.tensor(items, device=dev), torch.tensor(ib, device=dev)))
    ref = dz @ w1.astype(np.float64).T + b1 + ub[users] + ib[items]
    close(y.cpu().numpy(), ref)
    # the same on the GEMV path (identity operand without a mask), with one id out of range
    h_op, h = _operand_case(rng, m, [32], [False], [False], dev)
    w2 = rng.normal(size=(1, 32)).astype(np.float32)
    bad = users.copy()
    bad[7] = 50
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    y = ops.dense_forward(h_op, m, torch.tensor(w2, device=dev), torch.tensor(b1, device=dev), False,
                          row_bias=(torch.tensor(bad, device=dev), torch.tensor(ub, device=dev),
                                    torch.tensor(items, device=dev), torch.tensor(ib, device=dev)),
                          status=status)
    ref = h @ w2.astype(np.float64).T + b1 + ub[users] + ib[items]
    ref[7] = h[7] @ w2[0].astype(np.float64) + b1
    close(y.cpu().numpy(), ref)
    assert int(status.item()) == 1


def test_dense_forward_flags_out_of_range_rows(dev):
    from collie_amd import ops
    src = torch.ones(10, 8, device=dev)
    idx = torch.tensor([0, 3, 10, -1], device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    w = torch.ones(4, 8, device=dev)
    y = ops.dense_forward(ops.dense_operand((src, idx, None)), 4, w, None, False, status=status)
    assert int(status.item()) == 1
    np.testing.assert_array_equal(y.cpu().numpy(), np.array([[8.] * 4, [8.] * 4, [0.] * 4, [0.] * 4]))


WG_CASES = [
    (1, [8], [False], 4),
    (100, [5, 3], [True, False], 33),
    (3000, [128, 128, 128], [True, True, False], 128),
    (70000, [16], [True], 8),
    (5000, [32], [False], 1),
    (1111, [130], [True], 140),
    # whole 16-row slabs, dW of 128 x {128, 256, 384}: the LDS-DMA kernel
    (3008, [128, 128, 128], [True, True, False], 128),
    (4096, [256], [True], 128),
    (2064, [128], [False], 128),
    (16, [64, 64], [True, True], 128),
    (45056, [128, 256], [False, True], 128),
    (256 * 256 + 4096, [128], [True], 128),
    # the narrow layers of the default head (32-row slabs; 32 x 64: the waves also split the rows)
    (4128, [128], [False], 64),
    (2080, [64, 64], [True, False], 64),
    (4128, [64], [False], 32),
    (96, [32, 32], [True, True], 32),
]


@pytest.mark.parametrize('case', WG_CASES)
def test_dense_weight_grad_vs_float64(dev, case):
    from collie_amd import ops
    m, widths, gathered, n_out = case
    rng = np.random.default_rng(m)
    x_op, x = _operand_case(rng, m, widths, gathered, [False] * len(widths), dev)
    dz_op, dz = _operand_case(rng, m, [n_out], [False], [n_out > 1], dev)
    dw, db, _ = ops.dense_weight_grad(dz_op, x_op, m, n_out, x.shape[1])
    close(dw.cpu().numpy(), dz.T @ x)
    close(db.cpu().numpy(), dz.sum(0))
    # deterministic: a second run is bit-identical
    dw2, db2, _ = ops.dense_weight_grad(dz_op, x_op, m, n_out, x.shape[1])
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize('case', [c for c in WG_CASES if c[3] in (32, 64, 128) and c[0] % 16 == 0])
@pytest.mark.parametrize('masked', [True, False])
def test_dense_weight_grad_dma_equals_register_kernel(dev, case, masked, monkeypatch):
    """The LDS-DMA weight gradient feeds the MFMAs the same rows in the same order as the
    register-staged kernel: dW bit for bit while both cut the batch into the same chunks (up to
    256 x 256 rows; beyond that the DMA kernel uses half as many, and 32 x 64 splits the rows of a
    slab over its waves: 1e-5), db (another summation order) to 1e-5."""
    from collie_amd import ops
    m, widths, gathered, n_out = case
    rng = np.random.default_rng(m + 1)
    x_op, x = _operand_case(rng, m, widths, gathered, [False] * len(widths), dev)
    dz_op, dz = _operand_case(rng, m, [n_out], [False], [masked], dev)
    dw, db, _ = ops.dense_weight_grad(dz_op, x_op, m, n_out, x.shape[1])
    ops.dense_set_kernels(ops.DENSE_REGISTER_STAGED)
    try:
        dw0, db0, _ = ops.dense_weight_grad(dz_op, x_op, m, n_out, x.shape[1])
    finally:
        ops.dense_set_kernels(ops.DENSE_AUTO)
    if n_out == 32 or m > 256 * 256:
        close(dw.cpu().numpy(), dw0.cpu().numpy())
    else:
        assert torch.equal(dw, dw0)
    close(db.cpu().numpy(), db0.cpu().numpy())
    close(dw.cpu().numpy(), dz.T @ x)


def test_dense_weight_grad_dma_flags_out_of_range_rows(dev):
    from collie_amd import ops
    m = 64
    src = torch.ones(10, 128, device=dev)
    idx = torch.arange(m, device=dev) % 10
    idx[17] = 10
    dz = torch.ones(m, 128, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.dense_weight_grad(ops.dense_operand((dz, None, None)), ops.dense_operand((src, idx, None)),
                          m, 128, 128, status=status)
    assert int(status.item()) == 1
    status.zero_()
    idx[17] = 3
    dw, db, _ = ops.dense_weight_grad(ops.dense_operand((dz, None, None)),
                                      ops.dense_operand((src, idx, None)), m, 128, 128,
                                      status=status)
    assert int(status.item()) == 0
    assert float(dw.min()) == float(dw.max()) == m and float(db.min()) == m


@pytest.mark.parametrize('algo,kw', [('adam', dict(lr=1e-2, weight_decay=1e-3)),
                                     ('adam', dict(lr=1e-3)),
                                     ('sgd', dict(lr=0.1, weight_decay=1e-2)),
                                     ('adagrad', dict(lr=0.05))])
def test_dense_optimizers_vs_torch(dev, algo, kw):
    """collie_dense_optim_step (flat parameter) and collie_dense_rows_update (table + COO list)
    against torch.optim on the CPU for three steps."""
    from collie_amd import ops
    rng = np.random.default_rng(11)
    cls = {'adam': torch.optim.Adam, 'sgd': torch.optim.SGD, 'adagrad': torch.optim.Adagrad}[algo]
    p0 = rng.normal(size=(37, 12)).astype(np.float32)
    t0 = rng.normal(size=(90, 8)).astype(np.float32)
    pc = torch.nn.Parameter(torch.tensor(p0))
    tc = torch.nn.Parameter(torch.tensor(t0))
    opt = cls([pc, tc], **kw)
    pg = torch.tensor(p0, device=dev)
    tg = torch.tensor(t0, device=dev)
    s = {n: torch.zeros_like(x) for n, x in (('p1', pg), ('p2', pg), ('t1', tg), ('t2', tg))}
    stamp = torch.zeros(90, dtype=torch.int32, device=dev)
    for step in range(1, 4):
        g = rng.normal(size=p0.shape).astype(np.float32)
        n = 0 if step == 2 else 200              # step 2: an empty list = an all-zero gradient
        rows = rng.integers(0, 30, n)            # rows 30.. never get an entry
        vals = rng.normal(size=(n, 8)).astype(np.float32)
        pc.grad = torch.tensor(g)
        tc.grad = torch.zeros_like(tc).index_add_(0, torch.tensor(rows), torch.tensor(vals))
        opt.step()
        cfg = ops.optim_cfg(algo, step, kw['lr'], weight_decay=kw.get('weight_decay', 0.0),
                            eps=1e-10 if algo == 'adagrad' else 1e-8)
        ops.dense_optim_step(pg, torch.tensor(g, device=dev), cfg, s['p1'], s['p2'])
        ops.dense_rows_update(tg, torch.tensor(rows, dtype=torch.int64, device=dev),
                              torch.tensor(vals, device=dev), cfg, s['t1'], s['t2'], stamp)
    close(pg.cpu().numpy(), pc.detach().numpy(), 2e-5)
    close(tg.cpu().numpy(), tc.detach().numpy(), 2e-5)


# ---- model level ------------------------------------------------------------------------------------
def _mf_model(dev, U, N, D, K, seed):
    from collie_amd.interactions import Interactions
    from collie_amd.model import MatrixFactorizationModel
    rng = np.random.default_rng(seed)
    inter = Interactions(users=rng.integers(0, U, 4 * U), items=rng.integers(0, N, 4 * U),
                         num_users=U, num_items=N, allow_missing_ids=True, num_negative_samples=K,
                         seed=seed, max_number_of_samples_to_consider=0)
    torch.manual_seed(seed)
    mf = MatrixFactorizationModel(train=inter, embedding_dim=D, lr_scheduler_func=None)
    with torch.no_grad():
        mf.user_embeddings.weight.mul_(8.0)
        mf.item_embeddings.weight.mul_(8.0)
        mf.user_biases.weight.normal_(0, 0.1)
        mf.item_biases.weight.normal_(0, 0.1)
    return inter, mf


def _pair(dev, inter, mf, item_md, user_md, **kw):
    """The product model on the GPU and the torch restatement on the CPU with equal parameters."""
    from collie_amd.hybrid import HybridPretrainedModel
    from oracle.torch_path import TorchHybrid
    model = HybridPretrainedModel(train=inter, trained_model=mf, item_metadata=item_md,
                                  user_metadata=user_md, lr_scheduler_func=None, **kw)
    tables = {'user_embeddings': mf.user_embeddings.weight.detach().numpy(),
              'item_embeddings': mf.item_embeddings.weight.detach().numpy(),
              'user_biases': mf.user_biases.weight.detach().numpy(),
              'item_biases': mf.item_biases.weight.detach().numpy()}
    ref = TorchHybrid(tables, item_metadata=item_md, user_metadata=user_md,
                      item_metadata_layers_dims=kw.get('item_metadata_layers_dims'),
                      user_metadata_layers_dims=kw.get('user_metadata_layers_dims'),
                      combined_layers_dims=kw.get('combined_layers_dims', [128, 64, 32]),
                      freeze_embeddings=kw.get('freeze_embeddings', True))
    ref.load_state_dict(model.state_dict())
    assert [n for n, _ in ref.named_parameters()] == [n for n, _ in model.named_parameters()]
    return model.to(dev), ref


HYBRID_CASES = [
    dict(D=16, meta=(24, None), kw=dict(item_metadata_layers_dims=[32], combined_layers_dims=[32, 16, 8],
                                        loss='adaptive', optimizer='adam', lr=1e-2)),
    dict(D=16, meta=(24, 10), kw=dict(item_metadata_layers_dims=[32, 12], user_metadata_layers_dims=[8],
                                      combined_layers_dims=[20], freeze_embeddings=False,
                                      loss='bpr', optimizer='sgd', lr=0.2)),
    dict(D=10, meta=(7, None), kw=dict(combined_layers_dims=[6], freeze_embeddings=False,
                                       loss='warp', optimizer='adagrad', lr=5e-2)),
    dict(D=10, meta=(7, None), kw=dict(combined_layers_dims=[], freeze_embeddings=False,
                                       loss='bpr', optimizer='sgd', lr=0.1)),
    dict(D=128, meta=(256, None), kw=dict(item_metadata_layers_dims=[128], freeze_embeddings=False,
                                          loss='adaptive_bpr', optimizer='adam', lr=1e-3,
                                          weight_decay=1e-4)),
    dict(D=12, meta=(None, 9), kw=dict(combined_layers_dims=[16, 4], loss='hinge', optimizer='adam',
                                       lr=1e-2)),
]


@pytest.mark.parametrize('case', HYBRID_CASES)
def test_hybrid_forward_and_trajectory_vs_torch_path(dev, case):
    from oracle import torch_path as T
    U, N, K, B = 90, 70, 5, 96
    rng = np.random.default_rng(case['D'])
    inter, mf = _mf_model(dev, U, N, case['D'], K, seed=case['D'])
    ic, uc = case['meta']
    item_md = rng.normal(size=(N, ic)).astype(np.float32) if ic else None
    user_md = rng.normal(size=(U, uc)).astype(np.float32) if uc else None
    torch.manual_seed(1)
    model, ref = _pair(dev, inter, mf, item_md, user_md, **case['kw'])
    kw = case['kw']

    # forward(users, items)
    users = torch.tensor(rng.integers(0, U, 200))
    items = torch.tensor(rng.integers(0, N, 200))
    model.eval()
    with torch.no_grad():
        got = model(users.to(dev), items.to(dev))
        want = ref(users, items)
    close(got.cpu().numpy(), want.numpy())

    # three training steps on identical batches (trainer.py:458-470)
    model.train()
    opt = model.configure_optimizers()
    ropt = T.make_optimizers(ref, kw['optimizer'], None, kw['lr'], None, kw.get('weight_decay', 0.0))
    loss_kind = model._loss_kind
    names = [n for n, _ in model.named_parameters()]
    for it in range(3):
        batch = ((torch.tensor(rng.integers(0, U, B), dtype=torch.int32),
                  torch.tensor(rng.integers(0, N, B), dtype=torch.int32)),
                 torch.tensor(rng.integers(0, N, (B, K))))
        gbatch = ((batch[0][0].to(dev), batch[0][1].to(dev)), batch[1].to(dev))
        opt.zero_grad()
        loss = model.calculate_loss(gbatch)
        loss.backward()
        if it == 0:
            # gradients of every trainable tensor, before any optimizer amplifies rounding: the
            # Linear parameters arrive as .grad, the tables as staged (row, value) lists
            for o in ropt:
                o.zero_grad()
            ref_loss, _, _ = ref.calculate_loss(batch, T.TORCH_LOSSES[loss_kind])
            ref_loss.backward()
            rparams = dict(ref.named_parameters())
            for name, p in model.named_parameters():
                if not p.requires_grad:
                    assert rparams[name].grad is None
                    continue
                staged = model._staged_rows.get(id(p))
                if staged is not None:
                    ids, vals = staged
                    got = np.zeros(p.shape, np.float64)
                    if len(ids):     # (an empty list: an all-zero gradient)
                        np.add.at(got, ids.cpu().numpy(), vals.cpu().numpy().reshape(len(ids), -1))
                else:
                    got = p.grad.cpu().numpy()
                want = rparams[name].grad.numpy()
                # entries that cancel analytically (user bias, last Linear bias: +g and -g of every
                # sample, |g| ~ 1/B, 2B terms) are rounding residue on both sides: the floor of
                # the scale is the size of the summed terms
                scale = max(np.abs(want).max(), 0.05)
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5 * scale, err_msg=name)
        want_loss, _, _ = T.train_step(ref, ropt, batch, T.TORCH_LOSSES[loss_kind])
        model.optimizer_step(optimizer=opt)
        close(loss.item(), want_loss.item(), 2e-5)
    model.check_status()
    want_state = ref.state_dict()
    assert list(want_state) == names
    for name, got in model.state_dict().items():
        got, want = got.cpu().numpy(), want_state[name].numpy()
        if kw['optimizer'] == 'sgd':
            close(got, want, 2e-4)
        else:
            # Adam / Adagrad normalise the gradient, so an entry whose terms cancel (user bias,
            # the later Linear biases: +g for the positive, -g for the selected negative) turns
            # its fp32 rounding residue into a step of order lr on BOTH sides; such entries are
            # only comparable within that bound, all others agree far more tightly
            diff = np.abs(got - want)
            assert diff.max() <= 1.2 * 3 * kw['lr'], name
            last_bias = f'combined_layer_{len(model.combined_layers) - 1}.bias'
            if name not in ('biases.0.weight', last_bias):     # those two cancel entirely
                assert np.median(diff) <= 2e-4 * max(np.abs(want).max(), 1e-30), name


@pytest.mark.parametrize('kind', ['adaptive_bpr', 'adaptive_hinge', 'warp'])
def test_one_negative_losses_give_at_most_one_negative_a_gradient(dev, kind):
    """The invariant ``_compact_rows`` rests on (it keeps the positive and ONE negative row of a
    sample): with tied negative scores, a negative equal to the positive, and partial credit from
    metadata, the library's one-negative losses still hand a non-zero score gradient to at most one
    of a sample's K negatives."""
    from collie_amd import loss as L
    B, K, N = 257, 7, 40
    rng = np.random.default_rng(11)
    pos_items = torch.tensor(rng.integers(0, N, B), device=dev)
    neg_items = torch.tensor(rng.integers(0, N, (K, B)), device=dev)
    neg_items[3, ::2] = neg_items[1, ::2]                    # the same item twice
    neg_items[5, ::5] = pos_items[::5]                       # the positive among the negatives
    item_score = torch.tensor(rng.normal(size=N).astype(np.float32), device=dev)
    pos = (item_score[pos_items] + 0.5).clone().requires_grad_(True)
    neg = item_score[neg_items].clone()                      # equal items -> exactly tied scores
    neg[:, 7::16] = neg[0, 7::16]                            # all K tied
    neg.requires_grad_(True)
    genre = torch.tensor(rng.integers(0, 3, N), device=dev)
    L.LOSS_FUNCTIONS[kind](pos, neg, num_items=N, positive_items=pos_items, negative_items=neg_items,
                           metadata={'genre': genre}, metadata_weights={'genre': 0.3}).backward()
    per_sample = (neg.grad != 0).sum(0)
    assert int(per_sample.max()) <= 1
    # ... and that negative gets exactly minus the positive's gradient (the compact backward drops
    # the user-bias entries on this ground: (+g) + (-g) = 0 pair by pair)
    assert torch.equal(neg.grad.sum(0), -pos.grad)
    assert int(per_sample.sum()) > B // 4                    # (the case is not vacuous)


@pytest.mark.parametrize('loss', ['adaptive_bpr', 'adaptive', 'warp'])
@pytest.mark.parametrize('frozen', [True, False])
def test_hybrid_backward_on_active_rows_equals_full_backward(dev, loss, frozen, monkeypatch):
    """Under the one-negative losses the backward runs on the 2 B rows whose score gradient can
    differ from zero (``_compact_rows``); every gradient must equal the backward over all
    (1 + K) B rows, which is what the reference's autograd computes."""
    from collie_amd.hybrid import HybridPretrainedModel
    U, N, D, K, B = 300, 200, 128, 6, 512
    rng = np.random.default_rng(3)
    inter, mf = _mf_model(dev, U, N, D, K, 5)
    item_md = torch.tensor(rng.normal(size=(N, 256)).astype(np.float32))
    model = HybridPretrainedModel(train=inter, trained_model=mf, item_metadata=item_md,
                                  item_metadata_layers_dims=[128], freeze_embeddings=frozen,
                                  loss=loss, lr_scheduler_func=None,
                                  metadata_for_loss={'genre': torch.tensor(rng.integers(0, 4, N))},
                                  metadata_for_loss_weights={'genre': 0.25}).to(dev)
    model.train()
    neg = rng.integers(0, N, (B, K))
    neg[::3, 4] = neg[::3, 2]                                # tied negatives (the same item twice)
    batch = ((torch.tensor(rng.integers(0, U, B), dtype=torch.int32, device=dev),
              torch.tensor(rng.integers(0, N, B), dtype=torch.int32, device=dev)),
             torch.tensor(neg, device=dev))

    def grads():
        model._staged_rows.clear()
        for p in model.parameters():
            p.grad = None
        model.calculate_loss(batch).backward()
        out = {}
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            staged = model._staged_rows.get(id(p))
            if staged is not None:
                ids, vals = staged
                dense = np.zeros(p.shape, np.float64)
                if len(ids):     # (an empty list: an all-zero gradient)
                    np.add.at(dense, ids.cpu().numpy(), vals.cpu().numpy().reshape(len(ids), -1))
                out[name] = (dense, len(ids))
            else:
                out[name] = (p.grad.cpu().numpy().astype(np.float64), None)
        return out

    got = grads()
    user_bias_name = next(n for n, p in model.named_parameters() if p is model.biases[0].weight)
    monkeypatch.setattr(HybridPretrainedModel, '_ONE_NEGATIVE_LOSSES', ())
    want = grads()
    assert got.keys() == want.keys()
    for name in want:
        g, n_rows = got[name]
        w, n_rows_full = want[name]
        if n_rows is not None:       # staged (row, value) lists: 2 B entries instead of (1 + K) B
            # (the user biases: none at all -- their compact sums are exact zeros pair by pair)
            assert (n_rows, n_rows_full) == (0 if name == user_bias_name else 2 * B, (1 + K) * B)
        # entries that cancel analytically (+g and -g of every sample: user biases, the last
        # Linear bias) are rounding residue of 2 B terms of size ~|g| in the full backward and
        # cancel pair by pair in the compact one: the floor of the scale covers that residue
        scale = max(np.abs(w).max(), 0.05)
        np.testing.assert_allclose(g, w, rtol=1e-5, atol=1e-4 * scale, err_msg=name)
    model.check_status()


@pytest.mark.parametrize('tag,loss', [('item_meta_adaptive', 'adaptive'),
                                      ('both_meta_bpr_frozen', 'bpr'), ('raw_meta_warp', 'warp')])
def test_hybrid_vs_the_reference_hybrid_model(dev, golden_dir, tag, loss):
    """The product model against vectors made by the REFERENCE'S OWN ``HybridPretrainedModel``
    (``tests/golden/ref_hybrid.npz``, generated by tests/golden/make_golden_ref_hybrid.py through
    the import harness): ``calculate_loss``, the gradient of every trainable tensor, pair scores."""
    from collie_amd.hybrid import HybridPretrainedModel
    from collie_amd.model import MatrixFactorizationModel
    from tests.test_oracle_golden import REF_HYBRID_CASES, load_ref_hybrid_case
    c = load_ref_hybrid_case(golden_dir, tag)
    kw = dict(REF_HYBRID_CASES[tag])
    assert str(c['loss_name']) == kw.pop('loss') + '_loss'
    U, D = c['state|embeddings.0.weight'].shape
    N = c['state|embeddings.1.weight'].shape[0]

    class Shape:
        num_users, num_items, num_negative_samples = U, N, c['neg'].shape[1]

    mf = MatrixFactorizationModel(train=Shape(), embedding_dim=D, lr_scheduler_func=None)
    md = {k: torch.tensor(c[k]) for k in ('item_metadata', 'user_metadata') if k in c}
    model = HybridPretrainedModel(train=Shape(), trained_model=mf, lr_scheduler_func=None, loss=loss,
                                  **md, **kw)
    state = {k[len('state|'):]: torch.as_tensor(v) for k, v in c.items()
             if k.startswith('state|') and not k.startswith('state|_trained_model')}
    res = model.load_state_dict(state, strict=False)
    assert not res.unexpected_keys and all('_trained_model' in k for k in res.missing_keys)
    model = model.to(dev).train()
    batch = ((torch.tensor(c['users'], device=dev), torch.tensor(c['pos'], device=dev)),
             torch.tensor(c['neg'], device=dev))
    out = model.calculate_loss(batch)
    out.backward()
    assert abs(float(out.detach()) - float(c['loss'])) <= 2e-6 * abs(float(c['loss']))
    checked = 0
    for name, p in model.named_parameters():
        key = 'grad|' + name
        if not p.requires_grad:
            assert key not in c
            continue
        staged = model._staged_rows.get(id(p))
        if staged is not None:
            ids, vals = staged
            got = np.zeros(p.shape, np.float64)
            if len(ids):     # (an empty list: an all-zero gradient)
                np.add.at(got, ids.cpu().numpy(), vals.cpu().numpy().reshape(len(ids), -1))
        else:
            got = p.grad.cpu().numpy()
        want = c[key]
        # analytically cancelling entries (user biases, the last Linear bias) are rounding residue
        # on both sides: the floor of the scale is the size of the summed terms
        scale = max(np.abs(want).max(), 0.05)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5 * scale, err_msg=name)
        checked += 1
    assert checked == sum(1 for k in c if k.startswith('grad|') and '_trained_model' not in k)
    model.eval()
    with torch.no_grad():
        got = model(torch.tensor(c['pairs_users'], device=dev),
                    torch.tensor(c['pairs_items'], device=dev)).cpu().numpy()
    np.testing.assert_allclose(got, c['pairs_scores'], rtol=1e-5, atol=1e-5)
    model.check_status()


def test_hybrid_evaluate_save_load(dev, tmp_path):
    """evaluate_in_batches takes the generic get_preds route for the hybrid model; save_model /
    load_model_path round-trip predictions (tests/test_model.py:1346-1385 in the reference)."""
    from collie_amd.hybrid import HybridPretrainedModel
    from collie_amd.metrics import auc, evaluate_in_batches, mapk, mrr
    from oracle import mf_oracle as O
    U, N, K = 60, 50, 3
    rng = np.random.default_rng(2)
    inter, mf = _mf_model(dev, U, N, 8, K, seed=2)
    item_md = rng.normal(size=(N, 6)).astype(np.float32)
    torch.manual_seed(4)
    model, ref = _pair(dev, inter, mf, item_md, None, item_metadata_layers_dims=[8],
                       combined_layers_dims=[16, 8])
    model.eval()
    model._NO_GRAD_CHUNK = 128      # 7 users x 50 items per call: exercises the chunked scoring path
    got = evaluate_in_batches([mapk, mrr, auc], inter, model, k=5, batch_size=7)
    with torch.no_grad():
        uu = torch.arange(U).repeat_interleave(N)
        ii = torch.arange(N).repeat(U)
        scores = ref(uu, ii).view(U, N).numpy()
    users = np.unique(inter.mat.row)
    targets = inter.mat.tocsr()
    acc = [0.0, 0.0, 0.0]
    for i in range(int(np.ceil(len(users) / 7))):
        ur = users[i * 7:(i + 1) * 7]
        for j, fn in enumerate((O.mapk, O.mrr, O.auc)):
            acc[j] += fn(targets, ur, scores[ur], 5) * len(ur)
    want = [a / len(users) for a in acc]
    close(got, want, 1e-5)

    path = tmp_path / 'hybrid'
    model.save_model(path)
    with pytest.raises(ValueError):
        model.save_model(path)
    loaded = HybridPretrainedModel(load_model_path=str(path)).to(dev)
    with torch.no_grad():
        a = model(uu[:500].to(dev), ii[:500].to(dev))
        b = loaded(uu[:500].to(dev), ii[:500].to(dev))
    assert torch.equal(a, b)
    assert list(loaded.state_dict()) == list(model.state_dict())


@pytest.mark.parametrize('shape', [(5, 333, 128, True), (1, 1000, 64, True), (7, 1, 32, False),
                                   (3, 50, 6, True), (4, 17, 10, False), (2, 70000, 16, True)])
def test_dense_outer_sum_bit_exact(dev, shape):
    """collie_dense_outer_sum: y[j] = leaky((outer[j // period] + bias) + inner[j % period]) -- three
    fp32 operations per element, so the result is bit-equal to the same expression in torch; vector
    and scalar column paths, strided inputs, no bias."""
    from collie_amd import ops
    n_outer, period, n, leaky = shape
    rng = np.random.default_rng(n_outer * 1000 + period)
    outer_full = torch.tensor(rng.normal(size=(n_outer, n + 4)).astype(np.float32), device=dev)
    outer = outer_full[:, :n]                                         # row pitch n + 4
    inner = torch.tensor(rng.normal(size=(period, n)).astype(np.float32), device=dev)
    for bias in (torch.tensor(rng.normal(size=n).astype(np.float32), device=dev), None):
        got = ops.dense_outer_sum(outer, inner, bias, leaky)
        want = outer.repeat_interleave(period, 0)
        if bias is not None:
            want = want + bias
        want = want + inner.repeat(n_outer, 1)
        if leaky:
            want = torch.where(want > 0, want, want * np.float32(ops.LEAKY_SLOPE))
        assert got.shape == (n_outer * period, n)
        assert torch.equal(got, want)


@pytest.mark.parametrize('shape', [(3, 333), (1, 32), (5, 1), (2, 4099), (2500, 40), (70, 977)])
def test_dense_grid_tail_vs_float64(dev, shape):
    """collie_dense_grid_tail (default head's tail on the users x items grid in one launch) against
    the same network in float64; ragged item counts, a sub-range of the catalogue (item0 > 0, output
    rows with a pitch), both bias tables."""
    from collie_amd import ops
    n_users, period = shape
    rng = np.random.default_rng(period)
    f = lambda *sh: rng.normal(size=sh).astype(np.float32)
    us, it, b1 = f(n_users, 128), f(period, 128), f(128)
    w2, b2, w3, b3, w4, b4 = f(64, 128) / 8, f(64), f(32, 64) / 6, f(32), f(32) / 4, f(1)
    num_users, num_items, item0 = 50, period + 7, 5
    ub, ib = f(num_users), f(num_items)
    users = rng.integers(0, num_users, n_users)
    leaky = lambda v: np.where(v > 0, v, 0.01 * v)
    x1 = leaky(us.astype(np.float64)[:, None, :] + b1 + it.astype(np.float64)[None, :, :])
    x2 = leaky(x1 @ w2.astype(np.float64).T + b2)
    x3 = leaky(x2 @ w3.astype(np.float64).T + b3)
    want = x3 @ w4.astype(np.float64) + b4[0] + ub[users][:, None] + ib[item0:item0 + period][None, :]
    t = lambda a: torch.tensor(a, device=dev)
    out_full = torch.full((n_users, period + 3), -7.0, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.dense_grid_tail(t(us), t(it), t(b1), t(w2), t(b2), t(w3), t(b3), t(w4), t(b4), t(users), t(ub),
                        item0, t(ib), out_full[:, :period], status=status)
    assert int(status.item()) == 0
    got = out_full.cpu().numpy()
    assert (got[:, period:] == -7.0).all()
    np.testing.assert_allclose(got[:, :period], want, rtol=2e-5, atol=2e-5 * np.abs(want).max())


@pytest.mark.parametrize('layout', [
    dict(D=16, ic=24, kw=dict(item_metadata_layers_dims=[32], combined_layers_dims=[32, 16, 8])),
    dict(D=128, ic=256, kw=dict(item_metadata_layers_dims=[128])),          # BASELINE configs[4] head
    dict(D=10, ic=7, kw=dict(combined_layers_dims=[6])),                    # raw metadata columns
    dict(D=12, ic=5, kw=dict(item_metadata_layers_dims=[9, 4], combined_layers_dims=[24, 8])),
])
@pytest.mark.parametrize('chunk', [1 << 21, 1000, 97])
def test_hybrid_score_catalogue_equals_pairwise_scores(dev, layout, chunk):
    """get_preds' users x catalogue grid (metrics.py:77-107) with the item side of the first
    combined layer computed once per item == model(users, items) on the expanded pairs, for one
    pass, for user-wise splits (chunk 1000 >= N) and for item-range splits (chunk 97 < N)."""
    from collie_amd.metrics import get_preds
    U, N, K = 40, 333, 3
    rng = np.random.default_rng(layout['D'])
    inter, mf = _mf_model(dev, U, N, layout['D'], K, seed=3)
    item_md = rng.normal(size=(N, layout['ic'])).astype(np.float32)
    torch.manual_seed(5)
    model, ref = _pair(dev, inter, mf, item_md, None, **layout['kw'])
    model.eval()
    user_ids = rng.permutation(U)[:11]
    uu = torch.tensor(user_ids).repeat_interleave(N)
    ii = torch.arange(N).repeat(len(user_ids))
    with torch.no_grad():
        pairwise = model(uu.to(dev), ii.to(dev)).view(len(user_ids), N).cpu().numpy()
        want = ref(uu, ii).view(len(user_ids), N).numpy()
    model._NO_GRAD_CHUNK = chunk
    grid = model.score_catalogue(user_ids, N)
    assert grid is not None and grid.shape == (len(user_ids), N)
    scale = np.abs(want).max()
    np.testing.assert_allclose(grid.cpu().numpy(), pairwise, rtol=1e-5, atol=2e-6 * scale)
    np.testing.assert_allclose(grid.cpu().numpy(), want, rtol=1e-5, atol=2e-6 * scale)
    assert torch.equal(get_preds(model, user_ids, N, dev), grid)
    assert torch.equal(model.score_catalogue(torch.tensor(user_ids), N), grid)     # ids as a tensor
    assert model.score_catalogue([], N).shape == (0, N)
    if layout['D'] == 128:
        # the default head takes the one-launch tail (collie_dense_grid_tail); layer by layer the
        # same scores up to summation order
        model._FUSED_GRID_TAIL = False
        by_layer = model.score_catalogue(user_ids, N)
        model._FUSED_GRID_TAIL = True
        assert not torch.equal(by_layer, grid)          # i.e. the fused kernel did run above
        np.testing.assert_allclose(grid.cpu().numpy(), by_layer.cpu().numpy(), rtol=1e-5,
                                   atol=2e-6 * scale)
    # inside a scope the item share is kept between calls; outside it is rebuilt (parameters may
    # have changed in between)
    with model.catalogue_scope():
        assert torch.equal(model.score_catalogue(user_ids, N), grid)
        assert torch.equal(model.score_catalogue(user_ids[:3], N), grid[:3])
        assert len(model._catalogue_share) == 1
    with torch.no_grad():
        model.combined_layers[0].weight.mul_(0.5)
    assert not torch.equal(model.score_catalogue(user_ids, N), grid)
    model.check_status()


def test_hybrid_score_catalogue_declines_other_layouts(dev):
    """User metadata or a one-layer head: no split, get_preds takes model(users, items)."""
    from collie_amd.metrics import get_preds
    U, N, K = 30, 45, 3
    rng = np.random.default_rng(8)
    inter, mf = _mf_model(dev, U, N, 8, K, seed=8)
    item_md = rng.normal(size=(N, 6)).astype(np.float32)
    user_md = rng.normal(size=(U, 4)).astype(np.float32)
    for imd, umd, kw in [(item_md, user_md, dict(combined_layers_dims=[8])),
                         (item_md, None, dict(combined_layers_dims=[]))]:
        torch.manual_seed(1)
        model, ref = _pair(dev, inter, mf, imd, umd, **kw)
        model.eval()
        assert model.score_catalogue(np.arange(5), N) is None
        with torch.no_grad():
            want = ref(torch.arange(5).repeat_interleave(N), torch.arange(N).repeat(5)).view(5, N)
        close(get_preds(model, np.arange(5), N, dev).cpu().numpy(), want.numpy())


def test_hybrid_forward_backward_vs_numpy_oracle(dev):
    """The product model against oracle/hybrid_oracle.py (numpy): scores, Linear gradients and the
    staged table gradients for an arbitrary d loss / d scores."""
    from oracle import hybrid_oracle as H
    U, N, K = 70, 55, 4
    rng = np.random.default_rng(9)
    inter, mf = _mf_model(dev, U, N, 12, K, seed=9)
    item_md = rng.normal(size=(N, 8)).astype(np.float32)
    user_md = rng.normal(size=(U, 4)).astype(np.float32)
    torch.manual_seed(2)
    model, _ = _pair(dev, inter, mf, item_md, user_md, item_metadata_layers_dims=[16, 8],
                     combined_layers_dims=[24, 12], freeze_embeddings=False)
    model.train()
    M = 333
    users = rng.integers(0, U, M)
    items = rng.integers(0, N, M)
    g = rng.normal(size=M).astype(np.float32)
    scores = model(torch.tensor(users, device=dev), torch.tensor(items, device=dev))
    scores.backward(torch.tensor(g, device=dev))

    def wb(layers):
        return None if layers is None else [(l.weight.detach().cpu().numpy(), l.bias.detach().cpu().numpy())
                                            for l in layers]
    tables = {'user_embeddings': model.embeddings[0].weight.detach().cpu().numpy(),
              'item_embeddings': model.embeddings[1].weight.detach().cpu().numpy(),
              'user_biases': model.biases[0].weight.detach().cpu().numpy(),
              'item_biases': model.biases[1].weight.detach().cpu().numpy()}
    params = {'item_metadata_layers': wb(model.item_metadata_layers), 'user_metadata_layers': None,
              'combined_layers': wb(model.combined_layers)}
    s, cache = H.forward(params, tables, users, items, item_metadata=item_md, user_metadata=user_md)
    close(scores.detach().cpu().numpy(), s)
    want = H.backward(params, tables, cache, g)
    for (dw, db), layer in zip(want['combined_layers'], model.combined_layers):
        close(layer.weight.grad.cpu().numpy(), dw)
        close(layer.bias.grad.cpu().numpy(), db)
    for (dw, db), layer in zip(want['item_metadata_layers'], model.item_metadata_layers):
        close(layer.weight.grad.cpu().numpy(), dw)
        close(layer.bias.grad.cpu().numpy(), db)
    for name, table in (('user_embeddings', model.embeddings[0].weight),
                        ('item_embeddings', model.embeddings[1].weight),
                        ('user_biases', model.biases[0].weight), ('item_biases', model.biases[1].weight)):
        ids, vals = model._staged_rows[id(table)]
        got = np.zeros(table.shape, np.float64)
        if len(ids):     # (an empty list: an all-zero gradient)
            np.add.at(got, ids.cpu().numpy(), vals.cpu().numpy().reshape(len(ids), -1))
        close(got, want[name])


@pytest.mark.parametrize('B,K', [(1, 1), (257, 7), (4096, 10)])
def test_active_rows_equals_torch_selection(dev, B, K):
    """collie_active_rows = arange / |g|.argmax / stack / index_select of _compact_rows' first
    version: first maximum on ties, k = 0 when every negative gradient is zero."""
    from collie_amd import ops
    rng = np.random.default_rng(B + K)
    m = (1 + K) * B
    g = rng.normal(size=m).astype(np.float32)
    gn = g[B:].reshape(K, B)
    gn[:, ::3] = 0.0                                          # no negative carries a gradient
    if K > 2:
        gn[2, 1::3] = gn[0, 1::3] = 9.0                       # tie: the first maximum wins
        gn[1, 1::3] = -9.0 if K > 1 else gn[1, 1::3]
    g = torch.tensor(g, device=dev)
    users = torch.tensor(rng.integers(0, 1000, m), device=dev)
    items = torch.tensor(rng.integers(0, 1000, m), device=dev)
    act, g2, u2, i2 = ops.active_rows(g, B, K, users, items)
    ar = torch.arange(B, device=dev)
    ksel = g[B:].view(K, B).abs().argmax(dim=0)
    want = torch.stack([ar, B + ksel * B + ar], dim=1).reshape(-1)
    assert torch.equal(act, want)
    assert torch.equal(g2, g[want]) and torch.equal(u2, users[want]) and torch.equal(i2, items[want])


@pytest.mark.parametrize('widths', [(128, 128, 64, 32), (5,), (12, 7, 64, 3, 256, 8)])
def test_gather_rows_equals_index_select(dev, widths):
    from collie_amd import ops
    rng = np.random.default_rng(len(widths))
    m_src, m_out = 1000, 777
    base = [torch.tensor(rng.normal(size=(m_src, w + 4)).astype(np.float32), device=dev)
            for w in widths]
    tensors = [b[:, :w] for b, w in zip(base, widths)]        # row stride larger than the width
    idx = torch.tensor(rng.integers(0, m_src, m_out), device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    got = ops.gather_rows(tensors, idx, status)
    for t, o in zip(tensors, got):
        assert o.is_contiguous() and torch.equal(o, t.index_select(0, idx))
    assert int(status) == 0
    bad = idx.clone()
    bad[5] = m_src
    got = ops.gather_rows(tensors[:1], bad, status)
    assert int(status) == 1 and float(got[0][5].abs().sum()) == 0.0


@pytest.mark.parametrize('dtypes', [(torch.int32, torch.int32, torch.int64),
                                    (torch.int64, torch.int32, torch.int32)])
def test_pair_ids_equals_repeat_and_cat(dev, dtypes):
    from collie_amd import ops
    rng = np.random.default_rng(4)
    B, K = 333, 6
    users = torch.tensor(rng.integers(0, 10**6, B), device=dev).to(dtypes[0])
    pos = torch.tensor(rng.integers(0, 10**6, B), device=dev).to(dtypes[1])
    neg = torch.tensor(rng.integers(0, 10**6, (B, K)), device=dev).to(dtypes[2])
    for n in (neg, neg.t().contiguous().t()):                 # row-major and strided negatives
        au, ai = ops.pair_ids(users, pos, n)
        assert torch.equal(au, users.long().repeat(K + 1))
        assert torch.equal(ai, torch.cat([pos.long(), n.t().long().flatten()]))


@pytest.mark.parametrize('algo', ['adam', 'sgd', 'adagrad'])
def test_dense_optim_step_multi_equals_single_steps(dev, algo):
    """One launch over several parameters of a group = one launch per parameter, bit for bit."""
    from collie_amd import ops
    rng = np.random.default_rng(2)
    shapes = [(128, 384), (128,), (64, 128), (64,), (1, 32), (1,), (3, 5)] + [(7,)] * 12
    ps = [torch.tensor(rng.normal(size=s).astype(np.float32), device=dev) for s in shapes]
    gs = [torch.tensor(rng.normal(size=s).astype(np.float32), device=dev) for s in shapes]
    single = [p.clone() for p in ps]
    st = {k: [torch.zeros_like(p) for p in ps] for k in ('a1', 'a2', 'b1', 'b2')}
    for step in (1, 2):
        cfg = ops.optim_cfg(algo, step, 1e-2, weight_decay=1e-3 if algo != 'adagrad' else 0.0,
                            eps=1e-10 if algo == 'adagrad' else 1e-8)
        s1 = st['a1'] if algo != 'sgd' else None
        s2 = st['a2'] if algo == 'adam' else None
        ops.dense_optim_step_multi(ps, gs, cfg, s1, s2)           # 19 tensors: two launches
        for i, p in enumerate(single):
            ops.dense_optim_step(p, gs[i], cfg, st['b1'][i] if algo != 'sgd' else None,
                                 st['b2'][i] if algo == 'adam' else None)
    for a, b in zip(ps, single):
        assert torch.equal(a, b)
    for a, b in zip(st['a1'] + st['a2'], st['b1'] + st['b2']):
        assert torch.equal(a, b)
