"""
GPU parity tests (``-m gpu``): the HIP kernels, called through the C ABI, against the golden
vectors captured from the reference's loss functions / torch (tests/golden) and against the numpy
oracle on seeded inputs.  Bars: bit-exact for ids / touched-row sets, 1e-5 relative for fp32.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')
RTOL = 1e-5


def close(a, b, rtol=RTOL, scale=None):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    s = np.abs(b).max() if scale is None else scale
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(float(s), 1e-30))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def to_dev(tables, dev):
    return {n: torch.tensor(tables[n], device=dev).contiguous() for n in NAMES}


def tables_struct(td):
    from collie_amd import _lib
    return _lib.make_tables(td['user_embeddings'], td['item_embeddings'], td['user_biases'],
                            td['item_biases'])


def meta_codes(z, dev):
    return [torch.tensor(z['meta_a'], dtype=torch.int32, device=dev),
            torch.tensor(z['meta_b'], dtype=torch.int32, device=dev)], [0.2, 0.3]


def test_forward_pairs_vs_golden(golden_dir, dev):
    from collie_amd import ops
    z = np.load(os.path.join(golden_dir, 'forward_cases.npz'))
    for D in (1, 3, 8, 10, 30, 64, 100, 128, 256):
        td = to_dev({n: z[f'D{D}|t|{n}'] for n in NAMES}, dev)
        t = tables_struct(td)
        for dtype in (torch.int64, torch.int32):
            users = torch.tensor(z[f'D{D}|users'], device=dev).to(dtype)
            items = torch.tensor(z[f'D{D}|items'], device=dev).to(dtype)
            for y, yr in ((0, None), (1, (1.0, 5.0))):
                out = ops.mf_score(t, users, items, yr).cpu().numpy()
                close(out, z[f'D{D}|y{y}|preds'])


def test_forward_out_of_range_ids_do_not_fault(golden_dir, dev):
    from collie_amd import ops
    z = np.load(os.path.join(golden_dir, 'forward_cases.npz'))
    td = to_dev({n: z[f'D64|t|{n}'] for n in NAMES}, dev)
    t = tables_struct(td)
    users = torch.tensor([0, 10 ** 9, 3, -1], device=dev)
    items = torch.tensor([1, 2, 10 ** 9, 0], device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    out = ops.mf_score(t, users, items, None, status)
    torch.cuda.synchronize()
    assert int(status.item()) == 1
    assert float(out[1]) == 0.0 and float(out[2]) == 0.0 and float(out[3]) == 0.0


def test_standalone_losses_vs_reference(golden_dir, dev):
    from collie_amd import ops
    z = np.load(os.path.join(golden_dir, 'loss_cases.npz'))
    codes, weights = meta_codes(z, dev)
    N = int(z['num_items'])
    for tag in z['cases']:
        kind, use_meta, variant = tag.split('|')
        sp = torch.tensor(z[tag + '|s_pos'], device=dev)
        sn = torch.tensor(z[tag + '|s_neg'], device=dev)
        cfg = ops.make_loss_cfg(kind, N, codes if use_meta == '1' else None,
                                weights if use_meta == '1' else None)
        pos = torch.tensor(z[tag + '|pos'], device=dev)
        neg = torch.tensor(z[tag + '|neg'], device=dev)
        loss, gp, gn = ops.loss_forward_backward(cfg, sp, sn, pos, neg)
        close(loss.item(), z[tag + '|loss'])
        scale = max(np.abs(z[tag + '|g_pos']).max(), 1e-12)
        close(gp.cpu().numpy(), z[tag + '|g_pos'], scale=scale)
        close(gn.cpu().numpy(), z[tag + '|g_neg'], scale=scale)


class Fused:
    """One fused forward on a fresh (zeroed) workspace, kept for the following update calls."""

    def __init__(self, td, users, pos, neg, kind, dev, codes=None, weights=None, need_grad=True,
                 ws=None, want_scores=True):
        from collie_amd import ops
        self.t = tables_struct(td)
        self.batch = ops.make_batch(users, pos, neg)
        B, K = neg.shape
        self.ws = ws or ops.TrainWorkspace(dev)
        self.buf = self.ws.ensure(B, K, self.t.dim, self.t.num_users, self.t.num_items)
        self.step_id = self.ws.next_step()
        cfg = ops.make_loss_cfg(kind, self.t.num_items, codes, weights)
        self.loss = torch.empty(1, device=dev)
        # without score outputs the dispatcher may pick the LDS-parked kernel variant
        self.s_pos = torch.empty(B, device=dev) if want_scores else None
        self.s_neg = torch.empty(K, B, device=dev) if want_scores else None
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.train_forward(self.t, self.batch, cfg, self.buf, need_grad, self.step_id, self.loss,
                          self.s_pos, self.s_neg, self.status)
        self.built = False

    def segments(self):
        from collie_amd import ops
        if not self.built:
            ops.train_build_segments(self.t, self.batch, self.buf)
            self.built = True

    def export(self, side, is_bias):
        from collie_amd import ops
        self.segments()
        return ops.train_export_grad(self.t, self.batch, self.buf, side, is_bias, self.step_id)

    def update(self, is_bias, dense, per_side):
        from collie_amd import ops
        self.segments()
        ops.train_update(self.t, self.batch, self.buf, is_bias, dense, self.step_id, per_side)


def test_fused_calculate_loss_and_grads_vs_reference(golden_dir, dev):
    from collie_amd import _lib, ops
    z = np.load(os.path.join(golden_dir, 'train_cases.npz'))
    codes, weights = meta_codes(z, dev)
    for tag in z['cases']:
        kind, D, K, m = tag.split('|')
        td = to_dev({n: z[f'{tag}|t|{n}'] for n in NAMES}, dev)
        users = torch.tensor(z[f'{tag}|users'], device=dev)      # int32, as the loaders emit
        pos = torch.tensor(z[f'{tag}|pos'], device=dev)          # int32
        neg = torch.tensor(z[f'{tag}|neg'], device=dev)          # int64 [B, K]
        kw = dict(codes=codes, weights=weights) if m == 'm1' else {}
        f = Fused(td, users, pos, neg, kind, dev, **kw)
        close(f.loss.item(), z[f'{tag}|loss'])
        close(f.s_pos.cpu().numpy(), z[f'{tag}|s_pos'])
        close(f.s_neg.cpu().numpy(), z[f'{tag}|s_neg'])
        assert int(f.status.item()) == 0
        gscale = max(np.abs(z[f'{tag}|g|item_embeddings|vals']).max(), 1e-12)
        for side, names in ((_lib.SIDE_USER, ('user_embeddings', 'user_biases')),
                            (_lib.SIDE_ITEM, ('item_embeddings', 'item_biases'))):
            for name in names:
                rows, vals = f.export(side, 'bias' in name)
                # touched-row set (zero-gradient rows included): exact
                np.testing.assert_array_equal(rows.cpu().numpy(), z[f'{tag}|g|{name}|rows'])
                ref = z[f'{tag}|g|{name}|vals']
                if name == 'user_biases':
                    # analytically zero; the reference holds only fp32 cancellation residue
                    assert np.abs(ref).max() <= 1e-6 * max(gscale, 1.0)
                    assert float(vals.abs().max()) == 0.0
                else:
                    close(vals.cpu().numpy(), ref, scale=gscale)


@pytest.mark.parametrize('kind,K,D', [('adaptive_bpr', 6, 64), ('warp', 5, 128), ('hinge', 1, 10)])
def test_fused_loss_with_many_metadata_arrays_vs_oracle(dev, kind, K, D):
    """``metadata_for_loss`` is a dict of any size in the reference
    (metadata_utils.py:60-71); the fused kernels take up to COLLIE_MAX_META = 16 arrays."""
    from oracle import mf_oracle as O
    from collie_amd import _lib, ops
    assert _lib.MAX_META == 16
    U, N, B, M = 300, 200, 512, 7
    rng = np.random.default_rng(K)
    t = O.init_tables(U, N, D, rng)
    t['user_embeddings'] *= 8
    t['item_embeddings'] *= 8
    t['item_biases'] = (rng.standard_normal((N, 1)) * 0.05).astype(np.float32)
    metadata = {f'm{i}': rng.integers(0, 2 + i, N) for i in range(M)}
    weights = {f'm{i}': 0.05 + 0.02 * i for i in range(M)}          # sum 0.77 <= 1
    users = rng.integers(0, U, B)
    pos = rng.integers(0, N, B)
    neg = rng.integers(0, N, (B, K))
    want = O.calculate_loss(t, users, pos, neg, kind, num_items=N, metadata=metadata,
                            metadata_weights=weights)
    td = to_dev({n: t[n] for n in NAMES}, dev)
    codes = [torch.tensor(metadata[f'm{i}'], dtype=torch.int32, device=dev) for i in range(M)]
    f = Fused(td, torch.tensor(users, dtype=torch.int32, device=dev),
              torch.tensor(pos, dtype=torch.int32, device=dev), torch.tensor(neg, device=dev), kind,
              dev, codes=codes, weights=[weights[f'm{i}'] for i in range(M)])
    close(f.loss.item(), want['loss'])
    gscale = max(np.abs(want['grads']['item_embeddings'][1]).max(), 1e-12)
    rows, vals = f.export(_lib.SIDE_ITEM, False)
    np.testing.assert_array_equal(rows.cpu().numpy(), want['grads']['item_embeddings'][0])
    close(vals.cpu().numpy(), want['grads']['item_embeddings'][1], scale=gscale)
    with pytest.raises(ValueError):
        ops.make_loss_cfg(kind, N, codes * 3, [0.01] * (3 * M))


ALGO_STATE = {'sparse_adam': ('exp_avg', 'exp_avg_sq'), 'adam': ('exp_avg', 'exp_avg_sq'),
              'adagrad': ('sum', None), 'sgd': (None, None)}


def test_optimizer_trajectories_vs_torch(golden_dir, dev):
    """4 training steps per optimizer combination; tables and optimizer state after every step."""
    from collie_amd import _lib, ops
    z = np.load(os.path.join(golden_dir, 'optimizer_cases.npz'))
    combos = json.loads(str(z['combos']))
    for tag, sparse, opt, bopt, lr, blr, wd, kind, K in combos:
        td = to_dev({n: z[f'{tag}|t0|{n}'] for n in NAMES}, dev)
        state = {n: {'s1': torch.zeros_like(td[n]), 's2': torch.zeros_like(td[n]),
                     'stamp': torch.zeros(td[n].shape[0], dtype=torch.int32, device=dev)}
                 for n in NAMES}
        ws = ops.TrainWorkspace(dev)
        for s in range(4):
            users = torch.tensor(z[f'{tag}|s{s}|users'], device=dev)
            pos = torch.tensor(z[f'{tag}|s{s}|pos'], device=dev)
            neg = torch.tensor(z[f'{tag}|s{s}|neg'], device=dev)
            f = Fused(td, users, pos, neg, kind, dev, ws=ws)
            close(f.loss.item(), z[f'{tag}|s{s}|loss'], rtol=2e-5)
            for is_bias in (False, True):
                per_side = {}
                for n in NAMES:
                    if ('bias' in n) != is_bias:
                        continue
                    algo, rate = (bopt, blr) if (is_bias and bopt is not None) else (opt, lr)
                    side = _lib.SIDE_USER if n.startswith('user') else _lib.SIDE_ITEM
                    eps = 1e-10 if algo == 'adagrad' else 1e-8
                    cfg = ops.optim_cfg(algo, s + 1, rate, eps=eps,
                                        weight_decay=0.0 if algo == 'sparse_adam' else wd)
                    st = state[n]
                    per_side[side] = (cfg, st['s1'] if algo != 'sgd' else None,
                                      st['s2'] if algo in ('adam', 'sparse_adam') else None,
                                      st['stamp'] if not sparse else None)
                f.update(is_bias, not sparse, per_side)
            for n in NAMES:
                if n == 'user_biases':
                    # d loss / d user_bias is analytically 0: exactly 0 here; torch holds fp32
                    # cancellation residue, which SGD scales by lr and Adam / Adagrad normalise
                    # into steps of up to lr each -- assert that bound on the reference values
                    assert float(td[n].abs().max()) == 0.0
                    balgo, brate = (bopt, blr) if bopt is not None else (opt, lr)
                    bound = 1e-6 if balgo == 'sgd' else (s + 1) * brate * 1.01
                    assert np.abs(z[f'{tag}|s{s}|t|{n}']).max() <= bound + wd * (s + 1) * brate
                    continue
                try:
                    close(td[n].cpu().numpy(), z[f'{tag}|s{s}|t|{n}'], rtol=2e-5)
                except AssertionError as e:
                    bad = np.nonzero(np.abs(td[n].cpu().numpy() - z[f'{tag}|s{s}|t|{n}']).max(axis=1)
                                     > 1e-4)[0]
                    raise AssertionError(f'{tag} step {s} table {n} rows {bad}: {e}')
                algo = bopt if ('bias' in n and bopt is not None) else opt
                for key, mine in zip(ALGO_STATE[algo], ('s1', 's2')):
                    if key is not None:
                        close(state[n][mine].cpu().numpy(), z[f'{tag}|s{s}|state|{n}|{key}'],
                              rtol=2e-5)


@pytest.mark.parametrize('B,K,D,U,N', [(1, 1, 64, 5, 7), (4099, 10, 64, 3000, 500),
                                        (70000, 10, 64, 138493, 26744), (513, 63, 32, 100, 4000),
                                        (2500, 5, 520, 300, 200), (3000, 3, 30, 70000, 90000)])
def test_fused_step_vs_oracle_seeded(dev, B, K, D, U, N):
    """Whole step (forward, coalesce, SparseAdam + SGD-bias update) vs the numpy oracle."""
    from oracle import mf_oracle as O
    from collie_amd import _lib, ops
    rng = np.random.default_rng(B + K)
    tables = O.init_tables(U, N, D, rng)
    tables['user_embeddings'] *= np.float32(4 * np.sqrt(D))
    tables['item_embeddings'] *= np.float32(4 * np.sqrt(D))
    tables['item_biases'] = rng.normal(0, 0.05, (N, 1)).astype(np.float32)
    kind = 'adaptive_bpr' if K > 1 else 'bpr'
    st = O.TrainState(tables, 'sparse_adam', 'sgd', lr=1e-2, bias_lr=1e-1, sparse=True)
    td = to_dev(tables, dev)
    m = {n: torch.zeros_like(td[n]) for n in NAMES}
    v = {n: torch.zeros_like(td[n]) for n in NAMES}
    ws = ops.TrainWorkspace(dev)
    for step in range(2):
        # Zipf-ish popularity so that hot rows have long segments
        users = (rng.zipf(1.3, B) % U).astype(np.int32)
        pos = (rng.zipf(1.2, B) % N).astype(np.int32)
        neg = rng.integers(0, N, (B, K)).astype(np.int64)
        out = st.step(users, pos, neg, kind)
        f = Fused(td, torch.tensor(users, device=dev), torch.tensor(pos, device=dev),
                  torch.tensor(neg, device=dev), kind, dev, ws=ws)
        close(f.loss.item(), out['loss'])
        close(f.s_pos.cpu().numpy(), out['s_pos'])
        close(f.s_neg.cpu().numpy(), out['s_neg'])
        per_side = {}
        for n in ('user_embeddings', 'item_embeddings'):
            side = _lib.SIDE_USER if n.startswith('user') else _lib.SIDE_ITEM
            rows, vals = f.export(side, False)
            np.testing.assert_array_equal(rows.cpu().numpy(), out['grads'][n][0])
            close(vals.cpu().numpy(), out['grads'][n][1], rtol=5e-5)
            per_side[side] = (ops.optim_cfg('sparse_adam', step + 1, 1e-2), m[n], v[n], None)
        f.update(False, False, per_side)
        sgd = ops.optim_cfg('sgd', step + 1, 1e-1)
        f.update(True, False, {_lib.SIDE_USER: (sgd, None, None, None),
                               _lib.SIDE_ITEM: (sgd, None, None, None)})
        for n in ('user_embeddings', 'item_embeddings', 'item_biases'):
            close(td[n].cpu().numpy(), st.t[n], rtol=5e-5)
            if 'bias' not in n:
                close(m[n].cpu().numpy(), st.state[n]['m'], rtol=5e-5)
                close(v[n].cpu().numpy(), st.state[n]['v'], rtol=5e-5)


@pytest.mark.parametrize('B,K,D,kind', [(3000, 10, 64, 'adaptive_bpr'), (3000, 20, 64, 'adaptive_hinge'),
                                        (2000, 10, 128, 'warp'), (1500, 1, 64, 'hinge'),
                                        (1500, 5, 32, 'warp'), (1000, 30, 128, 'adaptive_bpr'),
                                        (1200, 17, 256, 'warp'), (900, 3, 16, 'adaptive_hinge'),
                                        (800, 1, 512, 'bpr'), (800, 3, 512, 'adaptive_bpr'),
                                        (700, 3, 512, 'warp'), (1000, 10, 256, 'adaptive_bpr'),
                                        (1100, 15, 128, 'warp'), (900, 12, 256, 'adaptive_hinge')])
def test_fused_step_lds_parked_variants_vs_oracle(dev, B, K, D, kind):
    """The LDS-parked forward variants (one and two gather passes; picked when no score outputs
    are requested): loss, coalesced gradients and the updated tables vs the numpy oracle."""
    from oracle import mf_oracle as O
    from collie_amd import _lib, ops
    rng = np.random.default_rng(B + K + D)
    U, N = 700, 900
    tables = O.init_tables(U, N, D, rng)
    tables['user_embeddings'] *= np.float32(4 * np.sqrt(D))
    tables['item_embeddings'] *= np.float32(4 * np.sqrt(D))
    tables['item_biases'] = rng.normal(0, 0.05, (N, 1)).astype(np.float32)
    st = O.TrainState(tables, 'sparse_adam', 'sgd', lr=1e-2, bias_lr=1e-1, sparse=True)
    td = to_dev(tables, dev)
    m = {n: torch.zeros_like(td[n]) for n in NAMES}
    v = {n: torch.zeros_like(td[n]) for n in NAMES}
    ws = ops.TrainWorkspace(dev)
    for step in range(2):
        users = (rng.zipf(1.3, B) % U).astype(np.int32)
        pos = (rng.zipf(1.2, B) % N).astype(np.int32)
        neg = rng.integers(0, N, (B, K)).astype(np.int64)
        out = st.step(users, pos, neg, kind)
        f = Fused(td, torch.tensor(users, device=dev), torch.tensor(pos, device=dev),
                  torch.tensor(neg, device=dev), kind, dev, ws=ws, want_scores=False)
        close(f.loss.item(), out['loss'])
        assert int(f.status.item()) == 0
        per_side = {}
        for n in ('user_embeddings', 'item_embeddings'):
            side = _lib.SIDE_USER if n.startswith('user') else _lib.SIDE_ITEM
            rows, vals = f.export(side, False)
            np.testing.assert_array_equal(rows.cpu().numpy(), out['grads'][n][0])
            close(vals.cpu().numpy(), out['grads'][n][1], rtol=5e-5)
            per_side[side] = (ops.optim_cfg('sparse_adam', step + 1, 1e-2), m[n], v[n], None)
        f.update(False, False, per_side)
        sgd = ops.optim_cfg('sgd', step + 1, 1e-1)
        f.update(True, False, {_lib.SIDE_USER: (sgd, None, None, None),
                               _lib.SIDE_ITEM: (sgd, None, None, None)})
        # SparseAdam's first steps are sign-like (m / sqrt(v)): an element whose gradient is pure
        # rounding residue moves by O(lr) either way, hence the wider bar on the updated tables
        # (the gradients themselves were held to 5e-5 above)
        for n in ('user_embeddings', 'item_embeddings', 'item_biases'):
            close(td[n].cpu().numpy(), st.t[n], rtol=2e-4)


def test_fused_step_is_deterministic(dev):
    """Same inputs twice -> bit-identical tables and moments (no float atomics anywhere)."""
    from oracle import mf_oracle as O
    from collie_amd import _lib, ops
    rng = np.random.default_rng(0)
    U, N, D, B, K = 5000, 800, 64, 30000, 10
    tables = O.init_tables(U, N, D, rng)
    users = torch.tensor((rng.zipf(1.3, B) % U).astype(np.int32), device=dev)
    pos = torch.tensor((rng.zipf(1.2, B) % N).astype(np.int32), device=dev)
    neg = torch.tensor(rng.integers(0, N, (B, K)), device=dev)
    results = []
    for _ in range(2):
        td = to_dev(tables, dev)
        m = {n: torch.zeros_like(td[n]) for n in NAMES}
        v = {n: torch.zeros_like(td[n]) for n in NAMES}
        ws = ops.TrainWorkspace(dev)
        for step in range(3):
            f = Fused(td, users, pos, neg, 'adaptive_bpr', dev, ws=ws)
            loss = f.loss
            adam = ops.optim_cfg('sparse_adam', step + 1, 1e-2)
            f.update(False, False, {
                _lib.SIDE_USER: (adam, m['user_embeddings'], v['user_embeddings'], None),
                _lib.SIDE_ITEM: (adam, m['item_embeddings'], v['item_embeddings'], None)})
            sgd = ops.optim_cfg('sgd', step + 1, 1e-1)
            f.update(True, False, {_lib.SIDE_USER: (sgd, None, None, None),
                                   _lib.SIDE_ITEM: (sgd, None, None, None)})
        results.append([td[n].cpu().numpy().copy() for n in NAMES] + [loss.item()])
    for a, b in zip(*results):
        np.testing.assert_array_equal(a, b)
