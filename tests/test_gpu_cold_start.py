"""
GPU tests (``-m gpu``) of ``ColdStartModel`` (SURVEY.md §8f row 3;
``collie/model/cold_start_matrix_factorization.py`` + ``multi_stage_pipeline.py``): both stages run
the fused MF kernels, so each stage's trajectory is checked against the numpy oracle fed with what
that stage's ``forward`` looks up — bucket rows of ``item_buckets[item]`` in stage ``item_buckets``
(``:304-309``), the items' own rows after ``advance_stage()`` copied them over (``:213-241``).
Tolerances as in tests/test_gpu_model.py (5e-5 losses, 2e-4 tables after several steps).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def close(a, b, rtol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30))


def close_adam(a, b, lr, steps, rtol=2e-4):
    """Tables after `steps` Adam steps: Adam normalises the gradient, so the few entries whose
    gradient nearly cancels (WARP right after the bucket copy: positive and violating negative often
    share a row) turn fp32 rounding differences into a fraction of an lr-sized step on both sides.
    Bound: every entry within steps * lr, at least 99 % of them within `rtol` of the largest value."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    diff = np.abs(a - b)
    assert diff.max() <= 1.2 * steps * lr
    assert (diff > rtol * np.abs(b).max()).mean() <= 0.01


class Shape:
    num_users, num_items, num_negative_samples = 120, 90, 5


def _step(model, opts, batch):
    for o in opts:
        o.zero_grad()
    loss = model.calculate_loss(batch)
    loss.backward()
    for i, o in enumerate(opts):
        model.optimizer_step(optimizer=o, optimizer_idx=i)
    return loss


@pytest.mark.parametrize('loss', ['hinge', 'warp'])
def test_cold_start_two_stage_trajectory_vs_oracle(dev, loss):
    from collie_amd.cold_start import ColdStartModel
    from collie_amd.metrics import evaluate_in_batches, mapk
    from collie_amd.interactions import Interactions
    from oracle import mf_oracle as O
    U, N, K, B, NB = Shape.num_users, Shape.num_items, 5, 128, 12
    rng = np.random.default_rng(3)
    buckets = rng.integers(0, NB, N)
    buckets[:NB] = np.arange(NB)              # every bucket used, ids start at 0
    torch.manual_seed(0)
    model = ColdStartModel(train=Shape(), item_buckets=buckets, embedding_dim=16, loss=loss,
                           item_buckets_stage_lr=2e-2, no_buckets_stage_lr=1e-2,
                           lr_scheduler_func=None)
    with torch.no_grad():
        model.item_bucket_embeddings.weight.mul_(10.0)
        model.user_embeddings.weight.mul_(10.0)
    model.to(dev).train()
    opts = model.configure_optimizers()
    assert len(opts) == 2 and model.hparams.stage == 'item_buckets'
    kind = model._loss_kind

    def np_of(*mods):
        return [m.weight.detach().cpu().numpy().copy() for m in mods]

    def batch_():
        return (rng.integers(0, U, B).astype(np.int32), rng.integers(0, N, B).astype(np.int32),
                rng.integers(0, N, (B, K)))

    def to_dev(u, p, n):
        return ((torch.tensor(u, device=dev), torch.tensor(p, device=dev)), torch.tensor(n, device=dev))

    # ---- stage 1: items are represented by their buckets -------------------------------------
    ue, be, ub, bb = np_of(model.user_embeddings, model.item_bucket_embeddings, model.user_biases,
                           model.item_bucket_biases)
    oracle = O.TrainState({'user_embeddings': ue, 'item_embeddings': be, 'user_biases': ub,
                           'item_biases': bb}, optimizer='adam', bias_optimizer=None, lr=2e-2,
                          bias_lr=None, sparse=False)
    items_before = np_of(model.item_embeddings)[0]
    for _ in range(3):
        u, p, n = batch_()
        got = _step(model, opts, to_dev(u, p, n))
        ref = oracle.step(u, buckets[p], buckets[n], kind, num_items=N)
        close(got.item(), ref['loss'], 5e-5)
    close(np_of(model.item_bucket_embeddings)[0], oracle.t['item_embeddings'], 2e-4)
    close(np_of(model.user_embeddings)[0], oracle.t['user_embeddings'], 2e-4)
    close(np_of(model.item_bucket_biases)[0], oracle.t['item_biases'], 2e-4)
    # the second stage's tables were not touched by the first stage's optimizer
    np.testing.assert_array_equal(np_of(model.item_embeddings)[0], items_before)
    # d loss / d user_bias is analytically 0 for pairwise losses: the HIP path keeps it an exact 0,
    # the oracle (like the reference) feeds Adam its fp32 residue (DESIGN.md, K3b+K4) — score with
    # the product's user biases
    oracle.t['user_biases'] = np_of(model.user_biases)[0]
    # forward() in the bucket stage == MF forward on the bucket rows
    uu, ii = rng.integers(0, U, 50), rng.integers(0, N, 50)
    with torch.no_grad():
        s = model(torch.tensor(uu, device=dev), torch.tensor(ii, device=dev)).cpu().numpy()
    close(s, O.mf_forward(oracle.t, uu, buckets[ii]), 1e-5)
    # evaluation in the bucket stage takes the generic route (scores through forward)
    inter = Interactions(users=rng.integers(0, U, 400), items=rng.integers(0, N, 400), num_users=U,
                         num_items=N, allow_missing_ids=True, num_negative_samples=K, seed=1,
                         max_number_of_samples_to_consider=0)
    model.eval()
    users = np.unique(inter.mat.row)
    scores = O.mf_forward(oracle.t, np.repeat(users, N), buckets[np.tile(np.arange(N), len(users))])
    want = O.mapk(inter.mat.tocsr(), users, scores.reshape(len(users), N), 5)
    close(evaluate_in_batches([mapk], inter, model, k=5, batch_size=1000), want, 1e-5)
    model.train()

    # ---- stage 2: every item starts from its bucket's row, then trains on its own -----------
    model.advance_stage()
    assert model.hparams.stage == 'no_buckets'
    be, bb = np_of(model.item_bucket_embeddings, model.item_bucket_biases)
    ie, ib, ue, ub = np_of(model.item_embeddings, model.item_biases, model.user_embeddings,
                           model.user_biases)
    np.testing.assert_array_equal(ie, be[buckets])
    np.testing.assert_array_equal(ib, bb[buckets])
    oracle = O.TrainState({'user_embeddings': ue, 'item_embeddings': ie, 'user_biases': ub,
                           'item_biases': ib}, optimizer='adam', bias_optimizer=None, lr=1e-2,
                          bias_lr=None, sparse=False)
    for _ in range(3):
        u, p, n = batch_()
        got = _step(model, opts, to_dev(u, p, n))
        ref = oracle.step(u, p, n, kind, num_items=N)
        close(got.item(), ref['loss'], 5e-5)
    close_adam(np_of(model.item_embeddings)[0], oracle.t['item_embeddings'], 1e-2, 3)
    close_adam(np_of(model.user_embeddings)[0], oracle.t['user_embeddings'], 1e-2, 3)
    np.testing.assert_array_equal(np_of(model.item_bucket_embeddings)[0], be)   # stage 1 tables frozen now
    model.check_status()
    with pytest.raises(ValueError, match='final stage'):
        model.advance_stage()
    # fused catalog evaluation is available again and agrees with the oracle
    model.eval()
    oracle.t['user_biases'] = np_of(model.user_biases)[0]
    scores = O.get_preds(oracle.t, users, N)
    want = O.mapk(inter.mat.tocsr(), users, scores, 5)
    close(evaluate_in_batches([mapk], inter, model, k=5, batch_size=1000), want, 1e-5)
    sims = model.item_bucket_item_similarity(3)
    e = oracle.t['item_embeddings'].astype(np.float64)
    b3 = be[3].astype(np.float64)
    want_sims = (e @ b3) / (np.linalg.norm(e, axis=1) * np.linalg.norm(b3))
    close(sims.sort_index().to_numpy(), want_sims, 1e-5)


class _FixedBatches(list):
    def __init__(self, batches, U, N, K):
        super().__init__(batches)
        self.num_users, self.num_items, self.num_negative_samples = U, N, K


@pytest.mark.parametrize('tag,kw', [
    ('adam_hinge', dict(item_buckets_stage_lr=1e-2, no_buckets_stage_lr=1e-2)),
    ('adagrad_sgd_bpr', dict(item_buckets_stage_lr=5e-2, no_buckets_stage_lr=1e-1,
                             item_buckets_stage_optimizer='adagrad',
                             no_buckets_stage_optimizer='sgd', loss='bpr'))])
def test_cold_start_two_stages_vs_the_reference_model_and_trainer(dev, golden_dir, tag, kw):
    """``ColdStartModel`` + ``CollieMinimalTrainer`` against ``tests/golden/ref_cold_start.npz``,
    which the REFERENCE'S OWN ``ColdStartModel`` and trainer produced on the same fixed batches
    (tests/golden/make_golden_ref_cold_start.py): one epoch in stage ``item_buckets``,
    ``advance_stage()``, one epoch in stage ``no_buckets`` -- per-step losses and every table."""
    import os
    from collie_amd.cold_start import ColdStartModel
    from collie_amd.trainer import CollieMinimalTrainer
    r = np.load(os.path.join(golden_dir, 'ref_cold_start.npz'))
    tables = ('user_embeddings', 'item_bucket_embeddings', 'item_embeddings', 'user_biases',
              'item_bucket_biases', 'item_biases')
    nb = int(r[f'{tag}|n_batches'])
    batches = [((torch.tensor(r[f'{tag}|b{i}|users']), torch.tensor(r[f'{tag}|b{i}|pos'])),
                torch.tensor(r[f'{tag}|b{i}|neg'])) for i in range(nb)]
    U, D = r[f'{tag}|t0|user_embeddings'].shape
    N = r[f'{tag}|t0|item_embeddings'].shape[0]
    K = batches[0][1].shape[1]
    model = ColdStartModel(train=_FixedBatches(batches, U, N, K),
                           item_buckets=r[f'{tag}|item_buckets'], embedding_dim=D,
                           lr_scheduler_func=None, **kw)
    with torch.no_grad():
        for n in tables:
            getattr(model, n).weight.copy_(torch.tensor(r[f'{tag}|t0|{n}']))
    losses = []
    inner = model.calculate_loss

    def recording(batch):
        loss = inner(batch)
        losses.append(loss.detach())
        return loss

    model.calculate_loss = recording
    trainer = CollieMinimalTrainer(model, max_epochs=1, gpus=1, verbosity=0,
                                   early_stopping_patience=None, enable_model_summary=False)
    trainer.fit(model)
    assert model._loss_kind + '_loss' == str(r[f'{tag}|loss_name'])
    lr1, lr2 = kw['item_buckets_stage_lr'], kw['no_buckets_stage_lr']

    def check(stage_key, lr, steps):
        for n in tables:
            have = getattr(model, n).weight.detach().cpu().numpy()
            want = r[f'{tag}|{stage_key}|{n}']
            if n == 'user_biases':
                # d loss / d user_bias is analytically 0: the HIP path keeps an exact 0, the
                # reference feeds its optimizer the fp32 residue (DESIGN.md, K3b+K4)
                assert float(np.abs(have).max()) == 0.0
                assert float(np.abs(want).max()) <= 1.2 * steps * lr
            else:
                # Adam and Adagrad normalise the gradient: the few entries whose gradient nearly
                # cancels turn fp32 rounding differences into a fraction of an lr-sized step on
                # both sides (and the SGD stage carries them along)
                close_adam(have, want, lr, steps)

    check('t1', lr1, nb)
    model.advance_stage()
    check('t1_advanced', lr1, nb)
    trainer.max_epochs += 1
    trainer.fit(model)
    check('t2', max(lr1, lr2), 2 * nb)
    close([float(x) for x in losses], r[f'{tag}|step_losses'], rtol=5e-5)
    model.check_status()
