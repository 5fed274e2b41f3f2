"""
GPU tests of the product API (``-m gpu``): MatrixFactorizationModel / fused optimizers / trainers,
driven exactly as ``CollieMinimalTrainer`` drives the reference (trainer.py:458-470), against the
numpy oracle fed with the same batches.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
NAMES = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def close(a, b, rtol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30))


def synth(U, N, nnz, seed, K=10):
    from collie_amd.interactions import Interactions
    rng = np.random.default_rng(seed)
    users = (rng.zipf(1.4, nnz) % U)
    items = (rng.zipf(1.2, nnz) % N)
    return Interactions(users=users, items=items, num_users=U, num_items=N, allow_missing_ids=True,
                        num_negative_samples=K, seed=seed, max_number_of_samples_to_consider=0)


def tables_of(model):
    return {n: getattr(model, n).weight.detach().cpu().numpy().copy() for n in NAMES}


def step(model, opts, batch, scale=None):
    for o in opts:
        o.zero_grad()
    loss = model.calculate_loss(batch)
    (loss if scale is None else loss * scale).backward()
    for i, o in enumerate(opts):
        model.optimizer_step(optimizer=o, optimizer_idx=i)
    return loss


COMBOS = [
    dict(sparse=True, optimizer='sparse_adam', bias_optimizer='sgd', loss='bpr', lr=1e-2, bias_lr=1e-1),
    dict(sparse=False, optimizer='adam', bias_optimizer='sgd', loss='adaptive', lr=1e-1, bias_lr=1e-2),
    dict(sparse=False, optimizer='adam', bias_optimizer='sgd', loss='adaptive', lr=1e-2, bias_lr=1e-2,
         weight_decay=1e-3),
    dict(sparse=True, optimizer='adagrad', bias_optimizer='infer', loss='warp', lr=5e-2, bias_lr='infer'),
    dict(sparse=False, optimizer='sgd', bias_optimizer=None, loss='hinge', lr=5e-1, bias_lr=None),
]


@pytest.mark.parametrize('cfg', COMBOS)
def test_model_training_trajectory_vs_oracle(dev, cfg):
    from oracle import mf_oracle as O
    from collie_amd.interactions import ApproximateNegativeSamplingInteractionsDataLoader
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.layers import MultiOptimizer
    inter = synth(400, 300, 6000, seed=1)
    loader = ApproximateNegativeSamplingInteractionsDataLoader(interactions=inter, batch_size=512,
                                                               shuffle=True)
    torch.manual_seed(3)
    model = MatrixFactorizationModel(train=loader, embedding_dim=16, lr_scheduler_func=None, **cfg)
    with torch.no_grad():
        model.user_embeddings.weight.mul_(10.0)
        model.item_embeddings.weight.mul_(10.0)
    bopt = cfg['bias_optimizer']
    oracle = O.TrainState(tables_of(model), optimizer=cfg['optimizer'],
                          bias_optimizer=cfg['optimizer'] if bopt == 'infer' else bopt,
                          lr=cfg['lr'], bias_lr=cfg['lr'] if cfg['bias_lr'] == 'infer' else cfg['bias_lr'],
                          weight_decay=cfg.get('weight_decay', 0.0), sparse=cfg['sparse'])
    model.to(dev).train()
    ret = model.configure_optimizers()
    opts = ret if isinstance(ret, list) else [ret]
    kind = model._loss_kind
    n = 0
    for epoch in range(2):
        for (u, p), neg in loader:
            loss = step(model, opts, ((u.to(dev), p.to(dev)), neg.to(dev)))
            ref = oracle.step(u.numpy(), p.numpy(), neg.numpy(), kind)
            close(loss.item(), ref['loss'], rtol=5e-5)
            n += 1
    model.check_status()
    got = tables_of(model)
    for name in NAMES:
        if name == 'user_biases':
            # analytically zero gradient: exactly 0 here (weight decay of 0 stays 0 too); the
            # oracle reproduces the reference's fp32 residue, bounded by n steps of at most lr
            assert float(np.abs(got[name]).max()) == 0.0
            balgo = cfg['optimizer'] if bopt in ('infer', None) else bopt
            blr = cfg['lr'] if cfg['bias_lr'] in ('infer', None) else cfg['bias_lr']
            bound = 1e-5 if balgo == 'sgd' else n * blr * 1.01
            assert float(np.abs(oracle.t[name]).max()) <= bound
            continue
        close(got[name], oracle.t[name], rtol=2e-4)     # n steps of accumulated fp32 rounding
    assert n == 2 * len(loader)


class _FixedBatches(list):
    def __init__(self, batches, U, N, K):
        super().__init__(batches)
        self.num_users, self.num_items, self.num_negative_samples = U, N, K


@pytest.mark.parametrize('tag,kw', [
    ('sparse_adam_bpr', dict(sparse=True, lr=1e-2, bias_lr=1e-2, optimizer='sparse_adam',
                             bias_optimizer='sgd', loss='bpr', lr_scheduler_func=None)),
    ('default_adam_hinge', dict(lr=1e-2))])            # every default of the reference
def test_trainer_epochs_vs_the_reference_trainer(golden_dir, dev, tag, kw):
    """F6: ``CollieMinimalTrainer.fit`` for two epochs on the fixed batches of
    ``ref_pipeline.npz`` against the per-step losses, epoch losses and final tables the
    REFERENCE'S OWN ``CollieMinimalTrainer`` + ``MatrixFactorizationModel`` produced on them."""
    import os
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.trainer import CollieMinimalTrainer
    r = np.load(os.path.join(golden_dir, 'ref_pipeline.npz'))
    nb = int(r[f'{tag}|n_batches'])
    batches = [((torch.tensor(r[f'{tag}|b{i}|users']), torch.tensor(r[f'{tag}|b{i}|pos'])),
                torch.tensor(r[f'{tag}|b{i}|neg'])) for i in range(nb)]
    U, D = r[f'{tag}|t0|user_embeddings'].shape
    N = r[f'{tag}|t0|item_embeddings'].shape[0]
    K = batches[0][1].shape[1]
    model = MatrixFactorizationModel(train=_FixedBatches(batches, U, N, K), embedding_dim=D, **kw)
    with torch.no_grad():
        for n in NAMES:
            getattr(model, n).weight.copy_(torch.tensor(r[f'{tag}|t0|{n}']))
    losses, epochs = [], {}
    inner = model.calculate_loss

    def recording(batch):
        loss = inner(batch)
        losses.append(loss.detach())
        return loss

    model.calculate_loss = recording
    trainer = CollieMinimalTrainer(model, max_epochs=2, gpus=1, verbosity=0,
                                   early_stopping_patience=None, enable_model_summary=False)
    inner_epoch = trainer._train_loop_single_epoch

    def epoch_loop(model_, epoch):
        epochs[epoch] = inner_epoch(model_, epoch)
        return epochs[epoch]

    trainer._train_loop_single_epoch = epoch_loop
    trainer.fit(model)
    assert model._loss_kind + '_loss' == str(r[f'{tag}|loss_name'])
    assert model.hparams.num_epochs_completed == 2
    close([float(x) for x in losses], r[f'{tag}|step_losses'], rtol=2e-5)
    close([epochs[1], epochs[2]], r[f'{tag}|epoch_losses'], rtol=2e-5)
    for n in NAMES:
        have = getattr(model, n).weight.detach().cpu().numpy()
        if n == 'user_biases':
            assert float(np.abs(have).max()) == 0.0
            assert float(np.abs(r[f'{tag}|t1|{n}']).max()) <= 1e-6      # SGD: lr x fp32 residue
        else:
            close(have, r[f'{tag}|t1|{n}'], rtol=2e-4)


def test_backward_honours_grad_output_and_validation_path(dev):
    from collie_amd.model import MatrixFactorizationModel

    class Shape:
        num_users, num_items, num_negative_samples = 100, 80, 5

    rng = np.random.default_rng(0)
    batch = ((torch.tensor(rng.integers(0, 100, 64), dtype=torch.int32, device=dev),
              torch.tensor(rng.integers(0, 80, 64), dtype=torch.int32, device=dev)),
             torch.tensor(rng.integers(0, 80, (64, 5)), device=dev))
    outs = []
    for scale in (None, 2.0):
        torch.manual_seed(0)
        m = MatrixFactorizationModel(train=Shape(), embedding_dim=8, sparse=True, lr=0.5,
                                     optimizer='sgd', bias_optimizer=None, loss='bpr',
                                     lr_scheduler_func=None).to(dev).train()
        w0 = m.item_embeddings.weight.detach().clone()
        opt = m.configure_optimizers()
        step(m, [opt], batch, scale)
        outs.append((m.item_embeddings.weight.detach() - w0).cpu().numpy())
    np.testing.assert_allclose(outs[1], 2 * outs[0], rtol=1e-3, atol=1e-7)  # deltas of O(1e-5) on O(0.1) weights
    # validation: no grad, nothing staged, same loss value
    m.eval()
    with torch.no_grad():
        v = m.calculate_loss(batch)
    assert not v.requires_grad and m._staged_step is None
    t = m.calculate_loss(batch)
    assert t.requires_grad and abs(t.item() - v.item()) < 1e-6
    # optimizer.step() without backward() is a no-op (nothing armed)
    before = m.item_embeddings.weight.detach().clone()
    opt.zero_grad()
    m.calculate_loss(batch)
    opt.step()
    assert torch.equal(before, m.item_embeddings.weight.detach())


def test_direct_backward_equals_the_autograd_engine_route(dev):
    """``loss.backward()`` on the fused loss arms the staged step without the autograd engine;
    ``torch.autograd.backward(loss)`` / ``loss.backward(gradient=1)`` go through the engine.
    Same tables afterwards, bit for bit; a loss tensor with a hook takes the engine route."""
    from collie_amd.model import MatrixFactorizationModel

    class Shape:
        num_users, num_items, num_negative_samples = 100, 80, 5

    rng = np.random.default_rng(4)
    batch = ((torch.tensor(rng.integers(0, 100, 64), dtype=torch.int32, device=dev),
              torch.tensor(rng.integers(0, 80, 64), dtype=torch.int32, device=dev)),
             torch.tensor(rng.integers(0, 80, (64, 5)), device=dev))
    seen = []

    def run(how):
        torch.manual_seed(0)
        m = MatrixFactorizationModel(train=Shape(), embedding_dim=8, sparse=True, lr=1e-2,
                                     optimizer='sparse_adam', bias_optimizer='sgd', loss='bpr',
                                     lr_scheduler_func=None).to(dev).train()
        opts = m.configure_optimizers()
        for _ in range(3):
            for o in opts:
                o.zero_grad()
            loss = m.calculate_loss(batch)
            assert loss.requires_grad and loss.grad_fn is not None and loss.dim() == 0
            if how == 'direct':
                loss.backward()
                assert m._staged_step.armed and m._staged_step.grad_scale is None
            elif how == 'engine':
                torch.autograd.backward(loss)
                assert m._staged_step.armed and m._staged_step.grad_scale is not None
            elif how == 'gradient':
                loss.backward(gradient=torch.ones((), device=dev))
                assert m._staged_step.grad_scale is not None
            else:
                loss.register_hook(lambda g: seen.append(float(g)))
                loss.backward()
            for i, o in enumerate(opts):
                m.optimizer_step(optimizer=o, optimizer_idx=i)
        return [getattr(m, n).weight.detach().clone() for n in NAMES]

    want = run('direct')
    for how in ('engine', 'gradient', 'hook'):
        for a, b in zip(want, run(how)):
            assert torch.equal(a, b), how
    assert seen == [1.0, 1.0, 1.0]


def test_generic_route_matches_fused_route(dev):
    """A custom callable loss goes through model(users, items) + autograd + torch's own
    SparseAdam.step (inherited); results must agree with the fused kernels."""
    from collie_amd import loss as L
    from collie_amd.model import MatrixFactorizationModel

    class Shape:
        num_users, num_items, num_negative_samples = 120, 90, 4

    rng = np.random.default_rng(2)
    batches = [((torch.tensor(rng.integers(0, 120, 200), dtype=torch.int32, device=dev),
                 torch.tensor(rng.integers(0, 90, 200), dtype=torch.int32, device=dev)),
                torch.tensor(rng.integers(0, 90, (200, 4)), device=dev)) for _ in range(3)]
    results = []
    for loss in ('bpr', lambda *a, **k: L.adaptive_bpr_loss(*a, **k)):
        torch.manual_seed(1)
        m = MatrixFactorizationModel(train=Shape(), embedding_dim=32, sparse=True, lr=1e-2,
                                     bias_lr=1e-1, optimizer='sparse_adam', bias_optimizer='sgd',
                                     loss=loss, lr_scheduler_func=None).to(dev).train()
        with torch.no_grad():
            m.user_embeddings.weight.mul_(20.0)
            m.item_embeddings.weight.mul_(20.0)
        opts = m.configure_optimizers()
        losses = [step(m, opts, b).item() for b in batches]
        results.append((losses, tables_of(m)))
    close(results[1][0], results[0][0])
    for n in ('user_embeddings', 'item_embeddings', 'item_biases'):
        close(results[1][1][n], results[0][1][n], rtol=1e-4)


def test_quickstart_ml100k_shape_trains_and_evaluates(dev):
    """BASELINE configs[0] shape (943 x 1682, D=10, adaptive hinge, dense Adam + SGD bias,
    ReduceLROnPlateau) through CollieMinimalTrainer + evaluate_in_batches: the loss must fall and
    MAP@10 on held-out positives must beat the untrained model by a wide margin."""
    from collie_amd.interactions import Interactions
    from collie_amd.metrics import evaluate_in_batches, mapk
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.trainer import CollieMinimalTrainer
    rng = np.random.default_rng(42)
    U, N, nnz = 943, 1682, 55375
    # low-rank ground truth so that there is something to learn
    pu, qi = rng.normal(size=(U, 6)), rng.normal(size=(N, 6))
    users = rng.integers(0, U, nnz * 3)
    logits = (pu[users] @ qi.T)
    items = np.array([rng.choice(N, p=np.exp(l - l.max()) / np.exp(l - l.max()).sum())
                      for l in logits[:nnz]])
    users = users[:nnz]
    mask = rng.random(nnz) < 0.8
    mk = dict(num_users=U, num_items=N, allow_missing_ids=True, seed=1,
              check_num_negative_samples_is_valid=False)
    train = Interactions(users=users[mask], items=items[mask], **mk)
    test = Interactions(users=users[~mask], items=items[~mask], **mk)
    torch.manual_seed(0)
    model = MatrixFactorizationModel(train=train, embedding_dim=10, lr=1e-1, loss='adaptive',
                                     optimizer='adam')
    before = evaluate_in_batches([mapk], test, model.to(dev), k=10, verbose=False)
    trainer = CollieMinimalTrainer(model, max_epochs=4, gpus=1, verbosity=0,
                                   enable_model_summary=False)
    trainer.fit(model)
    assert trainer.num_epochs_completed == 4
    after = evaluate_in_batches([mapk], test, model, k=10, verbose=False)
    assert after > 5 * before and after > 0.01, (before, after)
    # save / load on the device keeps predictions identical (tests/test_model.py:1346-1385)
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'm.pth')
        model.save_model(path)
        loaded = MatrixFactorizationModel(load_model_path=path, map_location='cpu').to(dev)
        a = model.get_item_predictions(user_id=3, sort_values=False)
        b = loaded.get_item_predictions(user_id=3, sort_values=False)
        assert (a.values == b.values).all()
    with pytest.raises(ValueError, match='is not in the model'):
        model.get_item_predictions(user_id=U)


def test_out_of_range_ids_raise_at_check(dev):
    from collie_amd.model import MatrixFactorizationModel

    class Shape:
        num_users, num_items, num_negative_samples = 10, 10, 2

    m = MatrixFactorizationModel(train=Shape(), embedding_dim=8, sparse=True, optimizer='sparse_adam',
                                 lr_scheduler_func=None).to(dev).train()
    opts = m.configure_optimizers()
    batch = ((torch.tensor([1, 50], dtype=torch.int32, device=dev),
              torch.tensor([2, 3], dtype=torch.int32, device=dev)),
             torch.tensor([[1, 2], [3, 400]], device=dev))
    step(m, opts, batch)
    with pytest.raises(IndexError):
        m.check_status()


@pytest.mark.parametrize('loss,y_range,sparse,opt,bias_opt,D', [
    ('mse', None, True, 'sparse_adam', 'sgd', 64), ('mae', (1.0, 5.0), True, 'adagrad', 'sgd', 30),
    ('mse', (0.0, 5.0), False, 'adam', 'sgd', 10), ('mae', None, True, 'sgd', 'infer', 128),
    ('mse', None, True, 'sparse_adam', 'sgd', 8), ('mse', (1.0, 5.0), True, 'adagrad', 'sgd', 4),
    ('mae', None, True, 'sparse_adam', 'sgd', 3)])
def test_explicit_fused_trajectory_vs_torch_reference_path(loss, y_range, sparse, opt, bias_opt, D):
    """Explicit feedback (base_pipeline.py:635-650): three steps of the fused HIP path against the
    reference's torch op sequence on the CPU (oracle/torch_path.py), same tables, same batches."""
    from oracle import torch_path as T
    from collie_amd.model import MatrixFactorizationModel
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(7)
    U, N, B = 200, 150, 700

    class ExplicitShape:
        num_users, num_items = U, N

    torch.manual_seed(0)
    model = MatrixFactorizationModel(train=ExplicitShape(), embedding_dim=D, sparse=sparse, lr=5e-2,
                                     bias_lr=5e-2 if bias_opt != 'infer' else 'infer', loss=loss,
                                     optimizer=opt, bias_optimizer=bias_opt, y_range=y_range,
                                     lr_scheduler_func=None)
    with torch.no_grad():
        model.user_embeddings.weight.mul_(30.0)
        model.item_embeddings.weight.mul_(30.0)
        model.user_biases.weight.normal_(0, 0.1)
        model.item_biases.weight.normal_(0, 0.1)
    tables = {n: getattr(model, n).weight.detach().numpy().copy()
              for n in ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')}
    ref = T.TorchMF(U, N, D, sparse=sparse, tables=tables, y_range=y_range)
    ref_opts = T.make_optimizers(ref, opt, opt if bias_opt == 'infer' else bias_opt, 5e-2, 5e-2)
    model.to(dev)
    model.train()
    opts = model.configure_optimizers()
    for step in range(3):
        users = (rng.zipf(1.4, B) % U).astype(np.int32)
        items = rng.integers(0, N, B).astype(np.int64)
        ratings = rng.integers(1, 6, B).astype(np.float64)
        want_loss, _ = T.explicit_train_step(
            ref, ref_opts, (torch.tensor(users), torch.tensor(items), torch.tensor(ratings)), loss)
        for o in opts:
            o.zero_grad()
        got = model.calculate_loss((torch.tensor(users, device=dev), torch.tensor(items, device=dev),
                                    torch.tensor(ratings, device=dev)))
        got.backward()
        for o in opts:
            o.step()
        np.testing.assert_allclose(got.item(), want_loss.item(), rtol=1e-5)
        for n, want in ref.tables().items():
            have = getattr(model, n).weight.detach().cpu().numpy()
            np.testing.assert_allclose(have, want, rtol=2e-4, atol=2e-4 * np.abs(want).max(),
                                       err_msg=f'step {step} table {n}')
    model.check_status()
    # evaluation-mode loss (no grad) and implicit/explicit mismatch errors keep the reference's form
    with torch.no_grad():
        v = model.calculate_loss((torch.tensor(users, device=dev), torch.tensor(items, device=dev),
                                  torch.tensor(ratings, device=dev)))
    assert v.dim() == 0 and torch.isfinite(v)
    with pytest.raises(ValueError):
        model.calculate_loss(((torch.tensor(users, device=dev), torch.tensor(items, device=dev)),
                              torch.tensor(items, device=dev)))


class _ExplicitBatches(list):
    def __init__(self, batches, U, N):
        super().__init__(batches)
        self.num_users, self.num_items = U, N


@pytest.mark.parametrize('tag,kw', [
    ('explicit_mse_yrange_adam', dict(lr=5e-2, bias_lr=5e-2, loss='mse', optimizer='adam',
                                      bias_optimizer='sgd', y_range=(0.0, 5.0),
                                      lr_scheduler_func=None)),
    ('explicit_mae_sparse_adagrad', dict(sparse=True, lr=5e-2, bias_lr=5e-2, loss='mae',
                                         optimizer='adagrad', bias_optimizer='sgd',
                                         lr_scheduler_func=None))])
def test_explicit_trainer_epochs_vs_the_reference_trainer(golden_dir, dev, tag, kw):
    """Explicit feedback (base_pipeline.py:635-650): two epochs of ``CollieMinimalTrainer`` on the
    fixed (users, items, ratings) batches of ``ref_pipeline.npz`` against the per-step losses and
    final tables the REFERENCE'S OWN model and trainer produced on them."""
    import os
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.trainer import CollieMinimalTrainer
    r = np.load(os.path.join(golden_dir, 'ref_pipeline.npz'))
    nb = int(r[f'{tag}|n_batches'])
    batches = [(torch.tensor(r[f'{tag}|b{i}|users']), torch.tensor(r[f'{tag}|b{i}|items']),
                torch.tensor(r[f'{tag}|b{i}|ratings'])) for i in range(nb)]
    U, D = r[f'{tag}|t0|user_embeddings'].shape
    N = r[f'{tag}|t0|item_embeddings'].shape[0]
    model = MatrixFactorizationModel(train=_ExplicitBatches(batches, U, N), embedding_dim=D, **kw)
    with torch.no_grad():
        for n in NAMES:
            getattr(model, n).weight.copy_(torch.tensor(r[f'{tag}|t0|{n}']))
    losses = []
    inner = model.calculate_loss

    def recording(batch):
        loss = inner(batch)
        losses.append(loss.detach())
        return loss

    model.calculate_loss = recording
    trainer = CollieMinimalTrainer(model, max_epochs=2, gpus=1, verbosity=0,
                                   early_stopping_patience=None, enable_model_summary=False)
    trainer.fit(model)
    assert model.hparams.num_epochs_completed == 2
    close([float(x) for x in losses], r[f'{tag}|step_losses'], rtol=2e-5)
    for n in NAMES:
        have = getattr(model, n).weight.detach().cpu().numpy()
        want = r[f'{tag}|t1|{n}']
        np.testing.assert_allclose(have, want, rtol=2e-4, atol=2e-4 * np.abs(want).max(), err_msg=n)
    model.check_status()


def test_explicit_loader_and_trainer_end_to_end():
    """ExplicitInteractions -> device loader -> CollieMinimalTrainer: MSE falls (base_pipeline.py
    :635-650 through the fused HIP path, ratings generated by a planted low-rank model)."""
    from collie_amd.interactions import DeviceInteractionsDataLoader, ExplicitInteractions
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.trainer import CollieMinimalTrainer
    rng = np.random.default_rng(3)
    U, N, nnz = 300, 200, 12000
    keys = rng.choice(U * N, nnz, replace=False)
    rows, cols = keys // N, keys % N
    pu, qi = rng.normal(0, 1, (U, 4)), rng.normal(0, 1, (N, 4))
    ratings = np.clip(np.round(3 + (pu[rows] * qi[cols]).sum(1)), 1, 5)
    data = ExplicitInteractions(users=rows, items=cols, ratings=ratings, num_users=U, num_items=N,
                                allow_missing_ids=True)
    loader = DeviceInteractionsDataLoader(data, batch_size=2048, shuffle=True, device='cuda:0')
    torch.manual_seed(0)
    model = MatrixFactorizationModel(train=loader, embedding_dim=8, lr=5e-2, loss='mse',
                                     sparse=True, optimizer='sparse_adam', y_range=(1.0, 5.0))
    trainer = CollieMinimalTrainer(model, max_epochs=1, gpus=1, verbosity=0, logger=None,
                                   early_stopping_patience=None)
    trainer.fit(model)
    with torch.no_grad():
        first = float(np.mean([model.calculate_loss(b).item() for b in loader]))
    trainer.max_epochs = 15
    trainer.fit(model)
    with torch.no_grad():
        last = float(np.mean([model.calculate_loss(b).item() for b in loader]))
    assert last < 0.6 * first


def test_similarity_and_user_predictions_vs_oracle():
    """item_item_similarity / user_user_similarity / get_user_predictions
    (base_pipeline.py:720-864) against the numpy restatement."""
    from oracle import mf_oracle as O
    from collie_amd.model import MatrixFactorizationModel
    dev = torch.device('cuda:0')

    class S:
        num_users, num_items, num_negative_samples = 321, 123, 5

    for D in (7, 30, 64):
        torch.manual_seed(D)
        model = MatrixFactorizationModel(train=S(), embedding_dim=D, lr_scheduler_func=None)
        with torch.no_grad():
            model.user_biases.weight.normal_(0, 0.1)
            model.item_biases.weight.normal_(0, 0.1)
        tables = {n: getattr(model, n).weight.detach().numpy().copy()
                  for n in ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')}
        model.to(dev).eval()
        for item in (0, 57, 122):
            got = model.item_item_similarity(item)
            want = O.embedding_similarity(tables['item_embeddings'], item)
            assert got.index[0] == item                       # base_pipeline.py:799-801
            np.testing.assert_allclose(got.sort_index().values, want, rtol=1e-5, atol=1e-6)
        got = model.user_user_similarity(300)
        np.testing.assert_allclose(got.sort_index().values,
                                   O.embedding_similarity(tables['user_embeddings'], 300),
                                   rtol=1e-5, atol=1e-6)
        preds = model.get_user_predictions(item_id=5, sort_values=False)
        want = O.mf_forward(tables, np.arange(321), np.full(321, 5))
        np.testing.assert_allclose(preds.values, want, rtol=1e-5, atol=1e-6)
        with pytest.raises(ValueError):
            model.item_item_similarity(123)
        with pytest.raises(ValueError):
            model.get_user_predictions(item_id=123)


def test_dropout_forward_backward_vs_oracle_and_statistics(dev):
    """``dropout_p > 0`` in training mode (matrix_factorization.py:143-147).  Given the mask stream
    (counter-based, so the oracle can replay it) scores and sparse gradient values equal the
    reference's op sequence to 1e-5; over many calls the scores are unbiased with the analytic
    variance (statistical parity with nn.Dropout); eval mode applies no dropout."""
    from oracle import mf_oracle as O
    from collie_amd import _lib, ops
    rng = np.random.default_rng(0)
    U, N, D, M, p = 40, 30, 24, 57, 0.3
    tables = O.init_tables(U, N, D, rng)
    tables['user_embeddings'] *= np.float32(12.0)
    tables['item_embeddings'] *= np.float32(12.0)
    tables['item_biases'] = rng.normal(0, 0.1, (N, 1)).astype(np.float32)
    td = {n: torch.tensor(v, device=dev) for n, v in tables.items()}
    t = _lib.make_tables(td['user_embeddings'], td['item_embeddings'], td['user_biases'],
                         td['item_biases'])
    users = rng.integers(0, U, M)
    items = rng.integers(0, N, M)
    ut, it = torch.tensor(users, device=dev), torch.tensor(items, device=dev)
    g = rng.normal(size=M).astype(np.float32)
    for seed, call in ((5, 1), (2 ** 40 + 7, 2 ** 33 + 3)):
        want = O.mf_forward_dropout(tables, users, items, p, seed, call, g)
        got = ops.mf_score(t, ut, it, None, None, (p, seed, call)).cpu().numpy()
        close(got, want[0], rtol=1e-5)
        gu, gi, gb = ops.mf_score_backward(t, ut, it, torch.tensor(g, device=dev), None, None,
                                           (p, seed, call))
        close(gu.cpu().numpy(), want[1], rtol=1e-5)
        close(gi.cpu().numpy(), want[2], rtol=1e-5)
        close(gb.cpu().numpy(), want[3], rtol=1e-6)
        # zeros exactly where either row's element was dropped
        assert np.array_equal(gu.cpu().numpy() == 0, want[1] == 0)
    # statistics over calls: E[score] = undropped score, Var = sum_d (u_d i_d)^2 (1/(1-p)^2 - 1)
    calls = 4000
    acc = torch.zeros(M, dtype=torch.float64, device=dev)
    acc2 = torch.zeros(M, dtype=torch.float64, device=dev)
    for c in range(calls):
        s = ops.mf_score(t, ut, it, None, None, (p, 11, c + 1)).double()
        acc += s
        acc2 += s * s
    mean = (acc / calls).cpu().numpy()
    var = (acc2 / calls).cpu().numpy() - mean ** 2
    plain = O.mf_forward(tables, users, items)
    prod = (tables['user_embeddings'][users] * tables['item_embeddings'][items]).astype(np.float64)
    var_want = (prod ** 2).sum(1) * (1.0 / (1 - p) ** 2 - 1.0)
    assert np.all(np.abs(mean - plain) <= 5 * np.sqrt(var_want / calls) + 1e-6)
    np.testing.assert_allclose(var, var_want, rtol=0.15)


def test_dropout_model_trains_and_eval_mode_is_plain(dev):
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.trainer import CollieMinimalTrainer
    runs = []
    for rep in range(2):
        inter = synth(400, 300, 6000, seed=2, K=5)      # its negative sampler is seeded too
        torch.manual_seed(7)
        m = MatrixFactorizationModel(train=inter, embedding_dim=16, dropout_p=0.25, sparse=True,
                                     optimizer='sparse_adam', loss='bpr', lr=5e-2,
                                     lr_scheduler_func=None)
        with torch.no_grad():
            m.user_embeddings.weight.mul_(5.0)
            m.item_embeddings.weight.mul_(5.0)
        tr = CollieMinimalTrainer(m, max_epochs=3, verbosity=0, enable_model_summary=False,
                                  early_stopping_patience=None)
        first = []
        inner = tr._train_loop_single_epoch
        tr._train_loop_single_epoch = lambda model, epoch: first.append(inner(model, epoch)) or first[-1]
        tr.fit(m)
        runs.append((first, m.item_embeddings.weight.detach().cpu().numpy().copy()))
        assert first[-1] < first[0]                  # it learns
    # eval mode: no dropout, forward equals the plain score kernel; train mode differs
    m.eval()
    u = torch.arange(50, device=dev)
    i = torch.arange(50, device=dev)
    with torch.no_grad():
        a, b = m(u, i), m(u, i)
        m.train()
        c = m(u, i)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # the loader's negatives come from its own numpy generator (seeded); masks from the model's
    # seed drawn under torch.manual_seed: the two runs are identical
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
