"""
CPU tests (no GPU): the C-ABI library loads and exports every symbol ``include/collie_hip.h``
declares, and the host-side mirror of the reference API behaves like the reference where no
arithmetic is involved (batch grammar, loss / optimizer resolution, trainer control flow, loader
batch format).  Compute entry points are NOT called here: they need the MI355X and there is no
CPU fallback (asserted below).
"""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'collie_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(collie_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from collie_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f'{name} declared in collie_hip.h but not exported'
        assert name in _lib.SIGNATURES, f'{name} has no ctypes prototype in collie_amd/_lib.py'
    assert sorted(_lib.SIGNATURES) == names
    assert lib.collie_version() >= 100
    # size queries are pure host functions
    assert lib.collie_mf_train_workspace_bytes(1024, 10, 64, 1000, 500) > 1024 * 64 * 4 * 2
    assert lib.collie_mf_train_workspace_bytes(0, 10, 64, 1000, 500) == 0
    assert lib.collie_sparse_rows_workspace_bytes(1000, 64) > 1000 * 4 * 6
    assert lib.collie_mf_eval_topk_capacity() >= 10


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    from collie_amd import _lib
    lib = _lib.load()
    t = _lib.Tables(0, 0, 0, 0, 10, 10, 8)
    rc = lib.collie_mf_score(ctypes.byref(t), None, 0, None, 0, 5, 0, 0.0, 0.0, None, None, None)
    assert rc != 0 and b'NULL' in lib.collie_last_error()
    b = _lib.Batch(1, 1, 1, 0, 0, 1, 64, 4, 64, 1)
    t2 = _lib.Tables(1, 1, 1, 1, 10, 10, 8)
    cfg = _lib.LossCfg()
    cfg.kind = _lib.LOSS_ADAPTIVE_BPR
    rc = lib.collie_mf_train_forward(ctypes.byref(t2), ctypes.byref(b), ctypes.byref(cfg), None, 0,
                                     1, 1, 1, None, None, None, None)
    assert rc != 0 and b'num_negative_samples' in lib.collie_last_error()


class Shape:
    num_users, num_items, num_negative_samples = 30, 20, 10


class Shape1(Shape):
    num_negative_samples = 1


def test_no_cpu_fallback():
    from collie_amd._lib import CollieHipError
    from collie_amd.model import MatrixFactorizationModel
    model = MatrixFactorizationModel(train=Shape(), embedding_dim=8)
    users = torch.zeros(4, dtype=torch.long)
    with pytest.raises(CollieHipError, match='no CPU fallback'):
        model(users, users)
    with pytest.raises(CollieHipError):
        model.calculate_loss(((users, users), torch.zeros(4, 10, dtype=torch.long)))


def test_model_parameter_names_shapes_and_init():
    """matrix_factorization.py:106-117, layers.py:6-17."""
    from collie_amd.model import MatrixFactorizationModel
    torch.manual_seed(0)
    m = MatrixFactorizationModel(train=Shape(), embedding_dim=16, sparse=True)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        'user_biases.weight': (30, 1), 'item_biases.weight': (20, 1),
        'user_embeddings.weight': (30, 16), 'item_embeddings.weight': (20, 16)}
    assert float(sd['user_biases.weight'].abs().max()) == 0.0
    std = float(sd['user_embeddings.weight'].std())
    assert 0.6 / 40 < std < 1.5 / 40                       # 1 / (2.5 * 16) = 0.025
    assert m.user_embeddings.sparse and m.hparams.num_users == 30
    assert m.hparams.num_epochs_completed == 0 and m.hparams._is_implicit is True


def test_loss_resolution_follows_reference():
    """base_pipeline.py:277-340."""
    from collie_amd.model import MatrixFactorizationModel as MF
    assert MF(train=Shape(), loss='hinge')._loss_kind == 'adaptive_hinge'
    assert MF(train=Shape(), loss='adaptive')._loss_kind == 'adaptive_hinge'
    assert MF(train=Shape(), loss='bpr')._loss_kind == 'adaptive_bpr'
    assert MF(train=Shape(), loss='warp')._loss_kind == 'warp'
    assert MF(train=Shape1(), loss='bpr')._loss_kind == 'bpr'
    with pytest.warns(UserWarning, match='Using standard hinge loss'):
        assert MF(train=Shape1(), loss='adaptive_hinge')._loss_kind == 'hinge'
    with pytest.raises(ValueError, match='Cannot use WARP loss with a single negative sample!'):
        MF(train=Shape1(), loss='warp')
    with pytest.raises(ValueError, match='is not a valid loss function'):
        MF(train=Shape(), loss='nope')
    with pytest.raises(TypeError, match='``train`` must be provided'):
        MF()
    custom = MF(train=Shape(), loss=lambda *a, **k: None)
    assert custom._loss_kind is None and callable(custom.loss_function)


def test_batch_grammar_errors():
    """base_pipeline.py:603-605,636-637,651-652."""
    from collie_amd.model import MatrixFactorizationModel as MF
    u = torch.zeros(4, dtype=torch.long)
    implicit = MF(train=Shape(), loss='bpr')
    with pytest.raises(ValueError, match='Implicit loss with explicit data is invalid!'):
        implicit.calculate_loss((u, u, u.float()))
    with pytest.raises(ValueError, match='Unexpected format for batch'):
        implicit.calculate_loss((u,))

    class Explicit:
        num_users, num_items = 30, 20
    explicit = MF(train=Explicit(), loss='mse')
    with pytest.raises(ValueError, match='Explicit loss with implicit data is invalid!'):
        explicit.calculate_loss(((u, u), torch.zeros(4, 3, dtype=torch.long)))
    with pytest.raises(ValueError, match='num_negative_samples'):
        MF(train=Explicit(), loss='bpr')


def test_configure_optimizers_shapes_and_routing():
    """base_pipeline.py:342-479: name-based bias routing, scheduler dicts, return shapes."""
    from collie_amd import optim
    from collie_amd.model import MatrixFactorizationModel as MF
    m = MF(train=Shape(), sparse=True, optimizer='sparse_adam', bias_optimizer='sgd', lr=3e-3,
           bias_lr=7e-2)
    opts, scheds = m.configure_optimizers()
    assert isinstance(opts[0], optim.FusedSparseAdam) and isinstance(opts[0], torch.optim.SparseAdam)
    assert isinstance(opts[1], optim.FusedSGD) and isinstance(opts[1], torch.optim.SGD)
    emb = {id(m.user_embeddings.weight), id(m.item_embeddings.weight)}
    bias = {id(m.user_biases.weight), id(m.item_biases.weight)}
    assert {id(p) for p in opts[0].param_groups[0]['params']} == emb
    assert {id(p) for p in opts[1].param_groups[0]['params']} == bias
    assert opts[0].param_groups[0]['lr'] == 3e-3 and opts[1].param_groups[0]['lr'] == 7e-2
    assert scheds[0]['monitor'] == 'train_loss_epoch'
    assert isinstance(scheds[0]['scheduler'], torch.optim.lr_scheduler.ReduceLROnPlateau)

    m2 = MF(train=Shape(), bias_optimizer=None, lr_scheduler_func=None)
    single = m2.configure_optimizers()
    assert isinstance(single, optim.FusedAdam) and len(single.param_groups[0]['params']) == 4

    m3 = MF(train=Shape(), bias_optimizer='infer', bias_lr='infer', optimizer='adagrad',
            lr_scheduler_func=None, weight_decay=1e-4)
    both = m3.configure_optimizers()
    assert isinstance(both, list) and all(isinstance(o, optim.FusedAdagrad) for o in both)
    assert both[1].param_groups[0]['lr'] == m3.hparams.lr
    assert both[0].param_groups[0]['weight_decay'] == 1e-4

    with pytest.raises(ValueError, match='is not a valid optimizer!'):
        MF(train=Shape(), optimizer='lbfgs').configure_optimizers()
    # any torch.optim class is accepted as a callable (generic route)
    m4 = MF(train=Shape(), optimizer=torch.optim.RMSprop, bias_optimizer=None, lr_scheduler_func=None)
    assert isinstance(m4.configure_optimizers(), torch.optim.RMSprop)
    with pytest.warns(UserWarning, match='weight_decay'):
        m5 = MF(train=Shape(), sparse=True, weight_decay=0.1)
    assert m5.hparams.weight_decay == 0.0


def test_save_and_load_round_trip(tmp_path):
    """base_pipeline.py:245-257,880-900."""
    from collie_amd.model import MatrixFactorizationModel as MF
    m = MF(train=Shape(), embedding_dim=12, y_range=(1.0, 5.0))
    path = tmp_path / 'model.pth'
    m.save_model(path)
    m2 = MF(load_model_path=str(path))
    assert not m2.training and m2.hparams.embedding_dim == 12 and tuple(m2.hparams.y_range) == (1.0, 5.0)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    raw = torch.load(str(path), weights_only=False)
    assert set(raw) == {'state_dict', 'hparams'}


def test_loader_batch_format_matches_reference():
    """users/pos int32 [B], negatives int64 [B, K] (datasets.py:382-388,439-443)."""
    from collie_amd.interactions import (ApproximateNegativeSamplingInteractionsDataLoader,
                                         Interactions, InteractionsDataLoader)
    rng = np.random.default_rng(0)
    users, items = rng.integers(0, 50, 700), rng.integers(0, 40, 700)
    inter = Interactions(users=users, items=items, num_negative_samples=7, allow_missing_ids=True,
                         num_users=50, num_items=40, seed=3)
    assert inter.num_interactions == len(set(zip(users.tolist(), items.tolist())))
    for cls in (InteractionsDataLoader, ApproximateNegativeSamplingInteractionsDataLoader):
        loader = cls(interactions=inter, batch_size=256, shuffle=True)
        assert (loader.num_users, loader.num_items, loader.num_negative_samples) == (50, 40, 7)
        seen = 0
        for (u, p), n in loader:
            assert u.dtype == torch.int32 and p.dtype == torch.int32 and n.dtype == torch.int64
            assert n.shape == (len(u), 7) and int(n.min()) >= 0 and int(n.max()) < 40
            seen += len(u)
        assert seen == inter.num_interactions and len(loader) == -(-seen // 256)
    # exact negative sampling never returns one of the user's positives
    inter.max_number_of_samples_to_consider = 200
    pos = set(zip(inter.mat.row.tolist(), inter.mat.col.tolist()))
    for (u, p), n in InteractionsDataLoader(interactions=inter, batch_size=128):
        for a, row in zip(u.tolist(), n.tolist()):
            assert all((a, b) not in pos for b in row)


class FakeModel(torch.nn.Module):
    """Scripted epoch losses: exercises the trainer's control flow without any kernels."""

    def __init__(self, losses):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.losses = list(losses)
        self.hparams = type('H', (dict,), {'__getattr__': dict.__getitem__,
                                            '__setattr__': dict.__setitem__})(num_epochs_completed=0)
        self.steps = 0

    def train_dataloader(self):
        return [0, 1, 2]

    def val_dataloader(self):
        return None

    def _move_any_external_data_to_device(self):
        pass

    def configure_optimizers(self):
        opt = torch.optim.SGD(self.parameters(), lr=1.0)
        return ([opt], [{'scheduler': torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=0),
                         'monitor': 'train_loss_epoch'}])

    def calculate_loss(self, batch):
        self.steps += 1
        return self.w.sum() * 0 + self.losses[min(self.hparams.num_epochs_completed,
                                                   len(self.losses) - 1)]

    def optimizer_step(self, optimizer=None, **kw):
        optimizer.step()


def test_minimal_trainer_control_flow(capsys):
    """trainer.py:314-402: early stopping rule, resume from num_epochs_completed, plateau LR."""
    from collie_amd.trainer import CollieMinimalTrainer, CollieTrainer
    model = FakeModel([5.0, 4.0, 4.5, 4.6, 4.7, 1.0])
    trainer = CollieMinimalTrainer(model, max_epochs=10, gpus=0, early_stopping_patience=2,
                                   verbosity=0, enable_model_summary=False)
    trainer._move_batch_to_device = lambda b: b
    trainer.fit(model)
    # best epoch 2 (4.0); epoch 4 >= 2 + patience -> stop after epoch 4
    assert trainer.num_epochs_completed == 4 and model.hparams.num_epochs_completed == 4
    assert 'Early stopping activated' in capsys.readouterr().out
    assert trainer.best_epoch_loss == (2, 4.0)
    lr = trainer.optimizer.optimizers[0].param_groups[0]['lr']
    assert lr < 1.0                                   # ReduceLROnPlateau stepped with the loss
    # re-entrant fit continues from the completed epochs (tests/test_model.py:263-279)
    trainer.early_stopping_patience = None
    trainer.max_epochs = 6
    trainer.fit(model)
    assert trainer.num_epochs_completed == 6 and model.steps == 6 * 3

    class Logger:
        def __init__(self):
            self.calls = []

        def log_hyperparams(self, h):
            self.calls.append('hp')

        def log_metrics(self, metrics, step):
            self.calls.append((tuple(metrics), step))

        def save(self):
            self.calls.append('save')

        def finalize(self, status):
            self.calls.append(status)

    log = Logger()
    m2 = FakeModel([1.0])
    t2 = CollieTrainer(m2, max_epochs=2, gpus=0, logger=log, verbosity=0, log_every_n_steps=2,
                       enable_model_summary=False, enable_checkpointing=False, callbacks=[])
    t2._move_batch_to_device = lambda b: b
    t2.fit(m2)
    assert t2.ignored_kwargs == {'enable_checkpointing': False, 'callbacks': []}
    assert log.calls[0] == 'hp' and log.calls[-1] == 'FINISHED'
    assert (('train_loss_epoch',), 1) in log.calls and (('train_loss_step',), 2) in log.calls
    assert t2.max_epochs == 2


def test_explicit_interactions_batch_format_matches_reference():
    """datasets.py:448-562: non-nested (users, items, ratings) batches, no num_negative_samples."""
    import numpy as np
    import pytest
    from collie_amd.interactions import ExplicitInteractions, InteractionsDataLoader
    with pytest.raises(ValueError):
        ExplicitInteractions(users=[0, 1], items=[0, 1])
    e = ExplicitInteractions(users=[0, 1, 2, 2, 2], items=[1, 0, 2, 1, 1], ratings=[5, 3, 1, 4, 2])
    assert e.num_interactions == 4 and (e.num_users, e.num_items) == (3, 3)   # duplicate removed
    assert (e.min_rating, e.max_rating) == (1, 5)
    with pytest.raises(AttributeError):
        e.num_negative_samples
    loader = InteractionsDataLoader(e, batch_size=3)
    assert not hasattr(loader, 'num_negative_samples')
    batches = list(loader)
    assert [len(b) for b in batches] == [3, 3] and batches[0][0].dtype.is_floating_point is False
    np.testing.assert_array_equal(np.concatenate([b[2].numpy() for b in batches]), [5, 3, 1, 4])


# ---- hybrid head ---------------------------------------------------------------
def _hybrid(**kw):
    from collie_amd.hybrid import HybridPretrainedModel
    from collie_amd.model import MatrixFactorizationModel
    torch.manual_seed(0)
    mf = MatrixFactorizationModel(train=Shape(), embedding_dim=8)
    args = dict(train=Shape(), trained_model=mf, item_metadata=np.ones((20, 6)),
                item_metadata_layers_dims=[12], combined_layers_dims=[16, 4])
    args.update(kw)
    return mf, HybridPretrainedModel(**args)


def test_hybrid_construction_follows_reference(tmp_path):
    """hybrid_pretrained_matrix_factorization.py:129-136,227-325,476-484."""
    from collie_amd.hybrid import HeadAdam, HybridPretrainedModel
    with pytest.raises(ValueError, match='trained_model'):
        HybridPretrainedModel(train=Shape(), item_metadata=np.ones((20, 6)))
    mf, _ = _hybrid()
    with pytest.raises(ValueError, match='item metadata and/or user_metadata'):
        HybridPretrainedModel(train=Shape(), trained_model=mf)
    mf, model = _hybrid(user_metadata=np.ones((30, 3)))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes == {
        'embeddings.0.weight': (30, 8), 'embeddings.1.weight': (20, 8),
        'biases.0.weight': (30, 1), 'biases.1.weight': (20, 1),
        'item_metadata_layer_0.weight': (12, 6), 'item_metadata_layer_0.bias': (12,),
        # user metadata (3 raw columns, no layers) + 8 + 8 + item metadata output 12
        'combined_layer_0.weight': (16, 31), 'combined_layer_0.bias': (16,),
        'combined_layer_1.weight': (4, 16), 'combined_layer_1.bias': (4,),
        'combined_layer_2.weight': (1, 4), 'combined_layer_2.bias': (1,)}
    # the tables are copies (deepcopy, :243-250), frozen by default; biases stay trainable
    assert torch.equal(model.embeddings[0].weight, mf.user_embeddings.weight)
    assert model.embeddings[0].weight.data_ptr() != mf.user_embeddings.weight.data_ptr()
    assert not model.embeddings[0].weight.requires_grad and model.biases[0].weight.requires_grad
    model.unfreeze_embeddings()
    assert model.embeddings[1].weight.requires_grad
    assert model.hparams.user_num_embeddings == 30 and model.hparams.item_embeddings_dim == 8
    assert model.hparams.item_metadata_num_cols == 6 and model._loss_kind == 'adaptive_hinge'
    # one optimizer over all parameters (bias_optimizer is None for this model), HIP-kernel step
    ret = model.configure_optimizers()
    opts, scheds = ret
    assert len(opts) == 1 and isinstance(opts[0], HeadAdam) and isinstance(opts[0], torch.optim.Adam)
    assert len(opts[0].param_groups[0]['params']) == len(list(model.parameters()))
    # save / load directory layout (:486-534, :168-186)
    path = tmp_path / 'hybrid'
    model.save_model(path)
    assert sorted(os.listdir(path)) == ['item_metadata.pkl', 'model.pth', 'user_metadata.pkl']
    with pytest.raises(ValueError, match='overwrite'):
        model.save_model(path)
    loaded = HybridPretrainedModel(load_model_path=str(path))
    assert not loaded.training
    for k, v in model.state_dict().items():
        assert torch.equal(v, loaded.state_dict()[k])
    assert torch.equal(loaded.item_metadata, model.item_metadata)
    # no CPU fallback
    from collie_amd._lib import CollieHipError
    users = torch.zeros(4, dtype=torch.long)
    with pytest.raises(CollieHipError, match='no CPU fallback'):
        model(users, users)


def test_dense_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    from collie_amd import _lib
    lib = _lib.load()
    op = _lib.DenseOperand()
    op.n_seg = 1
    op.seg[0].src, op.seg[0].ld, op.seg[0].rows, op.seg[0].width = 256, 8, 4, 8
    args = (None, 0, None, 0, None, 0, None, 0)          # no bias-table add
    rc = lib.collie_dense_forward(ctypes.byref(op), 5, 256, 8, 1, None, 8, 4, 1, 0.01, *args,
                                  256, 4, None, None)
    assert rc != 0 and b'4 rows for 5 operand rows' in lib.collie_last_error()
    rc = lib.collie_dense_forward(ctypes.byref(op), 4, 256, 9, 1, None, 9, 4, 1, 0.01, *args,
                                  256, 4, None, None)
    assert rc != 0 and b'operand has 8 columns, layer expects 9' in lib.collie_last_error()
    op.n_seg = 5
    rc = lib.collie_dense_forward(ctypes.byref(op), 4, 256, 8, 1, None, 8, 4, 1, 0.01, *args,
                                  256, 4, None, None)
    assert rc != 0 and b'n_seg' in lib.collie_last_error()
    assert lib.collie_dense_weight_grad_workspace_bytes(100000, 128, 384) >= 391 * 128 * 385 * 4
    assert lib.collie_dense_weight_grad_workspace_bytes(0, 128, 384) == 0
    cfg = _lib.OptimCfg(_lib.OPT_SPARSE_ADAM, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.0)
    rc = lib.collie_dense_optim_step(256, 256, 10, ctypes.byref(cfg), 256, 256, None, None)
    assert rc != 0 and b'SparseAdam does not support dense gradients' in lib.collie_last_error()


def test_cold_start_construction_and_stages(tmp_path, capsys):
    """cold_start_matrix_factorization.py:159-241, multi_stage_pipeline.py:96-257."""
    from collie_amd.cold_start import ColdStartModel
    from collie_amd.optim import FusedAdam
    buckets = np.arange(20) % 4
    with pytest.raises(ValueError, match='Length of ``item_buckets``'):
        ColdStartModel(train=Shape(), item_buckets=buckets[:5])
    with pytest.raises(ValueError, match='must start at 0'):
        ColdStartModel(train=Shape(), item_buckets=buckets + 1)
    torch.manual_seed(0)
    m = ColdStartModel(train=Shape(), item_buckets=buckets, embedding_dim=8)
    assert list(m.state_dict()) == [
        'item_bucket_biases.weight', 'item_bucket_embeddings.weight', 'user_biases.weight',
        'item_biases.weight', 'user_embeddings.weight', 'item_embeddings.weight']
    assert m.hparams.num_item_buckets == 4 and m.hparams.stage_list == ['item_buckets', 'no_buckets']
    assert m.hparams.stage == 'item_buckets' and not m._mf_eval_fast
    opts, scheds = m.configure_optimizers()
    assert len(opts) == 2 and all(isinstance(o, FusedAdam) for o in opts)
    owned = [{id(p) for p in o.param_groups[0]['params']} for o in opts]
    assert id(m.item_bucket_embeddings.weight) in owned[0] and id(m.item_embeddings.weight) not in owned[0]
    assert id(m.item_embeddings.weight) in owned[1] and id(m.item_bucket_biases.weight) not in owned[1]
    assert id(m.user_embeddings.weight) in owned[0] and id(m.user_embeddings.weight) in owned[1]
    # only the current stage's optimizer steps (multi_stage_pipeline.py:253-256)
    calls = []

    class Probe:
        def __init__(self, tag):
            self.tag = tag

        def step(self, closure=None):
            calls.append(self.tag)
    m.optimizer_step(optimizer=Probe(0), optimizer_idx=0)
    m.optimizer_step(optimizer=Probe(1), optimizer_idx=1)
    assert calls == [0]
    with pytest.raises(ValueError, match='not a valid stage'):
        m.set_stage('nope')
    with torch.no_grad():
        m.item_bucket_embeddings.weight.normal_()
    m.advance_stage()
    assert m.hparams.stage == 'no_buckets' and m._mf_eval_fast
    assert torch.equal(m.item_embeddings.weight, m.item_bucket_embeddings.weight[torch.tensor(buckets)])
    m.optimizer_step(optimizer=Probe(0), optimizer_idx=0)
    m.optimizer_step(optimizer=Probe(1), optimizer_idx=1)
    assert calls == [0, 1]
    m.save_model(tmp_path / 'cs.pth')
    loaded = ColdStartModel(load_model_path=str(tmp_path / 'cs.pth'))
    assert loaded.hparams.stage == 'no_buckets' and not loaded.training
    assert torch.equal(loaded.item_embeddings.weight, m.item_embeddings.weight)
    assert 'Set ``self.hparams.stage``' in capsys.readouterr().out


def test_ctypes_prototypes_match_the_header():
    """Every prototype in include/collie_hip.h has as many parameters as its ctypes argtypes list
    in collie_amd/_lib.py, and pointer / integer / float kinds line up (a drifted binding would
    otherwise only show up as garbage arguments on the GPU)."""
    from collie_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'collie_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = re.findall(r'\b(collie_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', text, flags=re.S)
    assert len(protos) == len(_lib.SIGNATURES)
    for name, params in protos:
        params = ' '.join(params.split())
        plist = [] if params in ('', 'void') else [p.strip() for p in params.split(',')]
        argtypes = _lib.SIGNATURES[name][1]
        assert len(plist) == len(argtypes), f'{name}: header has {len(plist)} parameters, ctypes {len(argtypes)}'
        for decl, ct in zip(plist, argtypes):
            is_ptr = '*' in decl or 'collie_stream_t' in decl
            ct_is_ptr = ct in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(ct, 'contents') or \
                getattr(ct, '_type_', None) is not None and not isinstance(ct._type_, str)
            assert is_ptr == ct_is_ptr, f'{name}: "{decl}" vs {ct}'
            if not is_ptr:
                is_float = decl.startswith('float') or decl.startswith('double')
                assert is_float == (ct in (ctypes.c_float, ctypes.c_double)), f'{name}: "{decl}" vs {ct}'
                if not is_float:
                    wide = any(t in decl for t in ('int64_t', 'size_t', 'uint64_t'))
                    assert wide == (ctypes.sizeof(ct) == 8), f'{name}: "{decl}" vs {ct}'


def test_vectorised_batch_accumulation_equals_the_reference_loop():
    """evaluate_in_batches accumulates ``metric(batch) * len(batch)`` batch by batch
    (metrics.py:379-384); ``_weighted_batch_sum`` must reproduce that loop bit for bit."""
    from collie_amd.metrics import _batch_mean, _weighted_batch_sum
    rng = np.random.default_rng(0)
    for name in ('ap', 'rr', 'auc'):
        for n in (1, 19, 20, 21, 400, 65536, 70001):
            for bs in (1, 2, 7, 20, 33, 128, 200, 1000):
                if bs > n:
                    continue
                v = rng.random(n)
                if name == 'rr':
                    v = v.astype(np.float32)
                acc = 0
                for i in range(int(np.ceil(n / bs))):
                    vals = v[i * bs:(i + 1) * bs]
                    acc += _batch_mean(name, vals) * len(vals)
                assert _weighted_batch_sum(name, v, bs) == acc, (name, n, bs)
