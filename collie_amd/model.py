"""
``MatrixFactorizationModel`` — host-side mirror of the reference's
``collie/model/matrix_factorization.py`` + the parts of ``collie/model/base/base_pipeline.py`` the
matrix-factorization train/eval path uses.  Same constructor, hyper-parameter names, parameter
names/shapes (``user_biases.weight [U,1]`` ... ``item_embeddings.weight [N,D]``), ``forward``,
``calculate_loss``, ``configure_optimizers``, ``save_model`` / ``load_model_path``.

What is different is underneath: there is no ``pytorch_lightning`` (the class is a plain
``torch.nn.Module`` exposing the few LightningModule members the trainers and metrics touch) and all
arithmetic runs in ``libcollie_hip.so``:

  * ``calculate_loss`` on an implicit batch with one of the built-in losses launches ONE fused
    kernel (gathers, 1+K dot products, loss, backward coefficients, staged gradient rows) and
    returns a 0-d tensor whose ``backward()`` only arms the staged step; the fused optimizers
    (``collie_amd/optim.py``) then run the sorted-segment reduce + row update kernels.
  * ``forward(users, items)`` is an ``autograd.Function`` over the ``collie_mf_score`` kernels
    (generic route: custom callable losses, ``get_item_predictions`` ...).

There is no CPU fallback: tensors on a non-HIP device raise ``CollieHipError``.
"""
import textwrap
import warnings
from collections.abc import Iterable
from functools import partial
from typing import Callable, Dict, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.optim.lr_scheduler import ReduceLROnPlateau

from collie_amd import _lib, loss as loss_mod, ops
from collie_amd.interactions import (ApproximateNegativeSamplingInteractionsDataLoader,
                                     Interactions, InteractionsDataLoader)
from collie_amd.layers import ScaledEmbedding, ZeroEmbedding
from collie_amd.optim import FUSED_OPTIMIZERS


class HParams(dict):
    """Attribute dict: ``hasattr`` is False for missing keys, ``.get`` works (``base_pipeline.py:230,604``)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value


# ---- generic route: model(users, items) ---------------------------------------------------------
class _MFScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, users, items, ue, ie, ub, ib, y_range, sparse, status, dropout=None):
        tables = _lib.make_tables(ue, ie, ub, ib)
        out = ops.mf_score(tables, users, items, y_range, status, dropout)
        ctx.save_for_backward(users, items, ue, ie, ub, ib)
        ctx.y_range, ctx.sparse, ctx.status, ctx.dropout = y_range, sparse, status, dropout
        return out

    @staticmethod
    def backward(ctx, grad_out):
        users, items, ue, ie, ub, ib = ctx.saved_tensors
        tables = _lib.make_tables(ue, ie, ub, ib)
        gu, gi, gb = ops.mf_score_backward(tables, users, items, grad_out, ctx.y_range, ctx.status,
                                           ctx.dropout)
        u_idx = users.long().reshape(1, -1)
        i_idx = items.long().reshape(1, -1)

        def build(idx, vals, like):
            if ctx.sparse:      # the uncoalesced COO gradient nn.Embedding(sparse=True) produces
                return torch.sparse_coo_tensor(idx, vals, like.shape)
            return torch.zeros_like(like).index_add_(0, idx[0], vals)

        return (None, None, build(u_idx, gu, ue), build(i_idx, gi, ie), build(u_idx, gb, ub),
                build(i_idx, gb, ib), None, None, None, None)


# ---- fused route: calculate_loss ---------------------------------------------------------------
class _FusedStep:
    """One staged backward: batch, workspace, step id and whether the segments exist yet."""

    def __init__(self, model, tables, batch, buf, step_id, explicit=False):
        self.model, self.tables, self.batch, self.buf = model, tables, batch, buf
        self.step_id = step_id
        self.explicit = explicit      # explicit losses: the user-bias gradient is not zero
        self.armed = False
        self.grad_scale = None
        self.segments = False
        self.done = set()

    def update(self, is_bias, dense, per_side):
        """per_side: {side: (cfg, state1, state2, row_stamp)}; one launch for both sides."""
        per_side = {s: v for s, v in per_side.items() if (s, is_bias) not in self.done}
        if not per_side:
            return                      # a table is updated at most once per staged backward
        if not self.segments:
            # (small sparse batches are coalesced inside the update launch itself)
            cache = self.model.__dict__.setdefault('_needs_segments', {})
            key = (self.batch.batch, bool(dense), id(self.tables))
            if key not in cache:
                cache[key] = ops.train_update_needs_segments(self.tables, self.batch, dense)
            if cache[key]:
                ops.train_build_segments(self.tables, self.batch, self.buf)
            self.segments = True
        ops.train_update(self.tables, self.batch, self.buf,
                         (2 if self.explicit else 1) if is_bias else 0, dense, self.step_id,
                         per_side, self.grad_scale)
        self.done.update((s, is_bias) for s in per_side)


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, staged, loss_value, *weights):
        ctx.staged = staged
        return loss_value.reshape(()).clone()

    @staticmethod
    def backward(ctx, grad_out):
        staged = ctx.staged
        staged.grad_scale = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        staged.armed = True
        return (None, None, None) + (None,) * 4


class _FusedLossTensor(torch.Tensor):
    """The 0-d loss ``calculate_loss`` returns on the fused route.  It is an ordinary autograd
    tensor (``grad_fn`` = ``_FusedLoss``: ``loss / 2`` then ``.backward()``, ``torch.autograd.grad``
    ... go through the engine as usual); only the plain ``loss.backward()`` both trainers issue is
    answered directly: for this node the engine would do nothing but hand ``grad = 1`` to
    ``_FusedLoss.backward``, at ~40 us of engine and device-thread hand-off per call -- a third of
    a launch-bound step at the reference's default batch size."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        staged = self.__dict__.get('_collie_staged')
        if (staged is not None and gradient is None and not create_graph and inputs is None
                and self._backward_hooks is None and not staged.armed
                and staged.model.__dict__.get('_staged_step') is staged):
            staged.grad_scale = None        # d loss / d loss = 1: the kernels' default
            staged.armed = True
            return None
        return super().backward(gradient, retain_graph, create_graph, inputs)


def _fused_loss(model, staged, loss_value, weights):
    out = _FusedLoss.apply(model, staged, loss_value, *weights).as_subclass(_FusedLossTensor)
    out.__dict__['_collie_staged'] = staged
    return out


INTERACTIONS_LIKE_INPUT = Union[ApproximateNegativeSamplingInteractionsDataLoader, Interactions,
                                InteractionsDataLoader]
_TABLES = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')
_SLOTS = {'user_embeddings': (_lib.SIDE_USER, False), 'item_embeddings': (_lib.SIDE_ITEM, False),
          'user_biases': (_lib.SIDE_USER, True), 'item_biases': (_lib.SIDE_ITEM, True)}


class MatrixFactorizationModel(nn.Module):
    """
    Training pipeline for the matrix factorization model (see the module docstring).

    Parameters are those of the reference (``matrix_factorization.py:73-94``):
    ``train, val, embedding_dim=30, dropout_p=0.0, sparse=False, lr=1e-3, bias_lr=1e-2,
    lr_scheduler_func=partial(ReduceLROnPlateau, patience=1), weight_decay=0.0, optimizer='adam',
    bias_optimizer='sgd', loss='hinge', metadata_for_loss=None, metadata_for_loss_weights=None,
    y_range=None, load_model_path=None, map_location=None``.
    """
    _mf_eval_fast = True        # evaluate_in_batches may use the fused MFMA scoring kernels

    def __init__(self,
                 train: INTERACTIONS_LIKE_INPUT = None,
                 val: INTERACTIONS_LIKE_INPUT = None,
                 embedding_dim: int = 30,
                 dropout_p: float = 0.0,
                 sparse: bool = False,
                 lr: float = 1e-3,
                 bias_lr: Optional[Union[float, str]] = 1e-2,
                 lr_scheduler_func: Optional[Callable] = partial(ReduceLROnPlateau, patience=1),
                 weight_decay: float = 0.0,
                 optimizer: Union[str, Callable] = 'adam',
                 bias_optimizer: Optional[Union[str, Callable]] = 'sgd',
                 loss: Union[str, Callable[..., torch.Tensor]] = 'hinge',
                 metadata_for_loss: Optional[Dict[str, torch.Tensor]] = None,
                 metadata_for_loss_weights: Optional[Dict[str, float]] = None,
                 y_range: Optional[Tuple[float, float]] = None,
                 load_model_path: Optional[str] = None,
                 map_location: Optional[str] = None):
        super().__init__()
        if isinstance(train, Interactions):
            train = InteractionsDataLoader(interactions=train, shuffle=True)
        if isinstance(val, Interactions):
            val = InteractionsDataLoader(interactions=val, shuffle=False)
        self.train_loader = train
        self.val_loader = val
        self.lr_scheduler_func = lr_scheduler_func
        self.loss = loss
        self.optimizer = optimizer
        self.bias_optimizer = bias_optimizer
        self.hparams = HParams()
        self._staged_step = None
        self._workspace = None
        self._status = None
        self._meta_codes = None

        if load_model_path is not None:
            self._load_model_init_helper(load_model_path, map_location)
            return

        if self.train_loader is None:
            raise TypeError('``train`` must be provided to all newly-instantiated models!')
        if self.val_loader is not None:
            assert self.train_loader.num_users == self.val_loader.num_users, (
                'Both training and val ``num_users`` must equal: '
                f'{self.train_loader.num_users} != {self.val_loader.num_users}.')
            assert self.train_loader.num_items == self.val_loader.num_items, (
                'Both training and val ``num_items`` must equal: '
                f'{self.train_loader.num_items} != {self.val_loader.num_items}.')
            err = ('Training and val ``num_negative_samples`` property must both equal ``1`` or '
                   'both be greater than ``1``.')
            if self.train_loader.num_negative_samples == 1:
                assert self.val_loader.num_negative_samples == 1, err
            elif self.train_loader.num_negative_samples > 1:
                assert self.val_loader.num_negative_samples > 1, err
            else:
                raise ValueError('``self.train_loader.num_negative_samples`` must be greater than '
                                 f'``0``, not {self.train_loader.num_negative_samples}.')

        self.hparams.update(
            embedding_dim=embedding_dim, dropout_p=dropout_p, sparse=sparse, lr=lr, bias_lr=bias_lr,
            weight_decay=weight_decay, optimizer=optimizer if isinstance(optimizer, str) else None,
            bias_optimizer=bias_optimizer if isinstance(bias_optimizer, str) else None,
            loss=loss if isinstance(loss, str) else None, metadata_for_loss=metadata_for_loss,
            metadata_for_loss_weights=metadata_for_loss_weights, y_range=y_range,
            load_model_path=None, map_location=map_location)
        self.hparams.num_users = self.train_loader.num_users
        self.hparams.num_items = self.train_loader.num_items
        self.hparams.num_epochs_completed = 0

        self._configure_loss()

        if self.hparams.sparse and self.hparams.weight_decay != 0:
            warnings.warn(f'``weight_decay`` value must be 0 when ``sparse`` is flagged, not '
                          f'{self.hparams.weight_decay}. Setting to 0.')
            self.hparams.weight_decay = 0.0

        self._setup_model()

    # ---------------------------------------------------------------- construction / persistence
    def _setup_model(self, **kwargs) -> None:
        """``matrix_factorization.py:99-118``: same modules, names, shapes and init."""
        hp = self.hparams
        self.user_biases = ZeroEmbedding(hp.num_users, 1, sparse=hp.sparse)
        self.item_biases = ZeroEmbedding(hp.num_items, 1, sparse=hp.sparse)
        self.user_embeddings = ScaledEmbedding(hp.num_users, hp.embedding_dim, sparse=hp.sparse)
        self.item_embeddings = ScaledEmbedding(hp.num_items, hp.embedding_dim, sparse=hp.sparse)
        self.dropout = nn.Dropout(p=hp.dropout_p)

    def _load_model_init_helper(self, load_model_path, map_location) -> None:
        """``base_pipeline.py:245-257`` (loads files written by ``save_model``)."""
        loaded = torch.load(str(load_model_path), map_location=map_location, weights_only=False)
        for key, value in loaded['hparams'].items():
            self.hparams[key] = value
        self.hparams['load_model_path'] = load_model_path
        self.hparams['map_location'] = map_location
        self._setup_model()
        self.load_state_dict(loaded['state_dict'])
        self.eval()

    def save_model(self, filename='model.pth') -> None:
        """``base_pipeline.py:880-900``: ``{'state_dict', 'hparams'}``."""
        torch.save({'state_dict': self.state_dict(), 'hparams': dict(self.hparams)}, str(filename))

    # ----------------------------------------------------------------- LightningModule stand-ins
    @property
    def device(self):
        return self.user_embeddings.weight.device

    def log(self, *args, **kwargs):
        pass

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def train_dataloader(self):
        return self.train_loader

    def val_dataloader(self):
        return self.val_loader

    def on_fit_start(self):
        self._move_any_external_data_to_device()

    def optimizer_step(self, epoch=None, batch_idx=None, optimizer=None, optimizer_idx=None,
                       optimizer_closure=None, **kwargs):
        optimizer.step(closure=optimizer_closure)

    def training_step(self, batch, batch_idx=None, optimizer_idx=None):
        return self.calculate_loss(batch)

    def validation_step(self, batch, batch_idx=None, optimizer_idx=None):
        return self.calculate_loss(batch)

    def _move_any_external_data_to_device(self):
        """``base_pipeline.py:269-275`` (+ the int32 category codes the kernels read)."""
        md = self.hparams.get('metadata_for_loss')
        if md is not None:
            self.hparams.metadata_for_loss = {k: torch.as_tensor(v).to(self.device)
                                              for k, v in md.items()}
        self._meta_codes = None
        self._loss_cfg = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._retire_workspace()
        self._status = None
        self._meta_codes = None
        self._loss_cfg = None
        self._tables_cache = None
        self._staged_step = None
        return out

    # ------------------------------------------------------------------------- loss / optimizers
    def _configure_loss(self) -> None:
        """``base_pipeline.py:277-340``: string -> loss, adaptive whenever K > 1."""
        self.loss_function = None
        self._loss_kind = None
        if callable(self.loss):
            self.loss_function = self.loss
            return
        self.hparams._is_implicit = False
        if self.loss == 'mse':
            self.loss_function = torch.nn.MSELoss(reduction='mean')
            return
        elif self.loss == 'mae':
            self.loss_function = torch.nn.L1Loss(reduction='mean')
            return
        self.hparams._is_implicit = True
        if not hasattr(self.train_loader, 'num_negative_samples'):
            raise ValueError('``num_negative_samples`` attribute not found in ``train_loader`` - '
                             'are you using explicit data with an implicit loss function?')
        many = self.train_loader.num_negative_samples > 1
        if self.loss == 'warp':
            if not many:
                raise ValueError('Cannot use WARP loss with a single negative sample!')
            self._loss_kind = 'warp'
        elif 'bpr' in self.loss:
            if not many and 'adaptive' in self.loss:
                warnings.warn('Adaptive BPR loss specified, but ``num_negative_samples`` == 1. '
                              'Using standard BPR loss instead.')
            self._loss_kind = 'adaptive_bpr' if many else 'bpr'
        elif 'hinge' in self.loss or self.loss == 'adaptive':
            if not many and 'adaptive' in self.loss:
                warnings.warn('Adaptive hinge loss specified, but ``num_negative_samples`` == 1. '
                              'Using standard hinge loss instead.')
            self._loss_kind = 'adaptive_hinge' if many else 'hinge'
        else:
            raise ValueError('{} is not a valid loss function.'.format(self.loss))
        self.loss_function = loss_mod.LOSS_FUNCTIONS[self._loss_kind]

    # Which parameters an optimizer role owns: the substring 'bias' in the parameter NAME decides
    # (``base_pipeline.py:450-467``), and which learning-rate hyper-parameter it reads.
    _ROLES = {'bias': (lambda name: 'bias' in name, 'bias_lr'),
              'all_but_bias': (lambda name: 'bias' not in name, 'lr'),
              'all': (lambda name: True, 'lr')}

    def configure_optimizers(self):
        """Contract of ``base_pipeline.py:342-404``: one optimizer over every parameter, or -- when
        a ``bias_optimizer`` is configured -- a main optimizer over the non-bias parameters followed
        by a bias optimizer (``'infer'`` copies the main optimizer / learning rate).  Returns the
        optimizer, the list of both, or ``(optimizers, scheduler dicts)`` when an
        ``lr_scheduler_func`` is set; string optimizers become the fused HIP ones."""
        if self.bias_optimizer == 'infer':
            self.bias_optimizer = self.optimizer
        if self.bias_optimizer is not None and self.hparams.bias_lr == 'infer':
            self.hparams.bias_lr = self.hparams.lr
        plan = ([(self.optimizer, 'all')] if self.bias_optimizer is None else
                [(self.optimizer, 'all_but_bias'), (self.bias_optimizer, 'bias')])
        optimizers = [self._get_optimizer(spec, optimizer_type=role) for spec, role in plan]
        if self.lr_scheduler_func is None:
            return optimizers if len(optimizers) > 1 else optimizers[0]
        monitor = 'train_loss_epoch' if self.val_loader is None else 'val_loss_epoch'
        return (optimizers, [{'scheduler': self.lr_scheduler_func(opt), 'monitor': monitor}
                             for opt in optimizers])

    def _get_optimizer(self, optimizer, **kwargs):
        """One optimizer from a name (``'sgd' | 'adagrad' | 'adam' | 'sparse_adam'``, built as the
        fused subclass of that torch optimizer and attached to this model) or from any callable
        ``torch.optim`` class (``base_pipeline.py:406-437``; a class that rejects ``weight_decay``
        is retried without it; ``sparse_adam`` never receives it)."""
        if callable(optimizer):
            for with_decay in (True, False):
                try:
                    return optimizer(self._get_optimizer_parameters(
                        include_weight_decay=with_decay, **kwargs))
                except TypeError:
                    if not with_decay:
                        raise
        fused = FUSED_OPTIMIZERS.get(optimizer) if isinstance(optimizer, str) else None
        if fused is None:
            raise ValueError('{} is not a valid optimizer!'.format(optimizer))
        groups = self._get_optimizer_parameters(
            include_weight_decay=optimizer != 'sparse_adam', **kwargs)
        return fused(groups).attach(self)

    def _get_optimizer_parameters(self, include_weight_decay=True, optimizer_type='all', **kwargs):
        """The single parameter group of one optimizer role (``base_pipeline.py:439-479``)."""
        assert optimizer_type in ['bias', 'all_but_bias', 'all'], f'{optimizer_type} not valid!'
        owns, lr_key = self._ROLES[optimizer_type]
        group = {'params': [p for name, p in self.named_parameters() if owns(name)],
                 'lr': self.hparams[lr_key]}
        if include_weight_decay:
            group['weight_decay'] = self.hparams.weight_decay
        return [group]

    # ------------------------------------------------------------------------------- arithmetic
    def _weights(self):
        """The four weight Parameters in ``_TABLES`` order, read straight from the module dicts
        (``nn.Module.__getattr__`` costs more than the launches of a small batch allow)."""
        m = self._modules
        return tuple(m[n]._parameters['weight'] for n in _TABLES)

    def _tables(self):
        """The four tables as the C struct; rebuilt only when a parameter's storage moved."""
        ws = self._weights()
        key = (ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ws[3].data_ptr())
        cached = self.__dict__.get('_tables_cache')
        if cached is None or cached[0] != key:
            cached = (key, _lib.make_tables(*(w.data for w in ws)))
            self.__dict__['_tables_cache'] = cached
        return cached[1]

    def _table_slot(self, param):
        ws = self._weights()
        for i, name in enumerate(_TABLES):
            if ws[i] is param:
                return _SLOTS[name]
        return None

    def _status_flag(self):
        if self._status is None or self._status.device != self.device:
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._status

    def check_status(self):
        """Raise ``IndexError`` if any kernel saw an out-of-range id since the last check (syncs)."""
        if self._status is not None and int(self._status.item()) & 1:
            self._status.zero_()
            raise IndexError('index out of range in self: a user or item id is outside the tables')

    def check_anomaly(self):
        """``detect_anomaly`` on the fused route (``trainer.py:333``): raise ``RuntimeError`` if a
        fused loss kernel produced a non-finite loss since the last check (syncs; the trainers call
        it once per step only when ``detect_anomaly`` is set).  The autograd route is covered by
        ``torch.autograd.set_detect_anomaly`` itself."""
        if self._status is not None and int(self._status.item()) & 2:
            self._status.bitwise_and_(1)
            self._drop_staged_step()        # the staged gradients are poisoned: never apply them
            raise RuntimeError("Function 'CollieFusedLossBackward' returned nan values in its 0th "
                               'output.')

    def _retire_workspace(self):
        """Drop the training workspace but keep its step counter: the optimizers' ``_row_stamp``
        arrays outlive the workspace and are compared with the ids it hands out."""
        ws = self.__dict__.get('_workspace')
        if ws is not None:
            self.__dict__['_step_seq'] = max(self.__dict__.get('_step_seq', 0), ws.step_id)
        self._workspace = None

    def _train_workspace(self, dev):
        if self._workspace is None:
            self._workspace = ops.TrainWorkspace(dev, self.__dict__.get('_step_seq', 0))
        return self._workspace

    def _drop_staged_step(self):
        self.__dict__['_staged_step'] = None    # a plain attribute: skip nn.Module.__setattr__

    def forward(self, users: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
        """``matrix_factorization.py:120-159``."""
        _lib.require_hip(self.user_embeddings.weight, 'MatrixFactorizationModel')
        y = self.hparams.get('y_range')
        dropout = None
        if self.training and self.hparams.dropout_p > 0:
            # nn.Dropout on both gathered rows (:143-147), masks from the library's counter-based
            # generator: seeded once per model from torch's generator (so ``torch.manual_seed``
            # makes runs repeatable), a fresh call number per forward (fresh masks per call)
            if self.__dict__.get('_dropout_seed') is None:
                self.__dict__['_dropout_seed'] = int(torch.randint(0, 2 ** 62, (1,)).item())
                self.__dict__['_dropout_calls'] = 0
            self.__dict__['_dropout_calls'] += 1
            dropout = (float(self.hparams.dropout_p), self.__dict__['_dropout_seed'],
                       self.__dict__['_dropout_calls'])
        return _MFScore.apply(users, items, self.user_embeddings.weight,
                              self.item_embeddings.weight, self.user_biases.weight,
                              self.item_biases.weight, tuple(y) if y is not None else None,
                              bool(self.hparams.sparse), self._status_flag(), dropout)

    def _can_fuse(self):
        if self._loss_kind is None or self.hparams.get('y_range') is not None:
            return False
        if self.training and self.hparams.dropout_p > 0:
            return False
        md = self.hparams.get('metadata_for_loss')
        return md is None or len(md) <= _lib.MAX_META

    def calculate_loss(self, batch) -> torch.Tensor:
        """``base_pipeline.py:582-654``: same batch grammar, same errors."""
        if len(batch) == 2 and isinstance(batch[0], Iterable) and len(batch[0]) == 2:
            if self.hparams.get('_is_implicit') is False:
                raise ValueError('Explicit loss with implicit data is invalid!')
            ((users, pos_items), neg_items) = batch
            if neg_items.dim() == 1:
                neg_items = neg_items.reshape(-1, 1)
            if self._can_fuse() and neg_items.shape[1] <= _lib.MAX_NEG:
                return self._fused_calculate_loss(users, pos_items, neg_items)
            # generic route: same op sequence as the reference, every op a HIP kernel
            users = users.long()
            pos_items = pos_items.long()
            neg_items = torch.transpose(neg_items, 0, 1).long()
            pos_preds = self(users, pos_items)
            neg_preds = self(users.repeat(neg_items.shape[0]), neg_items.flatten()).view(
                neg_items.shape[0], len(users))
            return self.loss_function(
                pos_preds, neg_preds, num_items=self.hparams.num_items, positive_items=pos_items,
                negative_items=neg_items, metadata=self.hparams.get('metadata_for_loss'),
                metadata_weights=self.hparams.get('metadata_for_loss_weights'))
        elif len(batch) == 3:
            if self.hparams.get('_is_implicit') is True:
                raise ValueError('Implicit loss with explicit data is invalid!')
            (users, items, ratings) = batch
            if (isinstance(self.loss, str) and self.loss in ops.EXPLICIT_LOSSES
                    and not (self.training and self.hparams.dropout_p > 0)):
                return self._fused_explicit_loss(users, items, ratings)
            preds = self(users.long(), items.long())
            return self.loss_function(preds, ratings.float())
        else:
            raise ValueError(f'Unexpected format for batch: {batch}. See docs for expected format.')

    def _fused_calculate_loss(self, users, pos_items, neg_items):
        ws = self._weights()
        dev = ws[0].device
        _lib.require_hip(ws[0], 'MatrixFactorizationModel')
        for t in (users, pos_items, neg_items):
            _lib.require_hip(t, 'batch')
        need_grad = torch.is_grad_enabled() and any(w.requires_grad for w in ws)
        self._train_workspace(dev)
        self.__dict__['_staged_step'] = None        # the workspace is about to be overwritten
        B, K = neg_items.shape
        buf = self._workspace.ensure(B, K, self.hparams.embedding_dim, self.hparams.num_users,
                                     self.hparams.num_items)
        if self._meta_codes is None:
            self._meta_codes = loss_mod.metadata_to_codes(
                self.hparams.get('metadata_for_loss'),
                self.hparams.get('metadata_for_loss_weights'), dev)
        codes, weights = self._meta_codes
        tables = self._tables()
        hbatch = ops.make_batch(users, pos_items, neg_items)
        cfg = self.__dict__.get('_loss_cfg')
        if cfg is None or cfg._key != (self._loss_kind, self.hparams.num_items, id(codes)):
            cfg = ops.make_loss_cfg(self._loss_kind, self.hparams.num_items, codes, weights)
            cfg._key = (self._loss_kind, self.hparams.num_items, id(codes))
            self.__dict__['_loss_cfg'] = cfg
        loss_value = torch.empty(1, dtype=torch.float32, device=dev)
        step_id = self._workspace.next_step() if need_grad else max(self._workspace.step_id, 1)
        ops.train_forward(tables, hbatch, cfg, buf, need_grad, step_id, loss_value, None, None,
                          self._status_flag())
        if not need_grad:
            return loss_value.reshape(())
        staged = _FusedStep(self, tables, hbatch, buf, step_id)
        self.__dict__['_staged_step'] = staged
        return _fused_loss(self, staged, loss_value, ws)

    def _fused_explicit_loss(self, users, items, ratings):
        """``base_pipeline.py:635-650`` with ``loss='mse' | 'mae'``: one fused HIP kernel."""
        dev = self.device
        _lib.require_hip(self.user_embeddings.weight, 'MatrixFactorizationModel')
        for t in (users, items, ratings):
            _lib.require_hip(t, 'batch')
        users = users.contiguous()
        items = items.contiguous()
        ratings = ratings.float().contiguous()
        need_grad = torch.is_grad_enabled() and any(
            getattr(self, n).weight.requires_grad for n in _TABLES)
        self._train_workspace(dev)
        self._staged_step = None
        B = users.shape[0]
        buf = self._workspace.ensure(B, 1, self.hparams.embedding_dim, self.hparams.num_users,
                                     self.hparams.num_items)
        tables = self._tables()
        y = self.hparams.get('y_range')
        loss_value = torch.empty(1, dtype=torch.float32, device=dev)
        step_id = self._workspace.next_step() if need_grad else max(self._workspace.step_id, 1)
        ops.explicit_forward(tables, users, items, ratings, self.loss,
                             tuple(y) if y is not None else None, buf, need_grad, loss_value, None,
                             self._status_flag())
        if not need_grad:
            return loss_value.reshape(())
        col = items.reshape(-1, 1)
        hbatch = ops.make_batch(users, items, col)
        hbatch._keepalive = (users, items, col, ratings)
        staged = _FusedStep(self, tables, hbatch, buf, step_id, explicit=True)
        self._staged_step = staged
        return _fused_loss(self, staged, loss_value, self._weights())

    # ------------------------------------------------------------------------------ conveniences
    def _get_item_embeddings(self) -> torch.Tensor:
        return self.item_embeddings.weight.data

    def _get_user_embeddings(self) -> torch.Tensor:
        return self.user_embeddings.weight.data

    def get_item_predictions(self, user_id: int = 0, unseen_items_only: bool = False,
                             sort_values: bool = True):
        """``base_pipeline.py:656-718``."""
        import pandas as pd
        if user_id >= self.hparams.num_users:
            raise ValueError(f'``user_id`` {user_id} is not in the model. Expected ID between '
                             f'``0`` and ``self.hparams.num_users - 1`` '
                             f'(``{self.hparams.num_users - 1}``), not ``{user_id}``')
        user = torch.full((self.hparams.num_items,), user_id, dtype=torch.long, device=self.device)
        item = torch.arange(self.hparams.num_items, dtype=torch.long, device=self.device)
        with torch.no_grad():
            preds = pd.Series(self(user, item).detach().cpu().numpy())
        if sort_values:
            preds = preds.sort_values(ascending=False)
        if unseen_items_only:
            drop = self.train_loader.mat.tocsr()[user_id, :].nonzero()[1]
            if self.val_loader is not None:
                drop = np.concatenate([drop, self.val_loader.mat.tocsr()[user_id, :].nonzero()[1]])
            return preds.drop(drop)
        return preds

    def get_user_predictions(self, item_id: int = 0, unseen_users_only: bool = False,
                             sort_values: bool = True):
        """``base_pipeline.py:720-783``."""
        import pandas as pd
        if item_id >= self.hparams.num_items:
            raise ValueError(f'``item_id`` {item_id} is not in the model. Expected ID between '
                             f'``0`` and ``self.hparams.num_items - 1`` '
                             f'(``{self.hparams.num_items - 1}``), not ``{item_id}``')
        item = torch.full((self.hparams.num_users,), item_id, dtype=torch.long, device=self.device)
        user = torch.arange(self.hparams.num_users, dtype=torch.long, device=self.device)
        with torch.no_grad():
            preds = pd.Series(self(user, item).detach().cpu().numpy())
        if sort_values:
            preds = preds.sort_values(ascending=False)
        if unseen_users_only:
            drop = self.train_loader.mat.tocsc()[:, item_id].nonzero()[0]
            if self.val_loader is not None:
                drop = np.concatenate([drop, self.val_loader.mat.tocsc()[:, item_id].nonzero()[0]])
            return preds.drop(drop)
        return preds

    def item_item_similarity(self, item_id: int):
        """``base_pipeline.py:785-817``: cosine similarity of every item with ``item_id``."""
        if item_id >= self.hparams.num_items:
            raise ValueError(f'``item_id`` {item_id} is not in the model. Expected ID between '
                             f'``0`` and ``self.hparams.num_items - 1`` '
                             f'(``{self.hparams.num_items - 1}``), not ``{item_id}``')
        return self._calculate_embedding_similarity(self._get_item_embeddings(), item_id)

    def user_user_similarity(self, user_id: int):
        """``base_pipeline.py:819-852``."""
        if user_id >= self.hparams.num_users:
            raise ValueError(f'``user_id`` {user_id} is not in the model. Expected ID between '
                             f'``0`` and ``self.hparams.num_users - 1`` '
                             f'(``{self.hparams.num_users - 1}``), not ``{user_id}``')
        return self._calculate_embedding_similarity(self._get_user_embeddings(), user_id)

    def _calculate_embedding_similarity(self, embeddings: torch.Tensor, id: int):
        """``base_pipeline.py:854-864`` as one HIP kernel (``collie_embedding_similarity``)."""
        import pandas as pd
        sims = ops.embedding_similarity(embeddings, id)
        return pd.Series(sims.cpu().numpy()).sort_values(ascending=False)
