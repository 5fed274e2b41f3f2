"""
``HybridPretrainedModel`` — host-side mirror of the reference's
``collie/model/hybrid_pretrained_matrix_factorization.py`` (BASELINE
configs[4]): the four tables of a trained ``MatrixFactorizationModel`` are copied (embeddings
optionally frozen), item and/or user metadata rows go through ``Linear + LeakyReLU`` layers, the
concatenation ``[user_meta_out, U[u], I[i], item_meta_out]`` goes through ``Linear + LeakyReLU``
layers and a final ``Linear(.., 1)``, and both bias-table rows are added.

Same constructor arguments, module / parameter names (``embeddings.0.weight`` ...
``combined_layer_3.bias``), ``save_model`` directory layout and errors as the reference.  All
arithmetic runs in ``libcollie_hip.so``: every layer, forward and backward, is an exact-fp32 MFMA
GEMM (``collie_dense_forward`` / ``collie_dense_weight_grad``) whose operand gathers the embedding /
metadata rows on the fly (the ``torch.cat`` input is never materialised), and the optimizers are
the ``collie_dense_optim_step`` / ``collie_dense_rows_update`` kernels.  There is no CPU fallback.
"""
import contextlib
import copy
import os
import warnings
from collections import OrderedDict
from functools import partial
from pathlib import Path
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import torch
from torch import nn
from torch.optim.lr_scheduler import ReduceLROnPlateau

from collie_amd import _lib, ops
from collie_amd.layers import ScaledEmbedding, ZeroEmbedding
from collie_amd.model import INTERACTIONS_LIKE_INPUT, MatrixFactorizationModel
from collie_amd.optim import FusedAdagrad, FusedAdam, FusedSGD


class _HeadScore(torch.autograd.Function):
    """``HybridPretrainedModel.forward`` over M (user, item) pairs; ``params`` only tell autograd
    which leaves the result depends on (their gradients come out of ``model._head_backward``)."""

    @staticmethod
    def forward(ctx, model, users, items, *params):
        scores, saved = model._head_forward(users, items)
        ctx.model, ctx.users, ctx.items, ctx.saved = model, users, items, saved
        ctx.n_params = len(params)
        # set by calculate_loss for this one call: (B, K) of a batch whose loss gives at most one
        # negative per sample a gradient
        ctx.active_hint, model._active_rows_hint = model._active_rows_hint, None
        model._user_period_hint = None
        return scores

    @staticmethod
    def backward(ctx, grad_out):
        grads = ctx.model._head_backward(ctx.users, ctx.items, ctx.saved,
                                         grad_out.contiguous().float(), ctx.active_hint)
        return (None, None, None) + tuple(grads)


class _SplitScores(torch.autograd.Function):
    """``preds[:B], preds[B:].view(K, B)`` whose backward is ONE concatenation (autograd's slice
    backward is two zero fills, two copies and an add per step)."""

    @staticmethod
    def forward(ctx, preds, B, K):
        ctx.shape = (B, K)
        return preds[:B], preds[B:].view(K, B)

    @staticmethod
    def backward(ctx, g_pos, g_neg):
        B, K = ctx.shape
        if g_pos is None:
            g_pos = g_neg.new_zeros(B)
        if g_neg is None:
            g_neg = g_pos.new_zeros(K, B)
        return torch.cat([g_pos.reshape(-1), g_neg.reshape(-1)]), None, None


class _HeadMixin:
    """``step()`` for the optimizers of the hybrid model: dense parameters with ``.grad`` go
    through ``collie_dense_optim_step``; tables whose gradient the backward staged as an
    uncoalesced COO list go through ``collie_dense_rows_update``."""

    def zero_grad(self, set_to_none: bool = True):
        model = getattr(self, '_collie_model', None)
        if model is not None:
            model._staged_rows.clear()
        return torch.optim.Optimizer.zero_grad(self, set_to_none=set_to_none)

    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        model = self._collie_model
        for group in self.param_groups:
            dense = {}      # step count -> the group's dense parameters at that count: one launch
            for p in group['params']:
                staged = model._staged_rows.pop(id(p), None)
                if staged is None and p.grad is None:
                    continue
                cfg, s1, s2, stamp = self._table_args(model, p, group, True)
                if staged is not None:
                    rows, vals = staged
                    model._rows_ws = ops.dense_rows_update(
                        p.data, rows, vals, cfg, s1, s2, stamp, model._status_flag(),
                        model._rows_ws)
                else:
                    grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    if grad.dtype != torch.float32 or not p.data.is_contiguous():
                        ops.dense_optim_step(p.data, grad, cfg, s1, s2)
                        continue
                    bucket = dense.setdefault(int(cfg.step), (cfg, [], [], [], []))
                    for lst, t in zip(bucket[1:], (p.data, grad, s1, s2)):
                        lst.append(t)
            for cfg, ps, gs, s1s, s2s in dense.values():
                if len(ps) == 1:
                    ops.dense_optim_step(ps[0], gs[0], cfg, s1s[0], s2s[0])
                else:
                    ops.dense_optim_step_multi(ps, gs, cfg,
                                               s1s if s1s[0] is not None else None,
                                               s2s if s2s[0] is not None else None)
        return loss


class HeadAdam(_HeadMixin, FusedAdam):
    pass


class HeadSGD(_HeadMixin, FusedSGD):
    pass


class HeadAdagrad(_HeadMixin, FusedAdagrad):
    pass


HEAD_OPTIMIZERS = {'adam': HeadAdam, 'sgd': HeadSGD, 'adagrad': HeadAdagrad}


class HybridPretrainedModel(MatrixFactorizationModel):
    """
    Parameters are those of the reference (``hybrid_pretrained_matrix_factorization.py:102-126``):
    ``train, val, item_metadata, user_metadata, trained_model, item_metadata_layers_dims,
    user_metadata_layers_dims, combined_layers_dims=[128, 64, 32], freeze_embeddings=True,
    dropout_p=0.0, lr=1e-3, lr_scheduler_func, weight_decay=0.0, optimizer='adam', loss='hinge',
    metadata_for_loss, metadata_for_loss_weights, load_model_path, map_location``.
    """
    _mf_eval_fast = False       # evaluation goes through get_preds (the score is not a dot product)
    _active_rows_hint = None    # see calculate_loss / _head_backward
    _user_period_hint = None    # see calculate_loss / _head_forward
    _USER_SHARE_ONCE = True     # tests switch it off to compare with the plain first layer
    _FUSED_GRID_TAIL = True     # score_catalogue: default head's tail in one launch (tests compare)
    # losses of this library under which only the positive and ONE negative of a sample receive a
    # score gradient (adaptive: the arg-max negative, bpr.py:153-167 / hinge.py:157-171; WARP: the
    # first violation, warp.py:73-154)
    _ONE_NEGATIVE_LOSSES = ('adaptive_bpr', 'adaptive_hinge', 'warp')
    _NO_GRAD_CHUNK = 1 << 21    # pairs per head pass when no gradient is needed

    def __init__(self,
                 train: INTERACTIONS_LIKE_INPUT = None,
                 val: INTERACTIONS_LIKE_INPUT = None,
                 item_metadata=None,
                 user_metadata=None,
                 trained_model: MatrixFactorizationModel = None,
                 item_metadata_layers_dims: Optional[List[int]] = None,
                 user_metadata_layers_dims: Optional[List[int]] = None,
                 combined_layers_dims: List[int] = [128, 64, 32],
                 freeze_embeddings: bool = True,
                 dropout_p: float = 0.0,
                 lr: float = 1e-3,
                 lr_scheduler_func: Optional[Callable] = partial(ReduceLROnPlateau, patience=1),
                 weight_decay: float = 0.0,
                 optimizer: Union[str, Callable] = 'adam',
                 loss: Union[str, Callable[..., torch.Tensor]] = 'hinge',
                 metadata_for_loss: Optional[Dict[str, torch.Tensor]] = None,
                 metadata_for_loss_weights: Optional[Dict[str, float]] = None,
                 load_model_path: Optional[str] = None,
                 map_location: Optional[str] = None):
        item_cols = user_cols = None
        if load_model_path is None:
            if trained_model is None:
                raise ValueError('Must provide ``trained_model`` for ``HybridPretrainedModel``.')
            if item_metadata is None and user_metadata is None:
                raise ValueError('Must provide item metadata and/or user_metadata for '
                                 '``HybridPretrainedModel``.')
            item_metadata = self._as_float_tensor(item_metadata)
            user_metadata = self._as_float_tensor(user_metadata)
            item_cols = item_metadata.shape[1] if item_metadata is not None else None
            user_cols = user_metadata.shape[1] if user_metadata is not None else None
        self._pending = dict(
            item_metadata=item_metadata, user_metadata=user_metadata, trained_model=trained_model,
            hp=dict(item_metadata_layers_dims=item_metadata_layers_dims,
                    user_metadata_layers_dims=user_metadata_layers_dims,
                    combined_layers_dims=list(combined_layers_dims),
                    freeze_embeddings=freeze_embeddings, item_metadata_num_cols=item_cols,
                    user_metadata_num_cols=user_cols))
        self.item_metadata = None
        self.user_metadata = None
        self._staged_rows = {}
        self._rows_ws = None
        self._wgrad_ws = None
        self._materialize_table_grads = False
        super().__init__(train=train, val=val, embedding_dim=0, dropout_p=dropout_p, sparse=False,
                         lr=lr, bias_lr=None, lr_scheduler_func=lr_scheduler_func,
                         weight_decay=weight_decay, optimizer=optimizer, bias_optimizer=None,
                         loss=loss, metadata_for_loss=metadata_for_loss,
                         metadata_for_loss_weights=metadata_for_loss_weights, y_range=None,
                         load_model_path=load_model_path, map_location=map_location)

    @staticmethod
    def _as_float_tensor(metadata):
        """``hybrid_pretrained_matrix_factorization.py:138-152``."""
        if metadata is None:
            return None
        if hasattr(metadata, 'to_numpy'):
            metadata = torch.from_numpy(metadata.to_numpy())
        elif isinstance(metadata, np.ndarray):
            metadata = torch.from_numpy(metadata)
        return metadata.float()

    # ---------------------------------------------------------------- construction / persistence
    def _setup_model(self, **kwargs) -> None:
        """``hybrid_pretrained_matrix_factorization.py:227-325``: same modules, names and init."""
        hp = self.hparams
        pending = getattr(self, '_pending', None)
        if pending is not None and hp.get('load_model_path') is None:
            hp.update(pending['hp'])
            trained = pending['trained_model']
            self.__dict__['_trained_model'] = trained     # not registered as a sub-module
            self.item_metadata = pending['item_metadata']
            self.user_metadata = pending['user_metadata']
            self.embeddings = nn.Sequential(copy.deepcopy(trained.user_embeddings),
                                            copy.deepcopy(trained.item_embeddings))
            self.biases = nn.Sequential(copy.deepcopy(trained.user_biases),
                                        copy.deepcopy(trained.item_biases))
            if hp.freeze_embeddings:
                self.freeze_embeddings()
            else:
                self.unfreeze_embeddings()
            hp.user_num_embeddings = self.embeddings[0].num_embeddings
            hp.user_embeddings_dim = self.embeddings[0].embedding_dim
            hp.item_num_embeddings = self.embeddings[1].num_embeddings
            hp.item_embeddings_dim = self.embeddings[1].embedding_dim
            hp.embedding_dim = hp.user_embeddings_dim
        else:
            self.embeddings = nn.Sequential(
                ScaledEmbedding(hp.user_num_embeddings, hp.user_embeddings_dim),
                ScaledEmbedding(hp.item_num_embeddings, hp.item_embeddings_dim))
            self.biases = nn.Sequential(ZeroEmbedding(hp.user_num_embeddings, 1),
                                        ZeroEmbedding(hp.item_num_embeddings, 1))
        self._pending = None
        self.dropout = nn.Dropout(p=hp.dropout_p)

        item_out = hp.item_metadata_num_cols
        self.item_metadata_layers = None
        if hp.item_metadata_layers_dims is not None:
            self._configure_metadata_layers('item', hp.item_metadata_layers_dims,
                                            hp.item_metadata_num_cols)
            item_out = hp.item_metadata_layers_dims[-1]
        user_out = hp.user_metadata_num_cols
        self.user_metadata_layers = None
        if hp.user_metadata_layers_dims is not None:
            self._configure_metadata_layers('user', hp.user_metadata_layers_dims,
                                            hp.user_metadata_num_cols)
            user_out = hp.user_metadata_layers_dims[-1]

        combined_in = hp.user_embeddings_dim + hp.item_embeddings_dim
        combined_in += (item_out or 0) + (user_out or 0)
        dims = [combined_in] + list(hp.combined_layers_dims) + [1]
        self.combined_layers = [nn.Linear(dims[i - 1], dims[i]) for i in range(1, len(dims))]
        for i, layer in enumerate(self.combined_layers):
            nn.init.xavier_normal_(layer.weight)
            self.add_module('combined_layer_{}'.format(i), layer)

    def _configure_metadata_layers(self, metadata_type, layers_dims, num_cols) -> None:
        """``hybrid_pretrained_matrix_factorization.py:188-225``."""
        full = [num_cols] + list(layers_dims)
        layers = [nn.Linear(full[i - 1], full[i]) for i in range(1, len(full))]
        setattr(self, f'{metadata_type}_metadata_layers', layers)
        for i, layer in enumerate(layers):
            nn.init.xavier_normal_(layer.weight)
            self.add_module(f'{metadata_type}_metadata_layer_{i}', layer)

    def __setattr__(self, name, value):
        # the layer LISTS are plain attributes in the reference too (the modules are registered
        # one by one through add_module)
        if name in ('item_metadata_layers', 'user_metadata_layers', 'combined_layers',
                    'item_metadata', 'user_metadata'):
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def _load_model_init_helper(self, load_model_path, map_location) -> None:
        """``hybrid_pretrained_matrix_factorization.py:168-186``: a directory written by ``save_model``."""
        import joblib
        super()._load_model_init_helper(os.path.join(str(load_model_path), 'model.pth'),
                                        map_location)
        self.hparams['load_model_path'] = load_model_path
        for kind in ('item', 'user'):
            try:
                setattr(self, f'{kind}_metadata',
                        joblib.load(os.path.join(str(load_model_path), f'{kind}_metadata.pkl')))
            except FileNotFoundError:
                if self.hparams.get(f'{kind}_metadata_layers_dims') is not None:
                    warnings.warn(f'``{kind}_metadata.pkl`` not found')

    def save_model(self, path='model', overwrite: bool = False) -> None:
        """``hybrid_pretrained_matrix_factorization.py:486-534``."""
        import joblib
        path = str(path)
        if os.path.exists(path):
            if os.listdir(path) and overwrite is False:
                raise ValueError(f'Data exists in ``path`` at {path} and ``overwrite`` is False.')
        Path(path).mkdir(parents=True, exist_ok=True)
        if self.item_metadata is not None:
            joblib.dump(self.item_metadata.cpu(), os.path.join(path, 'item_metadata.pkl'))
        if self.user_metadata is not None:
            joblib.dump(self.user_metadata.cpu(), os.path.join(path, 'user_metadata.pkl'))
        state = OrderedDict((k, v) for k, v in self.state_dict().items()
                            if '_trained_model' not in k)
        hparams = {k: v for k, v in self.hparams.items()}
        torch.save({'state_dict': state, 'hparams': hparams}, os.path.join(path, 'model.pth'))

    def load_from_hybrid_model(self, hybrid_model) -> None:
        """``hybrid_pretrained_matrix_factorization.py:536-555``."""
        for key, value in hybrid_model.hparams.items():
            self.hparams[key] = value
        self.hparams['load_model_path'] = 'copied'      # rebuild layers from hparams, not a trained model
        self._setup_model()
        self.hparams['load_model_path'] = None
        self.load_state_dict(state_dict=hybrid_model.state_dict())
        self.item_metadata = hybrid_model.item_metadata
        self.user_metadata = hybrid_model.user_metadata
        self.eval()

    def freeze_embeddings(self) -> None:
        self.embeddings[0].weight.requires_grad = False
        self.embeddings[1].weight.requires_grad = False

    def unfreeze_embeddings(self) -> None:
        self.embeddings[0].weight.requires_grad = True
        self.embeddings[1].weight.requires_grad = True

    # ----------------------------------------------------------------- LightningModule stand-ins
    @property
    def device(self):
        return self.embeddings[0].weight.device

    def _move_any_external_data_to_device(self):
        """``hybrid_pretrained_matrix_factorization.py:160-166``."""
        super()._move_any_external_data_to_device()
        if self.item_metadata is not None:
            self.item_metadata = self.item_metadata.to(self.device).contiguous()
        if self.user_metadata is not None:
            self.user_metadata = self.user_metadata.to(self.device).contiguous()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._staged_rows = {}
        self._rows_ws = None
        self._wgrad_ws = None
        return out

    def _get_optimizer(self, optimizer, **kwargs):
        """``base_pipeline.py:406-437``; string optimizers become the HIP-kernel ones."""
        if callable(optimizer):
            self._materialize_table_grads = True
            return super()._get_optimizer(optimizer, **kwargs)
        self._materialize_table_grads = False
        if optimizer == 'sparse_adam':
            raise RuntimeError('SparseAdam does not support dense gradients, please consider Adam '
                               'instead')
        if optimizer not in HEAD_OPTIMIZERS:
            raise ValueError('{} is not a valid optimizer!'.format(optimizer))
        params = self._get_optimizer_parameters(include_weight_decay=True, **kwargs)
        return HEAD_OPTIMIZERS[optimizer](params).attach(self)

    def _drop_staged_step(self):
        self._staged_rows.clear()

    def _can_fuse(self):
        return False        # calculate_loss: one head pass over all (1 + K) B pairs, see below

    def _get_item_embeddings(self) -> torch.Tensor:
        return self.embeddings[1].weight.data

    def _get_user_embeddings(self) -> torch.Tensor:
        return self.embeddings[0].weight.data

    # ------------------------------------------------------------------------------- arithmetic
    def forward(self, users: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
        """``hybrid_pretrained_matrix_factorization.py:401-464``."""
        if self.training and self.hparams.dropout_p > 0:
            raise NotImplementedError(
                'dropout_p > 0 in training mode is not implemented in the HIP path '
                '(SURVEY.md §7: torch-RNG parity is infeasible); use dropout_p=0.')
        _lib.require_hip(self.embeddings[0].weight, 'HybridPretrainedModel')
        _lib.require_hip(users, 'users')
        _lib.require_hip(items, 'items')
        if self.item_metadata is not None and not self.item_metadata.is_cuda:
            self._move_any_external_data_to_device()
        if self.user_metadata is not None and not self.user_metadata.is_cuda:
            self._move_any_external_data_to_device()
        users, items = users.contiguous(), items.contiguous()
        params = [p for p in self.parameters() if p.requires_grad]
        if not (torch.is_grad_enabled() and params) and users.shape[0] > self._NO_GRAD_CHUNK:
            # scoring only (get_preds expands users x items pairs): bounded activation memory
            step = self._NO_GRAD_CHUNK
            with torch.no_grad():
                return torch.cat([self._head_forward(users[i:i + step], items[i:i + step])[0]
                                  for i in range(0, users.shape[0], step)]).squeeze()
        return _HeadScore.apply(self, users, items, *params).squeeze()

    def calculate_loss(self, batch) -> torch.Tensor:
        """``base_pipeline.py:582-654``.  The positive and the K B negative pairs go through the
        head in ONE pass (rows [0, B) = positives, then the negatives k-major, exactly the
        reference's ``users.repeat(K)`` / ``neg_items.flatten()`` order), so every layer is one
        GEMM and every weight gradient one deterministic reduction."""
        if len(batch) == 2 and hasattr(batch[0], '__len__') and len(batch[0]) == 2:
            if self.hparams.get('_is_implicit') is False:
                raise ValueError('Explicit loss with implicit data is invalid!')
            ((users, pos_items), neg_items) = batch
            if neg_items.dim() == 1:
                neg_items = neg_items.reshape(-1, 1)
            B, K = neg_items.shape
            if (users.is_cuda and 1 <= K <= _lib.MAX_NEG and B >= 1
                    and B * (1 + K) < (1 << 30)):
                # the (1 + K) B pairs in one launch (int64 ids, positives first)
                all_users, all_items = ops.pair_ids(users, pos_items, neg_items)
                pos_items = all_items[:B]
                neg_items = all_items[B:].view(K, B)
            else:
                users = users.long()
                pos_items = pos_items.long()
                neg_items = torch.transpose(neg_items, 0, 1).long()
                all_users = users.repeat(K + 1)
                all_items = torch.cat([pos_items, neg_items.flatten()])
            from collie_amd import loss as loss_mod
            kind = getattr(self, '_loss_kind', None)
            if (K > 1 and kind in self._ONE_NEGATIVE_LOSSES
                    and self.loss_function is loss_mod.LOSS_FUNCTIONS.get(kind)):
                self._active_rows_hint = (B, K)
            # rows b, B + b, 2 B + b, ... share their user: the user embedding's share of the first
            # combined layer is computed once per sample (_head_forward)
            self._user_period_hint = B if (self._USER_SHARE_ONCE and B >= 256) else None
            try:
                preds = self(all_users, all_items)
            finally:        # the hints describe THIS call only, whatever happens inside it
                self._active_rows_hint = self._user_period_hint = None
            pos_preds, neg_preds = _SplitScores.apply(preds, B, K)
            return self.loss_function(
                pos_preds, neg_preds, num_items=self.hparams.num_items, positive_items=pos_items,
                negative_items=neg_items, metadata=self.hparams.get('metadata_for_loss'),
                metadata_weights=self.hparams.get('metadata_for_loss_weights'))
        elif len(batch) == 3:
            if self.hparams.get('_is_implicit') is True:
                raise ValueError('Implicit loss with explicit data is invalid!')
            (users, items, ratings) = batch
            return self.loss_function(self(users.long(), items.long()), ratings.float())
        raise ValueError(f'Unexpected format for batch: {batch}. See docs for expected format.')

    def _metadata_stack(self, kind, ids, m, saved):
        """``_compute_metadata_output`` (:327-361): the operand segment this side contributes."""
        metadata = getattr(self, f'{kind}_metadata')
        if metadata is None:
            return None
        layers = getattr(self, f'{kind}_metadata_layers')
        seg = (metadata, ids, None)
        if layers is None:
            return seg
        outs = []
        for layer in layers:
            x_op = ops.dense_operand(seg)
            out = ops.dense_forward(x_op, m, layer.weight.data, layer.bias.data, True,
                                    status=self._status_flag())
            outs.append((seg, out))
            seg = (out, None, None)
        saved[kind] = outs
        return seg

    def _head_forward(self, users, items):
        m = users.shape[0]
        saved = {}
        user_seg = self._metadata_stack('user', users, m, saved)
        item_seg = self._metadata_stack('item', items, m, saved)
        tagged = [('user_meta', user_seg),
                  ('user_emb', (self.embeddings[0].weight.data, users, None)),
                  ('item_emb', (self.embeddings[1].weight.data, items, None)),
                  ('item_meta', item_seg)]
        tagged = [(tag, seg) for tag, seg in tagged if seg is not None]
        segs = [seg for _, seg in tagged]
        saved['segments'] = tagged
        x_segs = segs
        outs = []
        for li, layer in enumerate(self.combined_layers[:-1]):
            period = self._user_period_hint
            if (li == 0 and period is not None and m % period == 0 and m > period
                    and tagged[0][0] == 'user_emb' and len(tagged) > 1):
                # users = u.repeat(1 + K): X W^T = U[u] W_u^T + (rest of the row) W_rest^T, and the
                # first term is the same for the 1 + K rows of a sample -- one small GEMM over B
                # rows, added in the epilogue of the layer's GEMM over the remaining columns
                d = tagged[0][1][0].shape[1]
                part = ops.dense_forward(
                    ops.dense_operand((self.embeddings[0].weight.data, users[:period], None)),
                    period, layer.weight.data, None, False, status=self._status_flag(),
                    col_range=(0, d))
                out = ops.dense_forward(ops.dense_operand(*x_segs[1:]), m, layer.weight.data,
                                        layer.bias.data, True, status=self._status_flag(),
                                        col_range=(d, layer.weight.shape[1]),
                                        addend=(part, period))
            else:
                out = ops.dense_forward(ops.dense_operand(*x_segs), m, layer.weight.data,
                                        layer.bias.data, True, status=self._status_flag())
            outs.append((x_segs, out))
            x_segs = [(out, None, None)]
        last = self.combined_layers[-1]
        scores = ops.dense_forward(
            ops.dense_operand(*x_segs), m, last.weight.data, last.bias.data, False,
            row_bias=(users, self.biases[0].weight.data, items, self.biases[1].weight.data),
            status=self._status_flag())
        outs.append((x_segs, None))
        saved['combined'] = outs
        return scores.reshape(-1), saved

    @contextlib.contextmanager
    def catalogue_scope(self):
        """A stretch of ``score_catalogue`` calls during which no parameter changes (one
        ``evaluate_in_batches`` call, ``metrics.py:367-386``): the catalogue's item share is computed
        by the first call and kept until the scope closes."""
        self.__dict__['_catalogue_share'] = {}
        try:
            yield self
        finally:
            self.__dict__['_catalogue_share'] = None

    def score_catalogue(self, user_ids, n_items: int) -> torch.Tensor:
        """Scores of every ``(user, item < n_items)`` pair, ``[len(user_ids), n_items]``: what
        ``get_preds`` (``metrics.py:77-107``) obtains from ``model(users.repeat_interleave(n),
        arange(n).repeat(U))``.  On that grid the item side of the first combined layer (item
        embedding + item metadata stack, ``hybrid_pretrained_matrix_factorization.py:401-464``) does
        not depend on the user and the user embedding's share does not depend on the item: each is
        computed once (per ITEM, per USER), a pair's first layer is the sum of its two rows plus
        bias and activation (``collie_dense_outer_sum``), and only the tail layers run per pair.  Returns None when the model's layout does
        not allow the split (user metadata, a head of one layer): the caller then takes
        ``model(users, items)``."""
        if (self.user_metadata is not None or len(self.combined_layers) < 2 or n_items < 1
                or (self.training and self.hparams.dropout_p > 0)):
            return None
        _lib.require_hip(self.embeddings[0].weight, 'HybridPretrainedModel')
        if self.item_metadata is not None and not self.item_metadata.is_cuda:
            self._move_any_external_data_to_device()
        dev = self.embeddings[0].weight.device
        if isinstance(user_ids, torch.Tensor):
            users = user_ids.to(device=dev, dtype=torch.int64).contiguous()
        else:
            users = torch.as_tensor(np.asarray(user_ids), dtype=torch.int64, device=dev).contiguous()
        n_users = users.shape[0]
        if n_users == 0:
            return torch.empty((0, n_items), dtype=torch.float32, device=dev)
        first, last = self.combined_layers[0], self.combined_layers[-1]
        d = self.embeddings[0].weight.shape[1]
        step = self._NO_GRAD_CHUNK
        status = self._status_flag()
        with torch.no_grad():
            catalogue = torch.arange(n_items, device=dev)
            kept = self.__dict__.get('_catalogue_share')
            share = kept.get(n_items) if kept is not None else None
            have_share = share is not None
            if not have_share:
                share = torch.empty((n_items, first.weight.shape[0]), dtype=torch.float32,
                                    device=dev)
            for i0 in range(0, 0 if have_share else n_items, step):
                ids = catalogue[i0:i0 + step]
                n = ids.shape[0]
                segs = [(self.embeddings[1].weight.data, ids, None)]
                meta = self._metadata_stack('item', ids, n, {})
                if meta is not None:
                    segs.append(meta)
                ops.dense_forward(ops.dense_operand(*segs), n, first.weight.data, None, False,
                                  status=status, col_range=(d, first.weight.shape[1]),
                                  out=share[i0:i0 + n])
            if kept is not None and not have_share:
                kept[n_items] = share       # only a completely filled share is ever reused
            user_share = ops.dense_forward(
                ops.dense_operand((self.embeddings[0].weight.data, users, None)), n_users,
                first.weight.data, None, False, status=status, col_range=(0, d))
            scores = torch.empty((n_users, n_items), dtype=torch.float32, device=dev)
            shapes = [tuple(layer.weight.shape) for layer in self.combined_layers]
            if (self._FUSED_GRID_TAIL and shapes[0][0] == 128
                    and shapes[1:] == [(64, 128), (32, 64), (1, 32)]):
                # the reference's default head: the whole tail in one launch, nothing but the
                # scores written
                second, third = self.combined_layers[1], self.combined_layers[2]
                ops.dense_grid_tail(user_share, share, first.bias.data, second.weight.data,
                                    second.bias.data, third.weight.data, third.bias.data,
                                    last.weight.data.reshape(-1), last.bias.data, users,
                                    self.biases[0].weight.data.reshape(-1), 0,
                                    self.biases[1].weight.data.reshape(-1), scores, status=status)
                return scores
            width = min(n_items, step)
            for i0 in range(0, n_items, width):
                ni = min(width, n_items - i0)
                per = max(1, step // ni)
                for u0 in range(0, n_users, per):
                    nu = min(per, n_users - u0)
                    m = nu * ni
                    pair_users = users[u0:u0 + nu].repeat_interleave(ni)
                    pair_items = catalogue[i0:i0 + ni].repeat(nu)
                    x = ops.dense_outer_sum(user_share[u0:u0 + nu], share[i0:i0 + ni],
                                            first.bias.data, True)
                    for layer in self.combined_layers[1:-1]:
                        x = ops.dense_forward(ops.dense_operand((x, None, None)), m,
                                              layer.weight.data, layer.bias.data, True,
                                              status=status)
                    out = ops.dense_forward(
                        ops.dense_operand((x, None, None)), m, last.weight.data, last.bias.data,
                        False, status=status,
                        row_bias=(pair_users, self.biases[0].weight.data,
                                  pair_items, self.biases[1].weight.data))
                    if ni == n_items:
                        scores[u0:u0 + nu] = out.view(nu, ni)
                    else:
                        scores[u0:u0 + nu, i0:i0 + ni] = out.view(nu, ni)
        return scores

    def _layer_backward(self, layer, x_segs, out, dz, m, grads):
        """Gradients of one ``Linear (+ LeakyReLU)``: parameters into ``grads``; returns the operand
        ``dz_op`` (dY with the activation backward folded in) for the caller's data backward."""
        dz_op = ops.dense_operand((dz, None, out))
        n_out, n_in = layer.weight.shape
        dw, db, self._wgrad_ws = ops.dense_weight_grad(
            dz_op, ops.dense_operand(*x_segs), m, n_out, n_in, True, self._wgrad_ws,
            self._status_flag())
        grads[id(layer.weight)] = dw
        grads[id(layer.bias)] = db
        return dz_op

    def _compact_rows(self, users, items, saved, g, hint):
        """
        The backward on the rows that can carry a gradient.  Under the losses named in
        ``_ONE_NEGATIVE_LOSSES`` the score gradient of a batch is zero outside the B positives and
        one negative per sample -- 2 B of the (1 + K) B rows the reference's autograd pushes through
        every backward GEMM (82 % zeros at K = 10).  A zero row adds nothing to any weight gradient
        and needs no data gradient, so the saved activations, the id arrays and ``g`` are gathered
        down to those 2 B rows (fixed size: no host read) and every backward kernel runs on them.
        """
        B, K = hint
        # which rows (all-zero negatives: k = 0, a zero row), and g / users / items at them: one
        # launch.  Positive and negative of a sample sit side by side: sums that cancel analytically
        # (user biases, the last Linear bias: +g and -g per sample) then cancel pair by pair
        act, g_c, users_c, items_c = ops.active_rows(g.contiguous(), B, K, users, items)
        # the saved activations, gathered down to those rows four tensors per launch
        floats = []
        for kind in ('user', 'item'):
            for in_seg, o in saved.get(kind, ()):
                floats += [in_seg[0] if in_seg[1] is None else None, o]
        for x_segs, o in saved['combined']:
            floats += [sg[0] if sg[1] is None else None for sg in x_segs] + [o]
        distinct = {}
        for t in floats:
            if t is not None and t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 \
                    and t.shape[0] == g.shape[0]:
                distinct.setdefault(id(t), t)
        cache = {id(users): users_c, id(items): items_c, id(g): g_c}
        if distinct:
            for t, c in zip(distinct.values(),
                            ops.gather_rows(list(distinct.values()), act, self._status_flag())):
                cache[id(t)] = c

        def take(t):
            if id(t) not in cache:
                cache[id(t)] = t.index_select(0, act)
            return cache[id(t)]

        def seg_c(seg):
            src, ids, ref = seg
            return (src, take(ids), ref) if ids is not None else (take(src), None, ref)

        out = {}
        for kind in ('user', 'item'):
            if kind in saved:
                out[kind] = [(seg_c(in_seg), take(o)) for in_seg, o in saved[kind]]
        out['segments'] = [(tag, seg_c(seg)) for tag, seg in saved['segments']]
        out['combined'] = [([seg_c(sg) for sg in x_segs], take(o) if o is not None else None)
                           for x_segs, o in saved['combined']]
        return take(users), take(items), out, take(g)

    def _head_backward(self, users, items, saved, g, active_hint=None):
        compact = active_hint is not None and g.shape[0] == (active_hint[1] + 1) * active_hint[0]
        if compact:
            users, items, saved, g = self._compact_rows(users, items, saved, g, active_hint)
        m = users.shape[0]
        grads = {}
        dz = g.reshape(m, 1)
        # bias tables (hybrid_pretrained...:394-398): d score / d bias row = 1
        for table, ids in ((self.biases[0].weight, users), (self.biases[1].weight, items)):
            if table.requires_grad:
                self._staged_rows[id(table)] = (ids, g)
        if compact and self.biases[0].weight.requires_grad:
            # The one-negative losses give a sample's negative exactly minus its positive's score
            # gradient, and both rows share the user: in the compact batch they are neighbours, so
            # every user-bias row's sum is (+g) + (-g) = an exact 0, pair by pair.  An empty entry
            # list says the same thing without the sort: every row takes the zero-gradient update.
            self._staged_rows[id(self.biases[0].weight)] = (users[:0], g[:0])
        combined = saved['combined']
        for li in range(len(combined) - 1, 0, -1):
            x_segs, out = combined[li]
            layer = self.combined_layers[li]
            dz_op = self._layer_backward(layer, x_segs, out, dz, m, grads)
            dz = ops.dense_forward(dz_op, m, layer.weight.data, None, False, transposed=True)
        x_segs, out = combined[0]
        layer = self.combined_layers[0]
        dz_op = self._layer_backward(layer, x_segs, out, dz, m, grads)
        # data backward of the first combined layer, one column slice of W per segment that needs it
        col = 0
        for tag, (src, ids, _) in saved['segments']:
            cols = (col, col + src.shape[1])
            col += src.shape[1]
            if tag in ('user_emb', 'item_emb'):
                table = self.embeddings[0 if tag == 'user_emb' else 1].weight
                if table.requires_grad:
                    d_rows = ops.dense_forward(dz_op, m, layer.weight.data, None, False,
                                               transposed=True, col_range=cols)
                    self._staged_rows[id(table)] = (ids, d_rows)
                continue
            kind = 'user' if tag == 'user_meta' else 'item'
            if kind not in saved:
                continue        # raw metadata rows feed the head directly: nothing to train there
            d_meta = ops.dense_forward(dz_op, m, layer.weight.data, None, False, transposed=True,
                                       col_range=cols)
            stack = saved[kind]
            layers = getattr(self, f'{kind}_metadata_layers')
            for lj in range(len(stack) - 1, -1, -1):
                in_seg, out_j = stack[lj]
                dzj = self._layer_backward(layers[lj], [in_seg], out_j, d_meta, m, grads)
                if lj > 0:
                    d_meta = ops.dense_forward(dzj, m, layers[lj].weight.data, None, False,
                                               transposed=True)
        if self._materialize_table_grads:
            # a user-supplied torch optimizer reads ``.grad``: hand the staged rows over as the
            # dense gradients nn.Embedding(sparse=False) would have produced
            for table in (self.biases[0].weight, self.biases[1].weight, self.embeddings[0].weight,
                          self.embeddings[1].weight):
                staged = self._staged_rows.pop(id(table), None)
                if staged is not None:
                    ids, vals = staged
                    grads[id(table)] = torch.zeros_like(table).index_add_(
                        0, ids.long(), vals.reshape(ids.shape[0], -1))
        return [grads.get(id(p)) for p in self.parameters() if p.requires_grad]
