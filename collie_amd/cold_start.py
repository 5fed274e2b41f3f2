"""
``ColdStartModel`` — host-side mirror of the reference's
``collie/model/cold_start_matrix_factorization.py`` + the multi-stage plumbing it inherits from
``collie/model/base/multi_stage_pipeline.py``: a matrix factorization model
trained in two stages.  In stage ``item_buckets`` every item is represented by the embedding and
bias of its *bucket* (``item_buckets[item]``); ``advance_stage()`` copies each item's bucket row
into its own row and stage ``no_buckets`` trains plain MF from there.

Both stages run on the same HIP kernels as ``MatrixFactorizationModel``: a stage only decides which
pair of item tables the fused training step / scoring kernels see, and stage ``item_buckets`` maps
the batch's item ids to bucket ids first (one integer gather).  There is no CPU fallback.
"""
from collections import OrderedDict
from functools import partial
from typing import Callable, Dict, Iterable, Optional, Union

import torch
from torch import nn
from torch.optim.lr_scheduler import ReduceLROnPlateau

from collie_amd import _lib, ops
from collie_amd.layers import ScaledEmbedding, ZeroEmbedding
from collie_amd.model import (INTERACTIONS_LIKE_INPUT, MatrixFactorizationModel, _MFScore)


class ColdStartModel(MatrixFactorizationModel):
    """
    Parameters are those of the reference (``cold_start_matrix_factorization.py:134-158``):
    ``train, val, item_buckets, embedding_dim=30, dropout_p=0.0, sparse=False,
    item_buckets_stage_lr=1e-3, no_buckets_stage_lr=1e-3, lr_scheduler_func, weight_decay=0.0,
    item_buckets_stage_optimizer='adam', no_buckets_stage_optimizer='adam', loss='hinge',
    metadata_for_loss, metadata_for_loss_weights, load_model_path, map_location``.

    As in the reference, every stage's optimizer is built from ``self.optimizer`` — the
    ``BasePipeline`` default ``'adam'`` (``multi_stage_pipeline.py:203`` passes ``self.optimizer``,
    not the ``'optimizer'`` entry of the stage's configuration).
    """

    def __init__(self,
                 train: INTERACTIONS_LIKE_INPUT = None,
                 val: INTERACTIONS_LIKE_INPUT = None,
                 item_buckets: Iterable[int] = None,
                 embedding_dim: int = 30,
                 dropout_p: float = 0.0,
                 sparse: bool = False,
                 item_buckets_stage_lr: float = 1e-3,
                 no_buckets_stage_lr: float = 1e-3,
                 lr_scheduler_func: Optional[Callable] = partial(ReduceLROnPlateau, patience=1),
                 weight_decay: float = 0.0,
                 item_buckets_stage_optimizer: Union[str, Callable] = 'adam',
                 no_buckets_stage_optimizer: Union[str, Callable] = 'adam',
                 loss: Union[str, Callable[..., torch.Tensor]] = 'hinge',
                 metadata_for_loss: Optional[Dict[str, torch.Tensor]] = None,
                 metadata_for_loss_weights: Optional[Dict[str, float]] = None,
                 load_model_path: Optional[str] = None,
                 map_location: Optional[str] = None):
        self._pending_stage = None
        if load_model_path is None:
            optimizer_config_list = [
                {'lr': item_buckets_stage_lr, 'optimizer': item_buckets_stage_optimizer,
                 'parameter_prefix_list': ['user_embed', 'user_bias', 'item_bucket_embed',
                                           'item_bucket_bias'],
                 'stage': 'item_buckets'},
                {'lr': no_buckets_stage_lr, 'optimizer': no_buckets_stage_optimizer,
                 'parameter_prefix_list': ['user_embed', 'user_bias', 'item_embed', 'item_bias'],
                 'stage': 'no_buckets'},
            ]
            if not isinstance(item_buckets, torch.Tensor):
                item_buckets = torch.tensor(item_buckets)
            assert item_buckets.dim() == 1, (
                f'``item_buckets`` must be 1-dimensional, not {item_buckets.dim()}-dimensional!')
            if train is not None and len(item_buckets) != train.num_items:
                raise ValueError('Length of ``item_buckets`` must be equal to the number of items in '
                                 f'the dataset: {len(item_buckets)} != {train.num_items}.')
            if min(item_buckets) != 0:
                raise ValueError(f'``item_buckets`` IDs must start at 0, not {min(item_buckets)}!')
            stage_list = list(OrderedDict.fromkeys(c['stage'] for c in optimizer_config_list))
            self._pending_stage = dict(
                item_buckets=item_buckets, num_item_buckets=item_buckets.max().item() + 1,
                optimizer_config_list=optimizer_config_list, stage_list=stage_list,
                item_buckets_stage_lr=item_buckets_stage_lr, no_buckets_stage_lr=no_buckets_stage_lr,
                stage=stage_list[0])
        super().__init__(train=train, val=val, embedding_dim=embedding_dim, dropout_p=dropout_p,
                         sparse=sparse, lr=item_buckets_stage_lr, bias_lr=None,
                         lr_scheduler_func=lr_scheduler_func, weight_decay=weight_decay,
                         optimizer='adam', bias_optimizer=None, loss=loss,
                         metadata_for_loss=metadata_for_loss,
                         metadata_for_loss_weights=metadata_for_loss_weights, y_range=None,
                         load_model_path=load_model_path, map_location=map_location)
        if load_model_path is None:
            self.set_stage(self.hparams.stage_list[0])

    # ---------------------------------------------------------------- construction / persistence
    def _setup_model(self, **kwargs) -> None:
        """``cold_start_matrix_factorization.py:243-282``: same modules, order, names and init."""
        hp = self.hparams
        if self._pending_stage is not None:
            hp.update(self._pending_stage)
            self._pending_stage = None
        self.item_bucket_biases = ZeroEmbedding(hp.num_item_buckets, 1, sparse=hp.sparse)
        self.item_bucket_embeddings = ScaledEmbedding(hp.num_item_buckets, hp.embedding_dim,
                                                      sparse=hp.sparse)
        self.user_biases = ZeroEmbedding(hp.num_users, 1, sparse=hp.sparse)
        self.item_biases = ZeroEmbedding(hp.num_items, 1, sparse=hp.sparse)
        self.user_embeddings = ScaledEmbedding(hp.num_users, hp.embedding_dim, sparse=hp.sparse)
        self.item_embeddings = ScaledEmbedding(hp.num_items, hp.embedding_dim, sparse=hp.sparse)
        self.dropout = nn.Dropout(p=hp.dropout_p)

    def _load_model_init_helper(self, load_model_path, map_location) -> None:
        """``multi_stage_pipeline.py:120-125``: a loaded model is in its last stage."""
        super()._load_model_init_helper(load_model_path, map_location)
        self.hparams.stage = self.hparams.stage_list[-1]
        print(f'Set ``self.hparams.stage`` to "{self.hparams.stage}"')

    def _move_any_external_data_to_device(self):
        """``cold_start_matrix_factorization.py:205-208``."""
        super()._move_any_external_data_to_device()
        self.hparams.item_buckets = self.hparams.item_buckets.to(self.device)

    # ------------------------------------------------------------------------------------ stages
    def _reset_stage_caches(self):
        # the training workspace keeps per-item stamps laid out for ONE pair of item tables
        self._retire_workspace()
        self._tables_cache = None
        self._staged_step = None

    def _copy_weights(self, old: nn.Embedding, new: nn.Embedding, buckets: torch.Tensor) -> None:
        new.weight.data.copy_(old.weight.data[buckets.to(new.weight.device)])

    def set_stage(self, stage: str) -> None:
        """``cold_start_matrix_factorization.py:213-241``."""
        current_stage = self.hparams.stage
        if stage in self.hparams.stage_list:
            if current_stage == 'item_buckets' and stage == 'no_buckets':
                print('Copying over item embeddings...')
                self._copy_weights(self.item_bucket_biases, self.item_biases,
                                   self.hparams.item_buckets)
                self._copy_weights(self.item_bucket_embeddings, self.item_embeddings,
                                   self.hparams.item_buckets)
        else:
            raise ValueError(f'"{stage}" is not a valid stage, please choose one of '
                             f'{self.hparams.stage_list}')
        self.hparams.stage = stage
        self._reset_stage_caches()
        print(f'Set ``self.hparams.stage`` to "{stage}"')

    def advance_stage(self) -> None:
        """``multi_stage_pipeline.py:127-137``."""
        stage = self.hparams.stage
        if stage in self.hparams.stage_list:
            stage_idx = self.hparams.stage_list.index(stage)
            if (stage_idx + 1) >= len(self.hparams.stage_list):
                raise ValueError(f'Cannot advance stage past {stage} - it is the final stage!')
            self.set_stage(stage=self.hparams.stage_list[stage_idx + 1])

    @property
    def _bucketed(self):
        return self.hparams.get('stage') == 'item_buckets'

    @property
    def _mf_eval_fast(self):
        # in the bucket stage item i scores with the row of its BUCKET: the fused catalog scan
        # (one row per item id) only applies once the items have rows of their own
        return not self._bucketed

    def _item_modules(self):
        if self._bucketed:
            return self.item_bucket_embeddings, self.item_bucket_biases
        return self.item_embeddings, self.item_biases

    # ------------------------------------------------------------------------------- optimizers
    def _get_optimizer_parameters(self, optimizer_config=None, include_weight_decay=True, **kwargs):
        """``multi_stage_pipeline.py:157-184``: parameters whose name starts with a listed prefix."""
        prefixes = tuple(optimizer_config['parameter_prefix_list'])
        groups = [{'params': [p for n, p in self.named_parameters() if n.startswith(prefixes)],
                   'lr': optimizer_config['lr']}]
        if include_weight_decay:
            for g in groups:
                g['weight_decay'] = self.hparams.weight_decay
        return groups

    def configure_optimizers(self):
        """``multi_stage_pipeline.py:186-224``: one optimizer (+ scheduler) per stage."""
        optimizers = [self._get_optimizer(self.optimizer, optimizer_config=cfg)
                      for cfg in self.hparams.optimizer_config_list]
        if self.lr_scheduler_func is not None:
            monitor = 'val_loss_epoch' if self.val_loader is not None else 'train_loss_epoch'
            return optimizers, [{'scheduler': self.lr_scheduler_func(o), 'monitor': monitor}
                                for o in optimizers]
        return optimizers

    def optimizer_step(self, epoch=None, batch_idx=None, optimizer=None, optimizer_idx=None,
                       optimizer_closure=None, **kwargs):
        """``multi_stage_pipeline.py:226-257``: only the current stage's optimizer steps."""
        if self.hparams.optimizer_config_list[optimizer_idx]['stage'] == self.hparams.stage:
            optimizer.step(closure=optimizer_closure)
        elif optimizer_closure is not None:
            optimizer_closure()

    # ------------------------------------------------------------------------------- arithmetic
    def _tables(self):
        ie, ib = self._item_modules()
        ws = (self.user_embeddings.weight, ie.weight, self.user_biases.weight, ib.weight)
        key = tuple(w.data_ptr() for w in ws)
        cached = getattr(self, '_tables_cache', None)
        if cached is None or cached[0] != key:
            cached = (key, _lib.make_tables(*(w.data for w in ws)))
            self._tables_cache = cached
        return cached[1]

    def _table_slot(self, param):
        ie, ib = self._item_modules()
        for module, slot in ((self.user_embeddings, (_lib.SIDE_USER, False)),
                             (self.user_biases, (_lib.SIDE_USER, True)),
                             (ie, (_lib.SIDE_ITEM, False)), (ib, (_lib.SIDE_ITEM, True))):
            if module.weight is param:
                return slot
        return None

    def _can_fuse(self):
        # partial credit compares the metadata of the ITEMS (base_pipeline.py:626-634), while the
        # fused kernel would index it with the bucket ids: the bucket stage then takes the generic route
        if self._bucketed and self.hparams.get('metadata_for_loss'):
            return False
        return super()._can_fuse()

    def _bucket_ids(self, items):
        buckets = self.hparams.item_buckets
        if buckets.device != items.device:
            self.hparams.item_buckets = buckets = buckets.to(items.device)
        return buckets[items.long()]

    def _fused_calculate_loss(self, users, pos_items, neg_items):
        if self._bucketed:
            pos_items = self._bucket_ids(pos_items)
            neg_items = self._bucket_ids(neg_items)
        return super()._fused_calculate_loss(users, pos_items, neg_items)

    def forward(self, users: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
        """``cold_start_matrix_factorization.py:284-320``."""
        if self.training and self.hparams.dropout_p > 0:
            raise NotImplementedError(
                'dropout_p > 0 in training mode is not implemented in the HIP path '
                '(SURVEY.md §7: torch-RNG parity is infeasible); use dropout_p=0.')
        _lib.require_hip(self.user_embeddings.weight, 'ColdStartModel')
        _lib.require_hip(items, 'items')
        if self._bucketed:
            items = self._bucket_ids(items)
        ie, ib = self._item_modules()
        return _MFScore.apply(users, items, self.user_embeddings.weight, ie.weight,
                              self.user_biases.weight, ib.weight, None, bool(self.hparams.sparse),
                              self._status_flag()).squeeze()

    def item_bucket_item_similarity(self, item_bucket_id: int):
        """``cold_start_matrix_factorization.py:322-367``: cosine similarity of every item with one
        bucket (``collie_embedding_similarity`` over the item rows with the bucket row appended)."""
        import pandas as pd
        items = self._get_item_embeddings()
        table = torch.cat([items, self.item_bucket_embeddings.weight.data[[item_bucket_id]]])
        sims = ops.embedding_similarity(table, items.shape[0])[:items.shape[0]]
        return pd.Series(sims.cpu().numpy()).sort_values(ascending=False)
