"""
``CollieMinimalTrainer`` / ``CollieTrainer`` — the callers of the hot path
(reference: ``collie/model/base/trainer.py``).

``CollieMinimalTrainer`` mirrors the reference's plain-torch loop (``trainer.py:314-547``): same
constructor arguments, re-entrant ``fit`` that resumes from ``num_epochs_completed``, optimizers
re-created on every ``fit``, early stopping rule, ``ReduceLROnPlateau`` fallback, logger calls.
``pytorch_lightning`` is not available, so ``CollieTrainer`` keeps the reference constructor
(``trainer.py:60-79``: ``max_epochs``, ``benchmark``, ``deterministic``, ``**kwargs``) and drives
the same loop; Lightning-only keyword arguments are accepted and ignored.

The loop itself launches no arithmetic of its own: per step it is ``zero_grad`` ->
``calculate_loss`` (one fused HIP kernel) -> ``backward`` (arms the staged step) -> one
``optimizer.step()`` per optimizer (segment-reduce + row-update HIP kernels).  The detached loss
is accumulated on the device; the host only reads it when logging / verbosity ask for it
(``trainer.py:474-478``).
"""
import sys
import warnings
from typing import Optional, Union

import torch

from collie_amd.layers import MultiLRScheduler, MultiOptimizer


class CollieMinimalTrainer():
    def __init__(self,
                 model,
                 max_epochs: int = 10,
                 gpus: Optional[Union[bool, int]] = None,
                 logger=None,
                 early_stopping_patience: Optional[int] = 3,
                 log_every_n_steps: int = 50,
                 flush_logs_every_n_steps: int = 100,
                 enable_model_summary: bool = True,
                 weights_summary: Optional[str] = None,
                 detect_anomaly: bool = False,
                 terminate_on_nan: Optional[bool] = None,
                 benchmark: bool = True,
                 deterministic: bool = True,
                 progress_bar_refresh_rate: Optional[int] = None,
                 verbosity: Union[bool, int] = True):
        if gpus is None and torch.cuda.is_available():
            print('Detected GPU. Setting ``gpus`` to 1.')
            gpus = 1
        if logger is False:
            logger = None
        if early_stopping_patience is False:
            early_stopping_patience = None
        if verbosity is True:
            verbosity = 2
        elif verbosity is False:
            verbosity = 0
        if weights_summary is not None:
            warnings.warn('``weights_summary`` is deprecated and is replaced with '
                          '``enable_model_summary``.', DeprecationWarning)
        if terminate_on_nan is not None:
            warnings.warn('``terminate_on_nan`` is deprecated and is replaced with '
                          '``detect_anomaly``.', DeprecationWarning)
            if detect_anomaly is False:
                detect_anomaly = terminate_on_nan

        self.max_epochs = max_epochs
        self.gpus = gpus
        self.benchmark = benchmark
        self.deterministic = deterministic
        self.logger = logger
        self.early_stopping_patience = early_stopping_patience
        self.log_every_n_steps = log_every_n_steps
        self.flush_logs_every_n_steps = flush_logs_every_n_steps
        self.enable_model_summary = enable_model_summary
        self.weights_summary = weights_summary
        self.detect_anomaly = detect_anomaly
        self.terminate_on_nan = terminate_on_nan
        self.progress_bar_refresh_rate = progress_bar_refresh_rate
        self.verbosity = verbosity

        self.best_epoch_loss = (0, sys.maxsize)
        self.train_steps = 0
        self.val_steps = 0
        self.num_epochs_completed = 0

        if self.gpus is None or self.gpus is False or self.gpus == 0:
            self.device = 'cpu'
        else:
            self.device = 'cuda'

    @property
    def max_epochs(self):
        return self._max_epochs

    @max_epochs.setter
    def max_epochs(self, value: int):
        self._max_epochs = value

    def fit(self, model) -> None:
        """Runs the full optimization routine (``trainer.py:314-334``)."""
        if not getattr(self, 'first_run_pre_training_setup_complete_', False):
            self._pre_training_setup(model=model)
            self.first_run_pre_training_setup_complete_ = True
        self._initialize_optimizers_and_lr_schedulers(model=model)
        self._fit(model)

    def _fit(self, model) -> None:
        for epoch in range(self.num_epochs_completed + 1, self.max_epochs + 1):
            model.train()
            train_loss = self._train_loop_single_epoch(model, epoch)
            model.eval()

            epoch_summary = f'Epoch {epoch: >5}: train loss: {train_loss :<1.5f}'
            early_stop_loss = train_loss
            if self.logger is not None:
                self.logger.log_metrics(metrics={'train_loss_epoch': train_loss}, step=epoch)

            if self.val_dataloader is not None:
                val_loss = self._val_loop_single_epoch(model)
                epoch_summary += f', val loss: {val_loss :<1.5f}'
                early_stop_loss = val_loss
                if self.logger is not None:
                    self.logger.log_metrics(metrics={'val_loss_epoch': val_loss}, step=epoch)

            if self.logger is not None:
                self.logger.save()
            if self.verbosity >= 1:
                print(epoch_summary)

            model.hparams.num_epochs_completed += 1
            self.num_epochs_completed += 1

            if (
                self.early_stopping_patience is not None
                and early_stop_loss >= self.best_epoch_loss[1]
                and epoch >= (self.early_stopping_patience + self.best_epoch_loss[0])
            ):
                print(f'Epoch {epoch :>5}: Early stopping activated.')
                self._finalize_training()
                return

            if early_stop_loss < self.best_epoch_loss[1]:
                self.best_epoch_loss = (epoch, early_stop_loss)

            if self.lr_scheduler is not None:
                try:
                    self.lr_scheduler.step()
                except TypeError:
                    self.lr_scheduler.step(early_stop_loss)       # ``ReduceLROnPlateau``

        self._finalize_training()

    def _pre_training_setup(self, model) -> None:
        self.train_dataloader = model.train_dataloader()
        self.val_dataloader = model.val_dataloader()
        if self.verbosity != 0 and (self.weights_summary is not None
                                    or self.enable_model_summary is True):
            print(model)
        if self.logger is not None:
            self.logger.log_hyperparams(model.hparams)
            self.logger.save()
        model.to(self.device)
        model._move_any_external_data_to_device()

    def _initialize_optimizers_and_lr_schedulers(self, model) -> None:
        self.lr_scheduler = None
        ret = model.configure_optimizers()
        if isinstance(ret, tuple):
            optimizers, lr_schedulers = ret
            self.optimizer = MultiOptimizer(optimizers)
            self.lr_scheduler = MultiLRScheduler(lr_schedulers)
        elif isinstance(ret, list):
            self.optimizer = MultiOptimizer(ret)
        elif isinstance(ret, torch.optim.Optimizer):
            self.optimizer = MultiOptimizer([ret])
        else:
            raise ValueError('Unexpected output from ``model.configure_optimizers()``!')

    def _train_loop_single_epoch(self, model, epoch: int) -> float:
        """``trainer.py:445-485``."""
        total_loss = 0
        for batch_idx, batch in enumerate(self.train_dataloader):
            self.optimizer.zero_grad()
            batch = self._move_batch_to_device(batch)
            loss = model.calculate_loss(batch)
            loss.backward()
            for optimizer_idx, optimizer in enumerate(self.optimizer.optimizers):
                model.optimizer_step(epoch=epoch, batch_idx=batch_idx, optimizer=optimizer,
                                     optimizer_idx=optimizer_idx, optimizer_closure=None)
            self.train_steps += 1
            total_loss += loss.detach()
            self._log_step(name='train', steps=self.train_steps, total_loss=total_loss,
                           batch_idx=batch_idx)
        if hasattr(model, 'check_status'):
            model.check_status()
        return (total_loss / len(self.train_dataloader)).item()

    def _val_loop_single_epoch(self, model) -> float:
        """``trainer.py:487-504``."""
        total_loss = 0
        with torch.no_grad():
            for batch_idx, batch in enumerate(self.val_dataloader):
                batch = self._move_batch_to_device(batch)
                loss = model.calculate_loss(batch)
                self.val_steps += 1
                total_loss += loss.detach()
                self._log_step(name='val', steps=self.val_steps, total_loss=total_loss,
                               batch_idx=batch_idx)
        return (total_loss / len(self.val_dataloader)).item()

    def _move_batch_to_device(self, batch):
        """``trainer.py:506-532``."""
        try:
            ((users, pos_items), neg_items) = batch
            return ((users.to(self.device, non_blocking=True),
                     pos_items.to(self.device, non_blocking=True)),
                    neg_items.to(self.device, non_blocking=True))
        except (AttributeError, ValueError, TypeError):
            users, items, ratings = batch
            return users.to(self.device), items.to(self.device), ratings.to(self.device)

    def _log_step(self, name: str, steps: int, total_loss, batch_idx: int) -> None:
        if self.logger is not None:
            if steps % self.log_every_n_steps == 0:
                batch_loss = (total_loss / (batch_idx + 1)).item()
                self.logger.log_metrics(metrics={f'{name}_loss_step': batch_loss}, step=steps)
            if steps % self.flush_logs_every_n_steps == 0:
                self.logger.save()

    def _finalize_training(self) -> None:
        if self.logger is not None:
            self.logger.save()
            self.logger.finalize(status='FINISHED')


class CollieTrainer(CollieMinimalTrainer):
    """
    Reference constructor (``trainer.py:60-79``): ``CollieTrainer(model, max_epochs=10,
    benchmark=True, deterministic=True, **kwargs)``.  Lightning's own ``Trainer`` is not available
    here, so the same single-process loop as ``CollieMinimalTrainer`` runs (one forward/backward
    per batch, then every optimizer steps); keyword arguments that only Lightning understands
    (``callbacks``, ``enable_checkpointing`` ...) are accepted and ignored.
    """

    _OWN = ('gpus', 'logger', 'early_stopping_patience', 'log_every_n_steps',
            'flush_logs_every_n_steps', 'enable_model_summary', 'weights_summary',
            'detect_anomaly', 'terminate_on_nan', 'progress_bar_refresh_rate', 'verbosity')

    def __init__(self, model, max_epochs: int = 10, benchmark: bool = True,
                 deterministic: bool = True, **kwargs):
        own = {k: v for k, v in kwargs.items() if k in self._OWN}
        own.setdefault('early_stopping_patience', None)   # Lightning has no default early stop
        if own.get('logger') is True:
            own['logger'] = None
        super().__init__(model, max_epochs=max_epochs, benchmark=benchmark,
                         deterministic=deterministic, **own)
        self.ignored_kwargs = {k: v for k, v in kwargs.items() if k not in self._OWN}
