"""
``CollieMinimalTrainer`` / ``CollieTrainer`` — the callers of the hot path.

Written from the behaviour contract of the reference's trainers (``collie/model/base/trainer.py``:
constructor arguments ``:218-233``, re-entrant ``fit`` that resumes from ``num_epochs_completed``
``:314-338``, optimizers re-created by every ``fit`` ``:331``, the early-stopping rule ``:378-390``,
scheduler stepping with the ``ReduceLROnPlateau`` fallback ``:392-399``, logger calls
``:419-421,534-547``), not from its text.  Method and attribute names that user code and the
reference's tests reach into (``_train_loop_single_epoch``, ``_val_loop_single_epoch``,
``_move_batch_to_device``, ``optimizer``, ``lr_scheduler``, ``best_epoch_loss``, ``train_steps`` ...)
are part of that contract and are kept.

What a step costs here: ``zero_grad`` -> ``calculate_loss`` (one fused HIP kernel) -> ``backward``
(arms the staged step) -> one ``optimizer.step()`` per optimizer (segment-reduce + row-update HIP
kernels).  The running loss stays on the device; the host reads it once per epoch, when a logger
step is due, or when the progress display refreshes -- never once per step.

``detect_anomaly`` is honoured on both routes: the autograd route runs under
``torch.autograd.set_detect_anomaly`` like the reference (``:333``); the fused route has no autograd
graph to inspect, so the loss kernels raise a flag word when a loss is not finite and the trainer
checks it after every step (one device read-back per step: the price of anomaly detection).

``pytorch_lightning`` is not available, so ``CollieTrainer`` keeps the reference's constructor
(``:60-79``) and drives the same loop; Lightning-only keyword arguments are accepted and recorded in
``ignored_kwargs``.
"""
import sys
import warnings
from typing import Optional, Union

import torch

from collie_amd.layers import MultiLRScheduler, MultiOptimizer


def _deprecated(old: str, new: str) -> None:
    warnings.warn(f'``{old}`` is deprecated and is replaced with ``{new}``.', DeprecationWarning)


class _DeviceMean:
    """Running mean of 0-d loss tensors that never leaves the device until ``value()`` is read."""

    def __init__(self):
        self.total = 0
        self.count = 0

    def add(self, loss) -> None:
        self.total = self.total + loss.detach()
        self.count += 1

    def value(self, over: Optional[int] = None) -> float:
        mean = self.total / (over if over is not None else max(self.count, 1))
        return mean.item() if torch.is_tensor(mean) else float(mean)


class _Progress:
    """Epoch / step progress display for ``verbosity >= 2`` (tqdm when importable, silent else)."""

    def __init__(self, enabled: bool, refresh: Optional[int]):
        self.refresh = refresh
        self.tqdm = None
        if enabled:
            try:
                from tqdm.auto import tqdm
                self.tqdm = tqdm
            except ImportError:             # no progress library: the epoch summaries remain
                pass

    def wrap(self, iterable, **kwargs):
        if self.tqdm is None:
            return iterable
        return self.tqdm(iterable, miniters=self.refresh, **kwargs)

    def show_loss(self, bar, step_index: int, meter: _DeviceMean) -> None:
        """Running epoch loss on the bar, read back only every ``refresh`` (default 50) steps."""
        every = self.refresh or 50
        if self.tqdm is not None and hasattr(bar, 'set_postfix') and (step_index + 1) % every == 0:
            bar.set_postfix(train_loss=meter.value(), refresh=False)


class CollieMinimalTrainer():
    """
    Plain-loop trainer with the reference's constructor (``trainer.py:218-233``):
    ``CollieMinimalTrainer(model, max_epochs=10, gpus=None, logger=None,
    early_stopping_patience=3, log_every_n_steps=50, flush_logs_every_n_steps=100,
    enable_model_summary=True, weights_summary=None, detect_anomaly=False, terminate_on_nan=None,
    benchmark=True, deterministic=True, progress_bar_refresh_rate=None, verbosity=True)``.
    """

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
        if weights_summary is not None:
            _deprecated('weights_summary', 'enable_model_summary')
        if terminate_on_nan is not None:
            _deprecated('terminate_on_nan', 'detect_anomaly')
            detect_anomaly = detect_anomaly or terminate_on_nan
        if gpus is None and torch.cuda.is_available():
            print('Detected GPU. Setting ``gpus`` to 1.')
            gpus = 1

        self.max_epochs = max_epochs
        self.gpus = gpus
        self.device = 'cuda' if gpus else 'cpu'
        self.logger = None if logger is False else logger
        self.early_stopping_patience = (None if early_stopping_patience is False
                                        else early_stopping_patience)
        self.log_every_n_steps = log_every_n_steps
        self.flush_logs_every_n_steps = flush_logs_every_n_steps
        self.enable_model_summary = enable_model_summary
        self.weights_summary = weights_summary
        self.detect_anomaly = detect_anomaly
        self.terminate_on_nan = terminate_on_nan
        self.benchmark = benchmark
        self.deterministic = deterministic
        self.progress_bar_refresh_rate = progress_bar_refresh_rate
        self.verbosity = (2 if verbosity else 0) if isinstance(verbosity, bool) else verbosity

        # small batches are launch-bound: replay their step as one HIP graph (collie_amd/captured.py)
        self.capture_small_batches = True
        self._captured = None

        self.best_epoch_loss = (0, sys.maxsize)       # (epoch, loss) of the best epoch so far
        self.train_steps = 0
        self.val_steps = 0
        self.num_epochs_completed = 0
        self.first_run_pre_training_setup_complete_ = False

    # ------------------------------------------------------------------------------- properties
    @property
    def max_epochs(self) -> int:
        return self._max_epochs

    @max_epochs.setter
    def max_epochs(self, value: int) -> None:
        self._max_epochs = value

    # -------------------------------------------------------------------------------------- fit
    def fit(self, model) -> None:
        """Train until ``max_epochs`` epochs are completed in total (or early stopping fires).
        A second call continues from the epochs already completed, with fresh optimizers."""
        if not self.first_run_pre_training_setup_complete_:
            self._pre_training_setup(model=model)
            self.first_run_pre_training_setup_complete_ = True
        self._initialize_optimizers_and_lr_schedulers(model=model)
        with torch.autograd.set_detect_anomaly(bool(self.detect_anomaly)):
            self._fit(model)

    def _fit(self, model) -> None:
        progress = _Progress(self.verbosity >= 2, self.progress_bar_refresh_rate)
        self._progress = progress
        epochs = progress.wrap(range(self.num_epochs_completed + 1, self.max_epochs + 1),
                               position=0, unit='epoch', desc='')
        for epoch in epochs:
            monitored = self._one_epoch(model, epoch)
            if self._early_stop_due(epoch, monitored):
                print(f'Epoch {epoch :>5}: Early stopping activated.')
                break
            if monitored < self.best_epoch_loss[1]:
                self.best_epoch_loss = (epoch, monitored)
            self._step_schedulers(monitored)
        self._finalize_training()

    def _one_epoch(self, model, epoch: int) -> float:
        """Train (and validate) one epoch; returns the loss early stopping and plateau schedulers
        watch: the validation loss when there is a validation loader, else the training loss."""
        model.train()
        train_loss = self._train_loop_single_epoch(model, epoch)
        model.eval()
        report = [f'Epoch {epoch: >5}: train loss: {train_loss :<1.5f}']
        epoch_metrics = {'train_loss_epoch': train_loss}
        monitored = train_loss
        if self.val_dataloader is not None:
            monitored = self._val_loop_single_epoch(model)
            report.append(f'val loss: {monitored :<1.5f}')
            epoch_metrics['val_loss_epoch'] = monitored
        if self.logger is not None:
            for name, value in epoch_metrics.items():       # one call per metric, train first
                self.logger.log_metrics(metrics={name: value}, step=epoch)
            self.logger.save()
        if self.verbosity >= 1:
            print(', '.join(report))
        model.hparams.num_epochs_completed += 1
        self.num_epochs_completed += 1
        return monitored

    def _early_stop_due(self, epoch: int, loss: float) -> bool:
        """``trainer.py:378-390``: no improvement on the best epoch for ``patience`` epochs."""
        if self.early_stopping_patience is None:
            return False
        best_epoch, best_loss = self.best_epoch_loss
        return loss >= best_loss and epoch >= best_epoch + self.early_stopping_patience

    def _step_schedulers(self, monitored: float) -> None:
        if self.lr_scheduler is None:
            return
        try:
            self.lr_scheduler.step()
        except TypeError:                   # ``ReduceLROnPlateau`` wants the monitored value
            self.lr_scheduler.step(monitored)

    # -------------------------------------------------------------------------------- set-up
    def _pre_training_setup(self, model) -> None:
        """Once per trainer: loaders, model summary, hyper-parameter logging, device placement."""
        self.train_dataloader = model.train_dataloader()
        self.val_dataloader = model.val_dataloader()
        wants_summary = self.weights_summary is not None or self.enable_model_summary is True
        if self.verbosity != 0 and wants_summary:
            print(model)
        if self.logger is not None:
            self.logger.log_hyperparams(model.hparams)
            self.logger.save()
        model.to(self.device)
        model._move_any_external_data_to_device()

    def _initialize_optimizers_and_lr_schedulers(self, model) -> None:
        """Every ``fit`` starts from fresh optimizers (and schedulers), as the reference does."""
        configured = model.configure_optimizers()
        schedulers = None
        if isinstance(configured, tuple):
            configured, schedulers = configured
        if isinstance(configured, torch.optim.Optimizer):
            configured = [configured]
        if not isinstance(configured, list):
            raise ValueError('Unexpected output from ``model.configure_optimizers()``!')
        self.optimizer = MultiOptimizer(configured)
        self._captured = None                 # recordings belong to one set of optimizers
        self.lr_scheduler = MultiLRScheduler(schedulers) if schedulers is not None else None

    # ------------------------------------------------------------------------------ epoch loops
    def _train_loop_single_epoch(self, model, epoch: int) -> float:
        """One pass over the training loader; returns the mean of the per-step losses (the last,
        possibly smaller batch weighs as much as the others, ``trainer.py:485``)."""
        meter = _DeviceMean()
        progress = getattr(self, '_progress', None) or _Progress(False, None)
        check_anomaly = getattr(model, 'check_anomaly', None) if self.detect_anomaly else None
        steps = progress.wrap(self.train_dataloader, total=len(self.train_dataloader), unit='step',
                              desc=f'({epoch :^5})', leave=False)
        replay = self._step_replayer(model) if check_anomaly is None else None
        for batch_idx, batch in enumerate(steps):
            batch = self._move_batch_to_device(batch)
            loss = replay(batch) if replay is not None else None
            if loss is None:
                self.optimizer.zero_grad()
                loss = model.calculate_loss(batch)
                loss.backward()
                if check_anomaly is not None:
                    check_anomaly()         # before any optimizer applies a poisoned gradient
                for optimizer_idx, optimizer in enumerate(self.optimizer.optimizers):
                    model.optimizer_step(epoch=epoch, batch_idx=batch_idx, optimizer=optimizer,
                                         optimizer_idx=optimizer_idx, optimizer_closure=None)
            self.train_steps += 1
            meter.add(loss)
            progress.show_loss(steps, batch_idx, meter)
            self._log_step(name='train', steps=self.train_steps, total_loss=meter.total,
                           batch_idx=batch_idx)
        if hasattr(model, 'check_status'):
            model.check_status()            # out-of-range ids seen by any kernel of the epoch
        return meter.value(over=len(self.train_dataloader))

    def _step_replayer(self, model):
        """The whole loop body as one replayed HIP graph, for models and batch sizes where that
        applies (``collie_amd/captured.py``); ``None`` otherwise.  A model whose class overrides
        ``optimizer_step`` or ``calculate_loss`` is stepped the ordinary way."""
        if not self.capture_small_batches or self.device != 'cuda':
            return None
        from collie_amd.model import MatrixFactorizationModel
        if not isinstance(model, MatrixFactorizationModel) or \
                type(model).optimizer_step is not MatrixFactorizationModel.optimizer_step or \
                type(model).calculate_loss is not MatrixFactorizationModel.calculate_loss or \
                any(name in vars(model) for name in ('calculate_loss', 'optimizer_step',
                                                     'training_step', 'forward')):
            return None                       # (also methods replaced on the instance)
        if self._captured is None or self._captured.model is not model:
            from collie_amd.captured import CapturedTrainStep
            # 'call': one library call per step; measured faster than replaying a HIP graph on
            # ROCm (Replayed step), same results bit for bit
            self._captured = CapturedTrainStep(model, self.optimizer.optimizers, mode='call')
        return self._captured

    def _val_loop_single_epoch(self, model) -> float:
        """One pass over the validation loader without gradients."""
        meter = _DeviceMean()
        with torch.no_grad():
            for batch_idx, batch in enumerate(self.val_dataloader):
                meter.add(model.calculate_loss(self._move_batch_to_device(batch)))
                self.val_steps += 1
                self._log_step(name='val', steps=self.val_steps, total_loss=meter.total,
                               batch_idx=batch_idx)
        return meter.value(over=len(self.val_dataloader))

    def _move_batch_to_device(self, batch):
        """Implicit batches are ``((users, items), negatives)``, explicit ones ``(users, items,
        ratings)``; copies are queued without waiting for them."""
        def move(t):
            return t.to(self.device, non_blocking=True)

        if len(batch) == 2 and not torch.is_tensor(batch[0]):
            (users, pos_items), neg_items = batch
            return (move(users), move(pos_items)), move(neg_items)
        return tuple(move(t) for t in batch)

    # ---------------------------------------------------------------------------------- logging
    def _log_step(self, name: str, steps: int, total_loss, batch_idx: int) -> None:
        """Running epoch loss every ``log_every_n_steps`` steps; flush every
        ``flush_logs_every_n_steps`` (``trainer.py:534-541``)."""
        if self.logger is None:
            return
        if steps % self.log_every_n_steps == 0:
            running = total_loss / (batch_idx + 1)
            running = running.item() if torch.is_tensor(running) else float(running)
            self.logger.log_metrics(metrics={f'{name}_loss_step': running}, step=steps)
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
    here, so the loop above runs (one forward/backward per batch, then every optimizer steps);
    keyword arguments that only Lightning understands (``callbacks``, ``enable_checkpointing`` ...)
    are accepted and kept in ``ignored_kwargs``.  Lightning has no default early stopping and a
    ``logger=True`` default that means "Lightning's own": both map to "none" here.
    """

    _LOOP_KWARGS = frozenset((
        'gpus', 'logger', 'early_stopping_patience', 'log_every_n_steps',
        'flush_logs_every_n_steps', 'enable_model_summary', 'weights_summary', 'detect_anomaly',
        'terminate_on_nan', 'progress_bar_refresh_rate', 'verbosity'))

    def __init__(self, model, max_epochs: int = 10, benchmark: bool = True,
                 deterministic: bool = True, **kwargs):
        self.ignored_kwargs = {k: v for k, v in kwargs.items() if k not in self._LOOP_KWARGS}
        loop_kwargs = {k: v for k, v in kwargs.items() if k in self._LOOP_KWARGS}
        loop_kwargs.setdefault('early_stopping_patience', None)
        if loop_kwargs.get('logger') is True:
            loop_kwargs['logger'] = None
        super().__init__(model, max_epochs=max_epochs, benchmark=benchmark,
                         deterministic=deterministic, **loop_kwargs)
