"""
BUILD-CONTAINER ONLY (imports ``/root/reference``; listed in ``.gpurunignore``).

Makes the reference's own ``collie.model.matrix_factorization``, ``collie.model.base.trainer`` and
``collie.metrics`` importable VERBATIM in a container that lacks ``pytorch_lightning``,
``torchmetrics``, ``docstring_parser``, ``fire`` and ``tables``.  The
stand-ins registered here replace those MISSING THIRD-PARTY PACKAGES only, and only with
non-arithmetic behaviour (an ``nn.Module`` shell with an ``hparams`` dict, frame introspection for
the constructor arguments, an empty docstring parser).  No line of Collie is re-implemented: every
number the fixtures hold is computed by the reference's own Python driving the installed torch.
The one arithmetic dependency that is absent, ``torchmetrics.functional.auroc``, maps to
``sklearn.metrics.roc_auc_score`` -- the oracle the reference's own test pins it to
(tests/test_metrics.py:140-154 of the reference).
"""
import inspect
import sys
import types

import torch

REF = '/root/reference/collie'


class _HParams(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value


class _LightningModule(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.__dict__['hparams'] = _HParams()

    def save_hyperparameters(self, *args, **kwargs):
        for a in args:
            if isinstance(a, dict):
                self.hparams.update(a)

    def log(self, *args, **kwargs):
        pass

    @property
    def device(self):
        return next(self.parameters()).device

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def optimizer_step(self, epoch=None, batch_idx=None, optimizer=None, optimizer_idx=None,
                       optimizer_closure=None, **kwargs):
        optimizer.step(closure=optimizer_closure)


class _Trainer:
    """pytorch_lightning.Trainer is only subclassed by CollieTrainer (never used here)."""
    def __init__(self, *args, **kwargs):
        """Stand-in."""


def _get_init_args(frame):
    """Constructor arguments of the ``__init__`` frame, ``**kwargs`` flattened."""
    info = inspect.getargvalues(frame)
    out = {k: info.locals[k] for k in info.args if k != 'self'}
    if info.keywords:
        out.update(info.locals.get(info.keywords, {}))
    out.pop('__class__', None)
    return out


def _move_data_to_device(batch, device):
    if torch.is_tensor(batch):
        return batch.to(device)
    if isinstance(batch, (list, tuple)):
        return type(batch)(_move_data_to_device(b, device) for b in batch)
    return batch


class _Parsed:
    params, meta = [], []
    short_description = long_description = ''


def install():
    """Register the stand-ins and the bare package modules (skipping collie/__init__.py)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    pl = mod('pytorch_lightning', LightningModule=_LightningModule, Trainer=_Trainer)
    util = mod('pytorch_lightning.utilities', move_data_to_device=_move_data_to_device)
    parsing = mod('pytorch_lightning.utilities.parsing', get_init_args=_get_init_args)
    summary = mod('pytorch_lightning.utilities.model_summary',
                  ModelSummary=lambda model, max_depth=1: '<model summary stand-in>')
    loggers = mod('pytorch_lightning.loggers')
    logger = mod('pytorch_lightning.loggers.logger', Logger=type('Logger', (), {}))
    pl.utilities, util.parsing, util.model_summary = util, parsing, summary
    pl.loggers, loggers.logger = loggers, logger

    from sklearn.metrics import roc_auc_score

    def auroc(preds, target, **kwargs):
        return torch.tensor(roc_auc_score(target.cpu().numpy(), preds.cpu().numpy()),
                            dtype=torch.float32)

    tm = mod('torchmetrics', Metric=type('Metric', (), {}))
    tm.functional = mod('torchmetrics.functional', auroc=auroc)

    numpydoc = mod('docstring_parser.numpydoc',
                   NumpydocParser=type('NumpydocParser', (), {'parse': lambda self, s: _Parsed()}))
    mod('docstring_parser', numpydoc=numpydoc, DocstringMeta=type('DocstringMeta', (), {}))
    mod('fire')
    mod('tables')

    # only the TOP package is a bare module (its __init__ star-imports the MovieLens / fire CLI
    # code); collie.model, collie.model.base, collie.metrics ... import normally from the tree
    top = types.ModuleType('collie')
    top.__path__ = [REF]
    sys.modules['collie'] = top
