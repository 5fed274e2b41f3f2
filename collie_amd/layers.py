"""
Parameter containers of the MF model (reference: ``collie/model/base/layers.py``).

They are ``torch.nn.Embedding`` subclasses only so that ``state_dict()`` keys, shapes, the
``sparse`` flag and the initial distributions are those of the reference; their ``forward`` is not
on the hot path (all lookups happen inside the HIP kernels, straight from ``.weight``).
"""
from typing import Dict, List

import torch


class ScaledEmbedding(torch.nn.Embedding):
    """N(0, (1 / (2.5 * embedding_dim))^2) initialisation (``layers.py:6-10``)."""

    def reset_parameters(self) -> None:
        self.weight.data.normal_(0, 1.0 / (self.embedding_dim * 2.5))


class ZeroEmbedding(torch.nn.Embedding):
    """All-zero initialisation (``layers.py:13-17``)."""

    def reset_parameters(self) -> None:
        self.weight.data.zero_()


class MultiOptimizer(object):
    """``zero_grad`` over a list of optimizers (``layers.py:20-38``)."""

    def __init__(self, optimizers: List[torch.optim.Optimizer]):
        assert isinstance(optimizers, list), f'Expected list, got {type(optimizers)}!'
        self.optimizers = optimizers

    def zero_grad(self) -> None:
        for optimizer in self.optimizers:
            optimizer.zero_grad()


class MultiLRScheduler(object):
    """``step`` over a list of ``{'scheduler': ...}`` dicts (``layers.py:41-63``)."""

    def __init__(self, lr_schedulers: List[Dict[str, object]]):
        assert isinstance(lr_schedulers, list), f'Expected list, got {type(lr_schedulers)}!'
        self.lr_schedulers = [d['scheduler'] for d in lr_schedulers]

    def step(self, *args, **kwargs) -> None:
        for lr_scheduler in self.lr_schedulers:
            lr_scheduler.step(*args, **kwargs)
