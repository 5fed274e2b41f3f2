"""
``torch.optim.Optimizer`` subclasses that ``MatrixFactorizationModel.configure_optimizers`` returns
(reference: ``collie/model/base/base_pipeline.py:342-479``).

Each class IS the torch optimizer the reference would have built (``SparseAdam``, ``Adam``,
``Adagrad``, ``SGD``): same constructor, ``param_groups`` (so LR schedulers work), same
``state_dict`` keys (``step``, ``exp_avg``, ``exp_avg_sq``, ``sum``).  ``step()`` differs: when the
model holds a staged fused backward (see ``model._FusedStep``) the parameters it owns are updated
by the ``collie_mf_train_update`` HIP kernel (sorted-segment reduce fused with the row update);
hyper-parameters are re-read from ``param_groups`` at every call.  If no fused step is staged but
ordinary ``.grad`` tensors exist (generic route: custom loss callables), the inherited torch
``step()`` runs unchanged.
"""
import torch

from collie_amd import _lib, ops


class _FusedMixin:
    _algo = None

    def attach(self, model):
        """Called by the model: lets ``step`` find the staged backward and the table slots."""
        self._collie_model = model
        return self

    def zero_grad(self, set_to_none: bool = True):
        model = getattr(self, '_collie_model', None)
        if model is not None:
            # Only a step that was already backward-ed (or consumed) is stale.  A step that is
            # staged but not yet armed belongs to the ``loss = calculate_loss(); zero_grad();
            # loss.backward(); step()`` order (Lightning's closure order) and must survive.
            st = model.__dict__.get('_staged_step')
            if st is not None and (st.armed or st.done):
                model._drop_staged_step()
        # the fused route never materialises ``.grad``: nothing to clear in the usual case
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    return super().zero_grad(set_to_none=set_to_none)
        return None

    def _staged(self):
        model = getattr(self, '_collie_model', None)
        if model is None:
            return None
        st = model._staged_step
        return st if (st is not None and st.armed) else None

    def step(self, closure=None):
        staged = self._staged()
        if staged is None or self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks:
            return self._torch_step(closure)
        return self._fused_step(staged, closure)

    def _fused_step(self, staged, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        model = self._collie_model
        dense = not model.hparams.sparse
        launches = {False: {}, True: {}}            # is_bias -> {side: (cfg, s1, s2, stamp)}
        for group in self.param_groups:
            for p in group['params']:
                slot = model._table_slot(p)
                if slot is None:
                    if p.grad is not None:
                        raise RuntimeError('fused optimizers only own the four MF tables')
                    continue
                if not p.requires_grad:
                    continue
                side, is_bias = slot
                launches[is_bias][side] = self._table_args(model, p, group, dense)
        for is_bias, per_side in launches.items():
            if per_side:
                staged.update(is_bias, dense, per_side)
        return loss

    # torch wraps ``step`` of every Optimizer class with profiler / hook plumbing (~45 us per call,
    # more than the launches of a small batch); a step marked ``hooked`` is left alone, and the
    # wrapped form is still used whenever hooks are registered or no fused step is staged.
    step.hooked = True

    def _torch_step(self, closure=None):
        base = next(c for c in type(self).__mro__
                    if c.__module__.startswith('torch.optim') and 'step' in c.__dict__)
        staged = self._staged()
        if staged is None:
            return base.step(self, closure)
        # hooks registered: run them around the fused update as torch's wrapper would
        from torch.optim.optimizer import _global_optimizer_pre_hooks, _global_optimizer_post_hooks
        for hook in list(_global_optimizer_pre_hooks.values()) + list(
                self._optimizer_step_pre_hooks.values()):
            hook(self, (closure,) if closure is not None else (), {})
        out = self._fused_step(staged, closure)
        for hook in list(self._optimizer_step_post_hooks.values()) + list(
                _global_optimizer_post_hooks.values()):
            hook(self, (closure,) if closure is not None else (), {})
        return out

    @staticmethod
    def _stamp(state, p):
        if '_row_stamp' not in state:
            state['_row_stamp'] = torch.zeros(p.shape[0], dtype=torch.int32, device=p.device)
        return state['_row_stamp']


def _step_count(state):
    s = state.get('step', 0)
    return int(s.item()) if torch.is_tensor(s) else int(s)


class FusedSparseAdam(_FusedMixin, torch.optim.SparseAdam):
    """``torch.optim.SparseAdam`` (``torch/optim/_functional.py:24-84``) on touched rows."""

    def _table_args(self, model, p, group, dense):
        if dense:
            raise RuntimeError('SparseAdam does not support dense gradients, please consider '
                               'Adam instead')
        state = self.state[p]
        if len(state) == 0:
            state['step'] = 0
            state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        state['step'] += 1
        beta1, beta2 = group['betas']
        cfg = ops.optim_cfg(_lib.OPT_SPARSE_ADAM, state['step'], group['lr'], beta1, beta2,
                            group['eps'])
        return cfg, state['exp_avg'], state['exp_avg_sq'], None


class FusedAdam(_FusedMixin, torch.optim.Adam):
    """``torch.optim.Adam`` (dense, ``torch/optim/adam.py:396-547``): every row, every step."""

    def _table_args(self, model, p, group, dense):
        if not dense:
            raise RuntimeError('Adam does not support sparse gradients, please consider '
                               'SparseAdam instead')
        if group.get('amsgrad') or group.get('maximize'):
            raise RuntimeError('amsgrad / maximize are not supported by the fused Adam')
        state = self.state[p]
        if 'exp_avg' not in state:
            state['step'] = torch.tensor(0.0, dtype=torch.float32)
            state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        state['step'] += 1
        beta1, beta2 = group['betas']
        cfg = ops.optim_cfg(_lib.OPT_ADAM, _step_count(state), group['lr'], beta1, beta2,
                            group['eps'], group['weight_decay'])
        return cfg, state['exp_avg'], state['exp_avg_sq'], self._stamp(state, p)


class FusedAdagrad(_FusedMixin, torch.optim.Adagrad):
    """``torch.optim.Adagrad`` (``torch/optim/adagrad.py:346-388``), sparse or dense gradients."""

    def _table_args(self, model, p, group, dense):
        state = self.state[p]           # torch creates 'step' and 'sum' at construction
        if state['sum'].device != p.device:
            state['sum'] = state['sum'].to(p.device)
        state['step'] += 1
        stamp = self._stamp(state, p) if (dense and group['weight_decay'] != 0) else None
        cfg = ops.optim_cfg(_lib.OPT_ADAGRAD, _step_count(state), group['lr'], eps=group['eps'],
                            weight_decay=group['weight_decay'], lr_decay=group['lr_decay'])
        return cfg, state['sum'], None, stamp


class FusedSGD(_FusedMixin, torch.optim.SGD):
    """``torch.optim.SGD`` with momentum 0 (the only form the reference constructs)."""

    def _table_args(self, model, p, group, dense):
        if group['momentum'] != 0 or group['nesterov'] or group.get('maximize'):
            raise RuntimeError('the fused SGD supports momentum=0 only')
        state = self.state[p]
        state['_steps'] = state.get('_steps', 0) + 1
        stamp = self._stamp(state, p) if (dense and group['weight_decay'] != 0) else None
        cfg = ops.optim_cfg(_lib.OPT_SGD, state['_steps'], group['lr'],
                            weight_decay=group['weight_decay'])
        return cfg, None, None, stamp


FUSED_OPTIMIZERS = {'sgd': FusedSGD, 'adagrad': FusedAdagrad, 'adam': FusedAdam,
                    'sparse_adam': FusedSparseAdam}
