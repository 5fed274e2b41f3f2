"""
The trainers' loop body -- ``zero_grad`` / ``calculate_loss`` / ``loss.backward()`` /
``optimizer.step()`` for every optimizer (reference: ``collie/model/base/trainer.py:458-470``) -- as
ONE replayed HIP graph.

At the reference's default batch size (1,024) a step is ~20 short kernels; issued one by one from
Python the host is the bottleneck (small-batch section).  ``CapturedTrainStep`` records
the launches of one step once (``torch.cuda.CUDAGraph`` over the library's own entry points) and
replays the recording for the following batches.  What changes from step to step is kept out of the
frozen kernel arguments:

* the batch is copied into the buffers the recording reads;
* the step id and the optimizers' step-dependent scalars come from a schedule in device memory that
  the library fills ahead (``collie_mf_train_schedule_steps``: the same host code that computes
  them by value, so a replayed trajectory is bit-identical to a stepped one);
* learning-rate changes (schedulers) re-fill the schedule, anything else that a recording froze
  (table storage, optimizer state tensors, betas / eps, batch shape) makes a new recording.

The optimizers' ``state['step']`` counters, the workspace's step counter and the model's staged-step
slot are kept exactly as a stepped batch leaves them, so stepped and replayed batches can be mixed.
"""
from ctypes import byref, c_void_p

import torch

from collie_amd import _lib, ops
from collie_amd.optim import _FusedMixin, _step_count

_TABLES = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')
_SLOTS = ((_lib.SIDE_USER, False), (_lib.SIDE_ITEM, False), (_lib.SIDE_USER, True),
          (_lib.SIDE_ITEM, True))


def _count(opt, p):
    state = opt.state[p]
    if '_steps' in state:
        return int(state['_steps'])
    return _step_count(state) if 'step' in state else 0


def _bump(opt, p):
    state = opt.state[p]
    if '_steps' in state:
        state['_steps'] += 1
    else:
        state['step'] += 1


class CapturedTrainStep:
    """``step = CapturedTrainStep(model, optimizers); loss = step(batch)``.

    ``step(batch)`` returns ``None`` when this batch cannot be replayed (the caller then steps it
    the usual way); otherwise the step has been queued and the returned 0-d tensor is the loss
    buffer of the recording: read or accumulate it before the next call.
    """

    def __init__(self, model, optimizers, max_batch=16384, mode='graph'):
        # mode 'graph': replay a recorded HIP graph; 'call': the same launches issued by ONE library
        # call per step (collie_mf_train_step), no recording -- see the measurements in DESIGN.md
        assert mode in ('graph', 'call')
        self.mode = mode
        self.model = model
        self.optimizers = list(optimizers)
        self.max_batch = max_batch
        self._graph = None
        self._graph_key = None
        self._owners_cache = None   # owners of the last fully checked call
        self._next_sig = None       # signature the next call must show to be replayed unchecked
        self._left = 0              # schedule entries not yet replayed
        self.replays = 0
        self.captures = 0

    # ------------------------------------------------------------------------------ eligibility
    def _owners(self):
        """[(optimizer, group, param) or None] per table, or None if a table's optimizer is not one
        of the fused classes (or has hooks): then nothing is replayed."""
        model = self.model
        weights = model._weights()
        owners = [None] * 4
        for opt in self.optimizers:
            if not isinstance(opt, _FusedMixin) or getattr(opt, '_collie_model', None) is not model:
                return None
            if opt._optimizer_step_pre_hooks or opt._optimizer_step_post_hooks:
                return None
            for group in opt.param_groups:
                for p in group['params']:
                    for i, w in enumerate(weights):
                        if w is p:
                            if owners[i] is not None:
                                return None
                            owners[i] = (opt, group, p)
                            break
                    else:
                        return None                     # a parameter that is not one of the tables
        for i, o in enumerate(owners):
            if o is not None and not o[2].requires_grad:
                owners[i] = None
        return owners if any(o is not None for o in owners) else None

    def _batch_ok(self, batch):
        model = self.model
        if not (model.training and torch.is_grad_enabled() and model._can_fuse()
                and model.hparams.sparse):
            return None
        try:
            (users, pos), neg = batch
        except (TypeError, ValueError):
            return None
        if not (torch.is_tensor(users) and torch.is_tensor(pos) and torch.is_tensor(neg)):
            return None
        if neg.dim() == 1:
            neg = neg.reshape(-1, 1)
        if (users.dim() != 1 or pos.dim() != 1 or neg.dim() != 2 or not users.is_cuda
                or users.shape[0] == 0 or users.shape[0] > self.max_batch
                or neg.shape[1] > _lib.MAX_NEG or pos.shape[0] != users.shape[0]
                or neg.shape[0] != users.shape[0]):
            return None
        return users.contiguous(), pos.contiguous(), neg

    # ----------------------------------------------------------------------------------- record
    def _record(self, users, pos, neg, owners):
        model = self.model
        dev = users.device
        self._u = users.clone()
        self._p = pos.clone()
        self._n = neg.contiguous().clone()
        B, K = self._n.shape
        ws = model._train_workspace(dev)
        self._buf = ws.ensure(B, K, model.hparams.embedding_dim, model.hparams.num_users,
                              model.hparams.num_items)
        self._tables = model._tables()
        self._hbatch = ops.make_batch(self._u, self._p, self._n)
        self._loss = torch.empty(1, dtype=torch.float32, device=dev)
        self._loss0 = self._loss.reshape(())
        # the loss configuration the model itself would use (built by a first ordinary step)
        cfg = model.__dict__.get('_loss_cfg')
        status = model._status_flag()
        per_table = [None if o is None else self._table_args(*o) for o in owners]
        self._status = status
        self._loss_cfg = cfg
        if self.mode == 'call':
            # nothing is recorded: the argument arrays of collie_mf_train_step, filled once
            self._cfgs = (_lib.OptimCfg * 4)()
            self._s1, self._s2, self._st = (c_void_p * 4)(), (c_void_p * 4)(), (c_void_p * 4)()
            for i, entry in enumerate(per_table):
                if entry is None:
                    self._cfgs[i].algo = -1
                    continue
                c, a, b, st = entry
                self._cfgs[i] = c
                self._s1[i] = a.data_ptr() if a is not None else None
                self._s2[i] = b.data_ptr() if b is not None else None
                self._st[i] = st.data_ptr() if st is not None else None
            self._owned = [i for i, e in enumerate(per_table) if e is not None]
            self._src = ops.make_batch(self._u, self._p, self._n)
            self._keep = per_table
            self._step_fn = _lib.load().collie_mf_train_step
            self._graph = None
            self.captures += 1
            return
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        # thread_local: calls other threads make meanwhile (a DataLoader's pin-memory thread) must
        # not invalidate the recording
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            # (a second captured stream for the loss and the bias update was tried: the fork / join
            # nodes cost more than the overlap gains -- a replayed step is bound by its node count)
            ops.train_forward(self._tables, self._hbatch, cfg, self._buf, 1 | ops.DEFER_LOSS,
                              ops.STEP_DEVICE, self._loss, None, None, status)
            if ops.train_update_needs_segments(self._tables, self._hbatch, False):
                ops.train_build_segments(self._tables, self._hbatch, self._buf)
            ops.train_update_all(self._tables, self._hbatch, self._buf, 1, False, ops.STEP_DEVICE,
                                 per_table, None)
            ops.train_advance_step(self._tables, self._hbatch, self._buf, self._loss, status)
        self._graph = graph
        self._src = ops.make_batch(self._u, self._p, self._n)    # refilled per call (pointers, strides)
        self._keep = per_table
        self.captures += 1

    @staticmethod
    def _table_args(opt, group, p):
        """(cfg, state1, state2, row_stamp) WITHOUT counting a step (``_table_args`` of the fused
        optimizers creates missing state and counts one)."""
        before = _count(opt, p)
        args = opt._table_args(opt._collie_model, p, group, False)
        state = opt.state[p]
        if '_steps' in state:
            state['_steps'] = before
        elif torch.is_tensor(state['step']):
            state['step'].fill_(before)
        else:
            state['step'] = before
        return args

    def _keys(self, users, pos, neg, owners):
        model = self.model
        ws = model._train_workspace(users.device)
        graph_key = [users.shape[0], neg.shape[1], users.dtype, pos.dtype, neg.dtype,
                     tuple(w.data_ptr() for w in model._weights()),
                     id(model.__dict__.get('_loss_cfg')), None if ws.buf is None else ws.buf.data_ptr()]
        sched_key = []
        for o in owners:
            if o is None:
                graph_key.append(None)
                sched_key.append(None)
                continue
            opt, group, p = o
            state = opt.state[p]
            graph_key.append((id(opt), tuple(group.get('betas', ())), group.get('eps'),
                              group.get('weight_decay'),
                              tuple(v.data_ptr() for k, v in sorted(state.items())
                                    if torch.is_tensor(v) and v.is_cuda)))
            sched_key.append((group['lr'], group.get('lr_decay')))
        return tuple(graph_key), tuple(sched_key)

    def _schedule(self, owners):
        model = self.model
        ws = model._workspace
        if self.mode == 'call':
            # by-value steps need no schedule: only the hyper-parameters are read again
            for i, o in enumerate(owners):
                if o is not None:
                    self._cfgs[i] = self._table_args(*o)[0]
            self._left = 1 << 60
            return
        cfgs = []
        for o in owners:
            if o is None:
                cfgs.append(None)
                continue
            opt, group, p = o
            cfg = self._table_args(opt, group, p)[0]
            cfg.step = _count(opt, p) + 1
            cfgs.append(cfg)
        n = ops.STEP_SCHEDULE_MAX
        ops.train_schedule_steps(self._tables, self._hbatch, self._buf, ws.step_id + 1, n, cfgs)
        self._left = n

    # ----------------------------------------------------------------------------------- replay
    _STEP_AT = 11       # position of the workspace step id in a signature

    def _signature(self, users, pos, neg, owners):
        """Everything a replay depends on that can change between two calls, as a flat list that is
        cheap to build (this runs every step: a launch-bound step has ~30 us of host time in all).
        Batch shape and dtypes, table / workspace storage, the step counters a stepped batch in
        between would have moved, and the learning rates."""
        model = self.model
        ws = model._workspace
        w = model._weights()
        if users.stride(0) != 1 or pos.stride(0) != 1:
            raise ValueError('strided ids')             # -> the slow path makes them contiguous
        sig = [users.shape[0], neg.shape[1], users.dtype, pos.dtype, neg.dtype, model.training,
               torch.is_grad_enabled(), w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(),
               w[3].data_ptr(), ws.step_id, 0 if ws.buf is None else ws.buf.data_ptr(),
               id(model.__dict__.get('_loss_cfg'))]
        # the other hyper-parameters a callback or scheduler may change between two steps (momentum
        # / beta schedules, weight decay): a change sends the step through the slow path, which
        # reads every cfg again.  Kept in front of the (count, lr) pairs, whose positions
        # ``_replay`` computes from the END of the list.
        for o in owners:
            if o is not None:
                g = o[1]
                sig.append((g.get('betas'), g.get('eps'), g.get('weight_decay'), g.get('lr_decay')))
        for o in owners:
            if o is not None:
                sig.append(_count(o[0], o[2]))
                sig.append(o[1]['lr'])
        return sig

    def __call__(self, batch):
        try:
            (users, pos), neg = batch
            if neg.dim() == 1:
                neg = neg.reshape(-1, 1)
            owners = self._owners_cache
            if owners is not None and self._signature(users, pos, neg, owners) == self._next_sig \
                    and self._left > 0:
                return self._replay(users, pos, neg, owners)
        except (TypeError, ValueError, AttributeError):
            pass
        return self._slow(batch)

    def _replay(self, users, pos, neg, owners):
        model = self.model
        src = self._src
        src.users = users.data_ptr()
        src.pos_items = pos.data_ptr()
        src.neg_items = neg.data_ptr()
        src.neg_stride_b = neg.stride(0)
        src.neg_stride_k = neg.stride(1)
        src._keepalive = (users, pos, neg)
        model.__dict__['_staged_step'] = None       # the workspace is overwritten
        sig = self._next_sig
        if self.mode == 'call':
            step_id = model._workspace.next_step()
            at = len(sig) - 2 * len(self._owned)
            cfgs = self._cfgs
            for i in self._owned:                   # this step's 1-based count of each table
                cfgs[i].step = sig[at] + 1
                at += 2
            buf = self._buf
            _lib.check(self._step_fn(byref(self._tables), byref(src), byref(self._loss_cfg),
                                     buf.data_ptr(), buf.numel(), step_id, 0,
                                     self._loss.data_ptr(), self._status.data_ptr(), cfgs, self._s1,
                                     self._s2, self._st, _lib.stream_ptr()), 'collie_mf_train_step')
        else:
            ops.batch_copy(src, self._hbatch)
            self._graph.replay()
            self._left -= 1
            model._workspace.next_step()
        # what the next call must find: one step on
        sig[self._STEP_AT] += 1
        at = len(sig) - 2 * sum(o is not None for o in owners)
        for o in owners:
            if o is not None:
                _bump(o[0], o[2])
                sig[at] += 1
                at += 2
        self.replays += 1
        return self._loss0

    def _slow(self, batch):
        """Full checks; records / schedules as needed; None if the batch has to be stepped."""
        self._owners_cache = None
        parts = self._batch_ok(batch)
        if parts is None:
            return None
        owners = self._owners()
        if owners is None:
            return None
        users, pos, neg = parts
        model = self.model
        if model.__dict__.get('_loss_cfg') is None or model._workspace is None:
            return None                 # the first step of a model is an ordinary one
        graph_key, sched_key = self._keys(users, pos, neg, owners)
        if graph_key != self._graph_key:
            if self._graph_key is not None and graph_key[:5] != self._graph_key[:5] and \
                    users.shape[0] < self._graph_key[0]:
                return None             # the ragged last batch of an epoch: stepped, not recorded
            try:
                self._record(users, pos, neg, owners)
            except RuntimeError:
                # the recording could not be made here (e.g. another capture is under way): this
                # object steps every batch the ordinary way from now on
                self._graph = self._graph_key = None
                self.max_batch = 0
                return None
            self._graph_key = self._keys(users, pos, neg, owners)[0]
        # a stepped batch, a new learning rate or an exhausted schedule brought us here
        self._schedule(owners)
        self._owners_cache = owners
        self._next_sig = self._signature(users, pos, neg, owners)
        return self._replay(users, pos, neg, owners)
