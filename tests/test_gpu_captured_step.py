"""CapturedTrainStep (collie_amd/captured.py): the replayed HIP graph of a training step must leave
the tables, the optimizer state and the losses bit-identical to the step issued launch by launch
(reference loop body: collie/model/base/trainer.py:458-470)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Shape:
    def __init__(self, U, N, K):
        self.num_users, self.num_items, self.num_negative_samples = U, N, K


def _model(U, N, K, D, optimizer, bias_optimizer, dev, **kw):
    from collie_amd.model import MatrixFactorizationModel
    torch.manual_seed(3)
    with torch.device(dev):
        m = MatrixFactorizationModel(train=_Shape(U, N, K), embedding_dim=D, sparse=True,
                                     loss='bpr' if K == 1 else 'adaptive_bpr', optimizer=optimizer,
                                     bias_optimizer=bias_optimizer, lr_scheduler_func=None, **kw)
    m.to(dev).train()
    opts = m.configure_optimizers()
    opts = list(opts) if isinstance(opts, (list, tuple)) else [opts]
    return m, opts


def _ordinary(model, opts, batch, i):
    for o in opts:
        o.zero_grad()
    loss = model.calculate_loss(batch)
    loss.backward()
    for j, o in enumerate(opts):
        model.optimizer_step(epoch=0, batch_idx=i, optimizer=o, optimizer_idx=j)
    return loss.detach().clone()


def _state_equal(ma, oa, mb, ob):
    for (na, pa), (nb, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), na
    for a, b in zip(oa, ob):
        for pa, pb in zip(a.param_groups[0]['params'], b.param_groups[0]['params']):
            sa, sb = a.state[pa], b.state[pb]
            assert set(sa) == set(sb)
            for k in sa:
                if torch.is_tensor(sa[k]):
                    assert torch.equal(sa[k].cpu(), sb[k].cpu()), k
                else:
                    assert sa[k] == sb[k], k


@pytest.mark.parametrize('mode', ['graph', 'call'])
@pytest.mark.parametrize('optimizer,bias_optimizer,K,steps', [
    ('sparse_adam', 'sgd', 10, 300),        # crosses the 255-step stamp-tag cycle
    ('adagrad', 'adagrad', 1, 40),
    ('sgd', 'sgd', 5, 40),
    ('sparse_adam', 'sparse_adam', 4, 60),
])
def test_replayed_steps_are_bit_identical_to_stepped_ones(optimizer, bias_optimizer, K, steps, mode):
    from collie_amd.captured import CapturedTrainStep
    dev = torch.device('cuda:0')
    U, N, D, B = 700, 300, 64, (256 if K != 5 else 3000)   # 3000: the 1,024-thread row kernel
    ma, oa = _model(U, N, K, D, optimizer, bias_optimizer, dev)
    mb, ob = _model(U, N, K, D, optimizer, bias_optimizer, dev)
    _state_equal(ma, oa, mb, ob)
    replay = CapturedTrainStep(mb, ob, mode=mode)
    g = torch.Generator(device='cpu').manual_seed(11)
    replayed = 0
    for i in range(steps):
        b_here = B if i % 17 != 16 else B - 37            # a ragged batch now and then: stepped
        batch = ((torch.randint(0, U, (b_here,), generator=g, dtype=torch.int32).to(dev),
                  torch.randint(0, N, (b_here,), generator=g, dtype=torch.int32).to(dev)),
                 torch.randint(0, N, (b_here, K), generator=g).to(dev))
        if i == steps // 2:                               # what an LR scheduler does between epochs
            for o in oa + ob:
                for grp in o.param_groups:
                    grp['lr'] *= 0.5
        la = _ordinary(ma, oa, batch, i)
        lb = replay(batch)
        if lb is None:
            lb = _ordinary(mb, ob, batch, i)
        else:
            replayed += 1
            lb = lb.clone()
        assert torch.equal(la, lb), i
    assert replayed >= steps - steps // 17 - 2
    assert replay.captures == 1
    _state_equal(ma, oa, mb, ob)
    mb.check_status()


def test_trainer_replays_small_batches_and_matches_the_ordinary_loop():
    """CollieMinimalTrainer with and without the replayed step: same epoch losses, same tables."""
    import scipy.sparse as sp
    from collie_amd.interactions import Interactions, InteractionsDataLoader
    from collie_amd.model import MatrixFactorizationModel
    from collie_amd.trainer import CollieMinimalTrainer
    rng = np.random.default_rng(0)
    U, N = 120, 90
    mat = sp.random(U, N, density=0.2, random_state=1, format='coo')
    mat.data[:] = 1
    results = []
    for capture in (False, True):
        torch.manual_seed(5)
        np.random.seed(5)
        import random
        random.seed(5)
        inter = Interactions(mat=mat, num_negative_samples=5, seed=7,
                             allow_missing_ids=True, check_num_negative_samples_is_valid=False)
        loader = InteractionsDataLoader(inter, batch_size=128, shuffle=False)
        model = MatrixFactorizationModel(train=loader, embedding_dim=32, lr=1e-2,
                                         optimizer='sparse_adam', bias_optimizer='sgd', sparse=True,
                                         lr_scheduler_func=None)
        trainer = CollieMinimalTrainer(model, max_epochs=3, logger=False, verbosity=0,
                                       early_stopping_patience=None)
        trainer.capture_small_batches = capture
        trainer.fit(model)
        if capture:
            assert trainer._captured is not None and trainer._captured.replays > 0
        results.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for k in results[0]:
        assert torch.equal(results[0][k], results[1][k]), k


def test_batch_copy_handles_strides_and_id_dtypes():
    """collie_mf_batch_copy: any strides of the negatives, int32 / int64 ids on either side."""
    from collie_amd import ops
    dev = torch.device('cuda:0')
    B, K = 777, 7
    g = torch.Generator().manual_seed(2)
    users = torch.randint(0, 1000, (B,), generator=g).to(dev)                     # int64
    pos = torch.randint(0, 1000, (B,), generator=g, dtype=torch.int32).to(dev)
    neg_t = torch.randint(0, 1000, (K, B), generator=g).to(dev)                   # [K, B] storage
    neg = neg_t.t()                                                               # strided [B, K] view
    du = torch.zeros(B, dtype=torch.int32, device=dev)
    dp = torch.zeros(B, dtype=torch.int64, device=dev)
    dn = torch.zeros(B, K, dtype=torch.int64, device=dev)
    ops.batch_copy(ops.make_batch(users, pos, neg), ops.make_batch(du, dp, dn))
    assert torch.equal(du.long(), users) and torch.equal(dp, pos.long()) and torch.equal(dn, neg)


@pytest.mark.parametrize('B,dense,D,want', [(1024, False, 64, False), (4096, False, 64, False),
                                            (4097, False, 64, True), (1024, True, 64, True),
                                            (1024, False, 62, True), (1024, False, 260, True)])
def test_update_needs_segments_rule(B, dense, D, want):
    from collie_amd import _lib, ops
    dev = torch.device('cuda:0')
    t = _lib.make_tables(torch.zeros(10, D, device=dev), torch.zeros(10, D, device=dev),
                         torch.zeros(10, 1, device=dev), torch.zeros(10, 1, device=dev))
    b = ops.make_batch(torch.zeros(B, dtype=torch.int32, device=dev),
                       torch.zeros(B, dtype=torch.int32, device=dev),
                       torch.zeros(B, 3, dtype=torch.int64, device=dev))
    assert ops.train_update_needs_segments(t, b, dense) is want
