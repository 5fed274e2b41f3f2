"""
Generates ``tests/golden/ref_pipeline.npz`` from the REFERENCE'S OWN ``MatrixFactorizationModel``,
``CollieMinimalTrainer`` and ``collie.metrics`` (imported verbatim through
``tests/golden/ref_harness.py``).  Build container only.  Usage:
``python tests/golden/make_golden_ref_pipeline.py``.

F6: one ``CollieMinimalTrainer`` epoch on a 2,000-interaction synthetic with the
     index tensors of every batch fixed and dumped -> per-step losses, epoch loss, final tables.
F5:  per-user AP@k / RR / AUC of ``tests/golden/eval_case.npz``'s inputs from
     ``collie/metrics.py:144-282`` called one user at a time (the callables return batch means).
"""
import os
import sys
from functools import partial

import numpy as np
import scipy.sparse as sp
import torch
from torch.optim.lr_scheduler import ReduceLROnPlateau

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.install()
from collie.model.matrix_factorization import MatrixFactorizationModel  # noqa: E402  (reference)
from collie.model.base.trainer import CollieMinimalTrainer             # noqa: E402  (reference)
import collie.metrics as ref_metrics                                    # noqa: E402  (reference)

NAMES = ('user_embeddings', 'item_embeddings', 'user_biases', 'item_biases')


class FixedBatches(list):
    """A 'data loader' that replays fixed batches (what the trainer iterates and ``len()``s) and
    carries the three attributes the model constructor reads from a loader."""

    def __init__(self, batches, num_users, num_items, num_negative_samples):
        super().__init__(batches)
        self.num_users, self.num_items = num_users, num_items
        self.num_negative_samples = num_negative_samples


def tables_of(model):
    return {n: getattr(model, n).weight.detach().numpy().copy() for n in NAMES}


def trainer_epoch(out, tag, U, N, D, K, B, nnz, model_kw, seed):
    rng = np.random.default_rng(seed)
    users = rng.integers(0, U, nnz).astype(np.int32)
    items = (rng.zipf(1.3, nnz) % N).astype(np.int32)
    batches = []
    for lo in range(0, nnz, B):
        u, p = users[lo:lo + B], items[lo:lo + B]
        neg = rng.integers(0, N, (len(u), K)).astype(np.int64)
        batches.append(((torch.from_numpy(u), torch.from_numpy(p)), torch.from_numpy(neg)))
    loader = FixedBatches(batches, U, N, K)
    torch.manual_seed(seed)
    model = MatrixFactorizationModel(train=loader, embedding_dim=D, **model_kw)
    with torch.no_grad():                       # scores of O(1): a loss that moves per step
        model.user_embeddings.weight.mul_(10.0)
        model.item_embeddings.weight.mul_(10.0)
        model.item_biases.weight.normal_(0, 0.05)
    for n, v in tables_of(model).items():
        out[f'{tag}|t0|{n}'] = v
    losses = []
    inner = model.calculate_loss

    def recording(batch):
        loss = inner(batch)
        losses.append(float(loss.detach()))
        return loss

    model.calculate_loss = recording
    trainer = CollieMinimalTrainer(model, max_epochs=2, gpus=0, logger=None, verbosity=0,
                                   early_stopping_patience=None, enable_model_summary=False)
    captured = {}
    inner_epoch = trainer._train_loop_single_epoch

    def epoch_loop(model_, epoch):
        captured[epoch] = inner_epoch(model_, epoch)
        return captured[epoch]

    trainer._train_loop_single_epoch = epoch_loop
    trainer.fit(model)
    assert model.hparams.num_epochs_completed == 2 and len(losses) == 2 * len(batches)
    out[f'{tag}|step_losses'] = np.asarray(losses, np.float64)
    out[f'{tag}|epoch_losses'] = np.asarray([captured[1], captured[2]], np.float64)
    for n, v in tables_of(model).items():
        out[f'{tag}|t1|{n}'] = v
    out[f'{tag}|n_batches'] = np.int64(len(batches))
    for i, ((u, p), neg) in enumerate(batches):
        out[f'{tag}|b{i}|users'] = u.numpy()
        out[f'{tag}|b{i}|pos'] = p.numpy()
        out[f'{tag}|b{i}|neg'] = neg.numpy()
    out[f'{tag}|loss_name'] = np.str_(model.loss_function.__name__)
    print(tag, 'loss fn', model.loss_function.__name__, 'epoch losses', captured)


class ExplicitBatches(list):
    """Fixed (users, items, ratings) batches; no ``num_negative_samples``: explicit feedback."""

    def __init__(self, batches, num_users, num_items):
        super().__init__(batches)
        self.num_users, self.num_items = num_users, num_items


def explicit_epoch(out, tag, U, N, D, B, nb, model_kw, seed):
    """F6 for the explicit path (base_pipeline.py:635-650): two epochs of the reference's trainer on
    fixed (users, items, ratings) batches -> per-step losses and final tables."""
    rng = np.random.default_rng(seed)
    batches = []
    for _ in range(nb):
        batches.append((torch.from_numpy((rng.zipf(1.4, B) % U).astype(np.int32)),
                        torch.from_numpy(rng.integers(0, N, B).astype(np.int64)),
                        torch.from_numpy(rng.integers(1, 6, B).astype(np.float64))))
    torch.manual_seed(seed)
    model = MatrixFactorizationModel(train=ExplicitBatches(batches, U, N), embedding_dim=D, **model_kw)
    with torch.no_grad():
        model.user_embeddings.weight.mul_(30.0)
        model.item_embeddings.weight.mul_(30.0)
        model.user_biases.weight.normal_(0, 0.1)
        model.item_biases.weight.normal_(0, 0.1)
    for n, v in tables_of(model).items():
        out[f'{tag}|t0|{n}'] = v
    losses = []
    inner = model.calculate_loss

    def recording(batch):
        loss = inner(batch)
        losses.append(float(loss.detach()))
        return loss

    model.calculate_loss = recording
    trainer = CollieMinimalTrainer(model, max_epochs=2, gpus=0, logger=None, verbosity=0,
                                   early_stopping_patience=None, enable_model_summary=False)
    trainer.fit(model)
    assert len(losses) == 2 * nb
    out[f'{tag}|step_losses'] = np.asarray(losses, np.float64)
    for n, v in tables_of(model).items():
        out[f'{tag}|t1|{n}'] = v
    out[f'{tag}|n_batches'] = np.int64(nb)
    for i, (u, it, ra) in enumerate(batches):
        out[f'{tag}|b{i}|users'], out[f'{tag}|b{i}|items'] = u.numpy(), it.numpy()
        out[f'{tag}|b{i}|ratings'] = ra.numpy()
    print(tag, 'explicit losses', np.round(losses, 4))


def eval_case_per_user(out):
    z = np.load(os.path.join(HERE, 'eval_case.npz'))
    tables = {n: z[f't|{n}'] for n in NAMES}
    U, D = tables['user_embeddings'].shape
    N = tables['item_embeddings'].shape[0]
    loader = FixedBatches([], U, N, 1)
    model = MatrixFactorizationModel(train=loader, embedding_dim=D)
    with torch.no_grad():
        for n in NAMES:
            getattr(model, n).weight.copy_(torch.from_numpy(tables[n]))
    model.eval()
    rows, cols = z['rows'], z['cols']
    targets = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, N)).tocsr()
    users = np.unique(rows)
    k = 10
    ap, rr, auc, all_preds, top10 = [], [], [], [], []
    for u in users:
        ids = np.array([u])
        preds = ref_metrics.get_preds(model, ids, N, 'cpu')
        all_preds.append(preds[0].numpy().copy())
        top10.append(preds.topk(k, dim=1).indices[0].numpy().copy())
        ap.append(ref_metrics.mapk(targets, ids, preds, k))
        rr.append(ref_metrics.mrr(targets, ids, preds))
        auc.append(ref_metrics.auc(targets, ids, preds))
    out['eval|users'] = users
    out['eval|preds_reference'] = np.stack(all_preds)          # collie.metrics.get_preds
    out['eval|top10_ids_reference'] = np.stack(top10)          # preds.topk(k): tie order is torch's
    out['eval|ap_reference'] = np.asarray(ap, np.float64)
    out['eval|rr_reference'] = np.asarray(rr, np.float64)
    out['eval|auc_reference'] = np.asarray(auc, np.float64)
    out['eval|k'] = np.int64(k)
    print('eval per-user AP', np.round(ap, 4), 'RR', np.round(rr, 4), 'AUC', np.round(auc, 4))


def main():
    out = {}
    trainer_epoch(out, 'sparse_adam_bpr', U=120, N=90, D=16, K=5, B=256, nnz=2000, seed=3,
                  model_kw=dict(sparse=True, lr=1e-2, bias_lr=1e-2, optimizer='sparse_adam',
                                bias_optimizer='sgd', loss='bpr', lr_scheduler_func=None))
    trainer_epoch(out, 'default_adam_hinge', U=120, N=90, D=10, K=10, B=512, nnz=2000, seed=4,
                  # every default of the reference, except that its default scheduler factory
                  # passes ``verbose=True``, which torch 2.10's ReduceLROnPlateau no longer takes
                  model_kw=dict(lr=1e-2, lr_scheduler_func=partial(ReduceLROnPlateau, patience=1)))
    explicit_epoch(out, 'explicit_mse_yrange_adam', U=200, N=150, D=10, B=300, nb=3, seed=7,
                   model_kw=dict(lr=5e-2, bias_lr=5e-2, loss='mse', optimizer='adam',
                                 bias_optimizer='sgd', y_range=(0.0, 5.0), lr_scheduler_func=None))
    explicit_epoch(out, 'explicit_mae_sparse_adagrad', U=200, N=150, D=30, B=300, nb=3, seed=8,
                   model_kw=dict(sparse=True, lr=5e-2, bias_lr=5e-2, loss='mae', optimizer='adagrad',
                                 bias_optimizer='sgd', lr_scheduler_func=None))
    eval_case_per_user(out)
    np.savez_compressed(os.path.join(HERE, 'ref_pipeline.npz'), **out)
    print('wrote ref_pipeline.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
