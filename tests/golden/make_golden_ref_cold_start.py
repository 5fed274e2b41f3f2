"""
Generates ``tests/golden/ref_cold_start.npz`` from the REFERENCE'S OWN ``ColdStartModel``
(``collie/model/cold_start_matrix_factorization.py`` + ``base/multi_stage_pipeline.py``) driven by
the reference's ``CollieMinimalTrainer``, both imported verbatim through
``tests/golden/ref_harness.py``.  Build container only.  Usage:
``python tests/golden/make_golden_ref_cold_start.py``.

One epoch over fixed batches in stage ``item_buckets``, ``advance_stage()`` (the bucket rows are
copied to the items, ``:213-241``), one epoch in stage ``no_buckets``: per-step losses and every
table after each stage.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.install()
from collie.model.cold_start_matrix_factorization import ColdStartModel    # noqa: E402  (reference)
from collie.model.base.trainer import CollieMinimalTrainer                 # noqa: E402  (reference)

TABLES = ('user_embeddings', 'item_bucket_embeddings', 'item_embeddings', 'user_biases',
          'item_bucket_biases', 'item_biases')


class FixedBatches(list):
    def __init__(self, batches, num_users, num_items, num_negative_samples):
        super().__init__(batches)
        self.num_users, self.num_items = num_users, num_items
        self.num_negative_samples = num_negative_samples


def tables_of(model):
    return {n: getattr(model, n).weight.detach().numpy().copy() for n in TABLES}


def case(out, tag, U, N, D, K, B, nb, n_buckets, seed, **kw):
    rng = np.random.default_rng(seed)
    batches = []
    for _ in range(nb):
        u = rng.integers(0, U, B).astype(np.int32)
        p = (rng.zipf(1.3, B) % N).astype(np.int32)
        neg = rng.integers(0, N, (B, K)).astype(np.int64)
        batches.append(((torch.from_numpy(u), torch.from_numpy(p)), torch.from_numpy(neg)))
    buckets = torch.tensor(rng.integers(0, n_buckets, N))
    torch.manual_seed(seed)
    model = ColdStartModel(train=FixedBatches(batches, U, N, K), item_buckets=buckets,
                           embedding_dim=D, lr_scheduler_func=None, **kw)
    with torch.no_grad():
        model.user_embeddings.weight.mul_(10.0)
        model.item_bucket_embeddings.weight.mul_(10.0)
        model.item_bucket_biases.weight.normal_(0, 0.05)
    out[f'{tag}|item_buckets'] = buckets.numpy()
    for n, v in tables_of(model).items():
        out[f'{tag}|t0|{n}'] = v
    losses = []
    inner = model.calculate_loss

    def recording(batch):
        loss = inner(batch)
        losses.append(float(loss.detach()))
        return loss

    model.calculate_loss = recording
    trainer = CollieMinimalTrainer(model, max_epochs=1, gpus=0, logger=None, verbosity=0,
                                   early_stopping_patience=None, enable_model_summary=False)
    trainer.fit(model)
    assert model.hparams.stage == 'item_buckets' and len(losses) == nb
    for n, v in tables_of(model).items():
        out[f'{tag}|t1|{n}'] = v
    model.advance_stage()
    assert model.hparams.stage == 'no_buckets'
    for n, v in tables_of(model).items():
        out[f'{tag}|t1_advanced|{n}'] = v
    trainer.max_epochs += 1
    trainer.fit(model)
    assert len(losses) == 2 * nb
    for n, v in tables_of(model).items():
        out[f'{tag}|t2|{n}'] = v
    out[f'{tag}|step_losses'] = np.asarray(losses, np.float64)
    out[f'{tag}|n_batches'] = np.int64(nb)
    for i, ((u, p), neg) in enumerate(batches):
        out[f'{tag}|b{i}|users'], out[f'{tag}|b{i}|pos'] = u.numpy(), p.numpy()
        out[f'{tag}|b{i}|neg'] = neg.numpy()
    out[f'{tag}|loss_name'] = np.str_(model.loss_function.__name__)
    print(tag, model.loss_function.__name__, 'losses', np.round(losses, 4))


def main():
    out = {}
    case(out, 'adam_hinge', U=120, N=90, D=10, K=5, B=256, nb=3, n_buckets=7, seed=5,
         item_buckets_stage_lr=1e-2, no_buckets_stage_lr=1e-2)      # the reference's defaults
    case(out, 'adagrad_sgd_bpr', U=100, N=80, D=16, K=4, B=200, nb=3, n_buckets=5, seed=6,
         item_buckets_stage_lr=5e-2, no_buckets_stage_lr=1e-1,
         item_buckets_stage_optimizer='adagrad', no_buckets_stage_optimizer='sgd', loss='bpr')
    np.savez_compressed(os.path.join(HERE, 'ref_cold_start.npz'), **out)
    print('wrote ref_cold_start.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
