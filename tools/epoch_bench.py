"""One training epoch END TO END through CollieMinimalTrainer (the reference's manual loop,
trainer.py:445-485) on the synthetic ML-20M shape: host loaders (numpy sampling + a host->device
copy per batch, as the reference does) vs the device-resident loader (collie_sample_batch).
usage: python tools/epoch_bench.py [batch_size ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collie_amd.interactions import (ApproximateNegativeSamplingInteractionsDataLoader,  # noqa: E402
                                     DeviceInteractionsDataLoader, Interactions)
from collie_amd.model import MatrixFactorizationModel  # noqa: E402
from collie_amd.trainer import CollieMinimalTrainer  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1024, 65536, 1048576]
nnz = int(os.environ.get('NNZ', bench.NUM_INTERACTIONS))
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
rows = rng.integers(0, bench.NUM_USERS, nnz).astype(np.int32)
cols = rng.integers(0, bench.NUM_ITEMS, nnz).astype(np.int32)
inter = Interactions(users=rows, items=cols, num_users=bench.NUM_USERS, num_items=bench.NUM_ITEMS,
                     allow_missing_ids=True, num_negative_samples=bench.NUM_NEG,
                     max_number_of_samples_to_consider=0, seed=1,
                     check_num_negative_samples_is_valid=False)
print(f'{len(inter)} interactions after de-duplication')


def epoch(loader, tag):
    torch.manual_seed(0)
    model = MatrixFactorizationModel(train=loader, embedding_dim=bench.DIM, sparse=True, lr=1e-3,
                                     optimizer='sparse_adam', bias_optimizer='sgd', loss='bpr',
                                     lr_scheduler_func=None)
    tr = CollieMinimalTrainer(model, max_epochs=1, gpus=1, verbosity=0, logger=None,
                              early_stopping_patience=None, enable_model_summary=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.fit(model)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = len(inter) * (1 + bench.NUM_NEG)
    print(f'{tag}: {dt:.3f} s/epoch, {n / dt / 1e9:.3f} G interactions/s', flush=True)


for bs in sizes:
    epoch(DeviceInteractionsDataLoader(inter, batch_size=bs, shuffle=True, device=dev),
          f'device loader  B={bs}')
    if bs >= 65536 or os.environ.get('HOST_SMALL'):
        epoch(ApproximateNegativeSamplingInteractionsDataLoader(inter, batch_size=bs, shuffle=True),
              f'host loader    B={bs}')
