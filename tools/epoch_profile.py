"""cProfile of one CollieMinimalTrainer epoch with the device loader at a small batch size: where
the host time of the launch-bound epoch goes.  python tools/epoch_profile.py [batch] [nnz]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collie_amd.interactions import DeviceInteractionsDataLoader, Interactions  # noqa: E402
from collie_amd.model import MatrixFactorizationModel  # noqa: E402
from collie_amd.trainer import CollieMinimalTrainer  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nnz = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
inter = Interactions(users=rng.integers(0, bench.NUM_USERS, nnz).astype(np.int32),
                     items=rng.integers(0, bench.NUM_ITEMS, nnz).astype(np.int32),
                     num_users=bench.NUM_USERS, num_items=bench.NUM_ITEMS, allow_missing_ids=True,
                     num_negative_samples=bench.NUM_NEG, max_number_of_samples_to_consider=0,
                     seed=1, check_num_negative_samples_is_valid=False)
loader = DeviceInteractionsDataLoader(inter, batch_size=bs, shuffle=True, device=dev)
torch.manual_seed(0)
model = MatrixFactorizationModel(train=loader, embedding_dim=bench.DIM, sparse=True, lr=1e-3,
                                 optimizer='sparse_adam', bias_optimizer='sgd', loss='bpr',
                                 lr_scheduler_func=None)
tr = CollieMinimalTrainer(model, max_epochs=1, gpus=1, verbosity=0, logger=None,
                          early_stopping_patience=None, enable_model_summary=False)
tr.fit(model)
torch.cuda.synchronize()
tr.max_epochs += 1
t0 = time.perf_counter()
tr.fit(model)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = len(loader)
print(f'epoch: {dt:.3f} s, {steps} steps, {dt / steps * 1e6:.1f} us per step')
tr.max_epochs += 1
pr = cProfile.Profile()
pr.enable()
tr.fit(model)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(32)
