"""cProfile of evaluate_in_batches([mapk]) on the configs[2] shape (bench.py's eval leg): where the
host time around the scoring kernel goes.  python tools/eval_profile.py [n_users]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collie_amd import metrics as M  # noqa: E402
from collie_amd.interactions import Interactions  # noqa: E402
from collie_amd.model import MatrixFactorizationModel  # noqa: E402

n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device('cuda:0')
U, N, D = bench.EVAL_USERS, bench.EVAL_ITEMS, bench.EVAL_DIM
rng = np.random.default_rng(0)
users = np.repeat(np.sort(rng.choice(U, n_users, replace=False)), 10)
items = rng.integers(0, N, len(users))
test = Interactions(users=users, items=items, num_users=U, num_items=N, allow_missing_ids=True,
                    check_num_negative_samples_is_valid=False, num_negative_samples=1,
                    max_number_of_samples_to_consider=0, seed=0)


class Shape:
    num_users, num_items, num_negative_samples = U, N, 10


with torch.device(dev):
    model = MatrixFactorizationModel(train=Shape(), embedding_dim=D, lr_scheduler_func=None)
model.to(dev).eval()
M.evaluate_in_batches([M.mapk], test, model, k=10, verbose=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
M.evaluate_in_batches([M.mapk], test, model, k=10, verbose=False)
torch.cuda.synchronize()
print(f'evaluate_in_batches([mapk]): {time.perf_counter() - t0:.4f} s for {n_users} users')
pr = cProfile.Profile()
pr.enable()
M.evaluate_in_batches([M.mapk], test, model, k=10, verbose=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
