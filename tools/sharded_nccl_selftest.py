"""World-size-1 RCCL run of the row-sharded step and evaluation: exercises every collective call of
collie_amd/sharded.py with the nccl backend on one GPU (argument dtypes, contiguity, split lists),
which the gloo tests cannot.  python tools/sharded_nccl_selftest.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collie_amd.sharded import ShardedMatrixFactorizationModel  # noqa: E402

os.environ['COLLIE_SHARDED_COLLECTIVES'] = '1'     # one rank, but every collective call is made
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
m = ShardedMatrixFactorizationModel(bench.NUM_USERS, bench.NUM_ITEMS, bench.NUM_NEG,
                                    embedding_dim=bench.DIM, loss='bpr', device=dev, seed=0)
batches = bench.synthetic_batches(4, 65536, dev, 'uniform', 1)
for i in range(6):
    # the three ways a step gets its plan: made by the step itself, on a side stream after the
    # previous step (plan_ahead), queued inside the previous step (next_batch=)
    loss = m.train_step(batches[i % 4], next_batch=batches[(i + 1) % 4] if i % 3 == 1 else None)
    if i % 3 == 0:
        m.plan_ahead(batches[(i + 1) % 4])
for i in range(3):      # users of this rank only (world 1: all of them): the users_local route
    loss = m.train_step(batches[i % 4], users_local=True,
                        next_batch=batches[(i + 1) % 4] if i < 2 else None)
torch.cuda.synchronize()
print('train ok, loss', float(loss))
host = m.state_dict_to_rank0(chunk_rows=50000)
print('streamed state_dict ok', {k: tuple(v.shape) for k, v in host.items()})
import scipy.sparse as sp
from collie_amd import metrics as M
rng = np.random.default_rng(0)
rows = np.repeat(np.arange(2000), 5)
cols = rng.integers(0, bench.NUM_ITEMS, rows.size)
mat = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(bench.NUM_USERS, bench.NUM_ITEMS))


class _I:
    def __init__(self, mat):
        self.mat = mat
        self.num_users, self.num_items = mat.shape


print('eval', m.evaluate_in_batches([M.mapk, M.mrr, M.auc], _I(mat), k=10, batch_size=512))
dist.destroy_process_group()
print('nccl selftest ok')
