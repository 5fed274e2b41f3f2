"""Time the requester-side plan (collie_shard_plan) alone: python tools/plan_bench.py [B] [users_local]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collie_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
local = len(sys.argv) > 2 and sys.argv[2] == 'users_local'
bench.set_shape('c4')
dev = torch.device('cuda:0')
batches = bench.synthetic_batches(4, B, dev, 'uniform', 1)
planner = ops.ShardPlanner(bench.NUM_USERS, bench.NUM_ITEMS, 1, bench.DIM, dev)
for i in range(3):
    (u, p), n = batches[i % 4]
    planner.plan(u, p, n, 0 if local else -1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    (u, p), n = batches[i % 4]
    planner.plan(u, p, n, 0 if local else -1)
torch.cuda.synchronize()
print(f'plan B={B} users_local={local}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us')
