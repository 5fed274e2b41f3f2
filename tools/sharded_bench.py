"""Time the row-sharded train step at world size 1 (exchange degenerates to local copies): shows
what the sharded route costs on top of the fused single-GPU step.  python tools/sharded_bench.py B"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collie_amd.sharded import ShardedMatrixFactorizationModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if len(sys.argv) > 3:
    bench.set_shape(sys.argv[3])            # c2 (default) | c4
out_json = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4].endswith('.json') else None
users_local = len(sys.argv) > 5 and sys.argv[5] == 'users_local'   # one rank owns every user
dev = torch.device('cuda:0')
m = ShardedMatrixFactorizationModel(bench.NUM_USERS, bench.NUM_ITEMS, bench.NUM_NEG,
                                    embedding_dim=bench.DIM, loss='bpr', device=dev)
batches = bench.synthetic_batches(8, B, dev, 'uniform', 1)
for i in range(5):
    m.train_step(batches[i % 8], users_local=users_local)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    m.train_step(batches[i % 8], users_local=users_local)
torch.cuda.synchronize()
dt = dt0 = (time.perf_counter() - t0) / steps
print(f'B={B}: {dt * 1e3:.3f} ms/step, {B * 11 / dt / 1e9:.3f} G interactions/s, '
      f'rows/step {m.last_exchange}')
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    m.train_step(batches[i % 8], users_local=users_local)
    if i + 1 < steps:
        m.plan_ahead(batches[(i + 1) % 8], users_local=users_local)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f'B={B} with plan_ahead: {dt * 1e3:.3f} ms/step, {B * 11 / dt / 1e9:.3f} G interactions/s')
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    m.train_step(batches[i % 8], users_local=users_local,
                 next_batch=batches[(i + 1) % 8] if i + 1 < steps else None)
torch.cuda.synchronize()
dt_next = (time.perf_counter() - t0) / steps
print(f'B={B} with next_batch: {dt_next * 1e3:.3f} ms/step, {B * 11 / dt_next / 1e9:.3f} G interactions/s')
m.profile_phases(True)
for i in range(10):
    m.train_step(batches[i % 8], users_local=users_local,
                 next_batch=batches[(i + 1) % 8] if i + 1 < 10 else None)
phases = {k: round(v, 3) for k, v in m.phase_times().items()}
print('phase ms:', phases)
m.profile_phases(False)
if out_json:
    import json
    with open(out_json, 'w') as f:
        json.dump({'shape': sys.argv[3] if len(sys.argv) > 3 else 'c2', 'world': 1, 'batch': B, 'users_local': users_local,
                   'ms_per_step': round(dt0 * 1e3, 3), 'ms_per_step_plan_ahead': round(dt * 1e3, 3),
                   'ms_per_step_next_batch': round(dt_next * 1e3, 3),
                   'phase_ms': phases, 'exchange': m.last_exchange}, f, indent=1)
if os.environ.get('PROFILE'):
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(5):
            m.train_step(batches[i % 8], users_local=users_local)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25))
