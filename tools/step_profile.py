"""cProfile of the host side of the small-batch training step (launch-bound regime)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device('cuda:0')
model = bench.build_model(dev)
opts = model.configure_optimizers()
batches = bench.synthetic_batches(16, B, dev, 'uniform', 1)
for i in range(50):
    bench.train_step(model, opts, batches[i % 16])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1000):
    bench.train_step(model, opts, batches[i % 16])
torch.cuda.synchronize()
print(f'B={B}: {(time.perf_counter() - t0):.3f} ms/step (wall, 1000 steps)')
pr = cProfile.Profile()
pr.enable()
for i in range(1000):
    bench.train_step(model, opts, batches[i % 16])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
