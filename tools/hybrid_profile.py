"""Which torch ops does a hybrid-head training step still launch?  torch.profiler table of one
steady-state step of tools/hybrid_bench.py's model (ops with device time, with input shapes).
python tools/hybrid_profile.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collie_amd.hybrid import HybridPretrainedModel      # noqa: E402
from collie_amd.model import MatrixFactorizationModel    # noqa: E402

U, N, D, K, MC, B = 10_000_000, 1_000_000, 128, 10, 256, 65536
dev = torch.device('cuda:0')


class Shape:
    num_users, num_items, num_negative_samples = U, N, K


torch.manual_seed(0)
with torch.device(dev):
    mf = MatrixFactorizationModel(train=Shape(), embedding_dim=D, lr_scheduler_func=None)
    item_md = torch.randn(N, MC)
model = HybridPretrainedModel(train=Shape(), trained_model=mf, item_metadata=item_md,
                              item_metadata_layers_dims=[128], combined_layers_dims=[128, 64, 32],
                              freeze_embeddings=True, loss='bpr', optimizer='adam', lr=1e-3,
                              lr_scheduler_func=None).to(dev)
model.train()
opt = model.configure_optimizers()
g = torch.Generator(device=dev)
g.manual_seed(1)
batches = [((torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32),
             torch.randint(0, N, (B,), device=dev, generator=g, dtype=torch.int32)),
            torch.randint(0, N, (B, K), device=dev, generator=g)) for _ in range(4)]


def step(batch):
    opt.zero_grad()
    loss = model.calculate_loss(batch)
    loss.backward()
    opt.step()
    return loss


for i in range(5):
    step(batches[i % 4])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(5):
        step(batches[i % 4])
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='cuda_time_total', row_limit=40,
                                                         max_name_column_width=48,
                                                         max_shapes_column_width=60))
