import cProfile, pstats, sys, os, torch
sys.path.insert(0, '/root/repo')
from collie_amd.captured import CapturedTrainStep
from collie_amd.model import MatrixFactorizationModel
dev = torch.device('cuda:0')
U, N, D, K, B = 138493, 26744, 64, 10, 1024
class Shape: num_users, num_items, num_negative_samples = U, N, K
with torch.device(dev):
    model = MatrixFactorizationModel(train=Shape(), embedding_dim=D, sparse=True, loss='bpr', optimizer='sparse_adam', bias_optimizer='sgd', lr_scheduler_func=None)
model.to(dev).train()
opts = list(model.configure_optimizers())
batches = [((torch.randint(0, U, (B,), device=dev, dtype=torch.int32), torch.randint(0, N, (B,), device=dev, dtype=torch.int32)), torch.randint(0, N, (B, K), device=dev)) for _ in range(8)]
replay = CapturedTrainStep(model, opts)
def step(i):
    if replay(batches[i % 8]) is None:
        for o in opts: o.zero_grad()
        loss = model.calculate_loss(batches[i % 8]); loss.backward()
        for j, o in enumerate(opts): model.optimizer_step(epoch=0, batch_idx=i, optimizer=o, optimizer_idx=j)
for i in range(300): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(3000): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
