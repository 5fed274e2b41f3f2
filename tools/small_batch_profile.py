"""cProfile of the product step at the reference's default batch size (B = 1,024): shows where the
host time of a launch-bound step goes.  python tools/small_batch_profile.py [--batch 1024]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=3000)
    args = ap.parse_args()
    from collie_amd.model import MatrixFactorizationModel
    dev = torch.device('cuda:0')
    U, N, D, K, B = 138493, 26744, 64, 10, args.batch

    class Shape:
        num_users, num_items, num_negative_samples = U, N, K

    torch.manual_seed(0)
    with torch.device(dev):
        model = MatrixFactorizationModel(train=Shape(), embedding_dim=D, sparse=True, loss='bpr',
                                         optimizer='sparse_adam', bias_optimizer='sgd',
                                         lr_scheduler_func=None)
    model.to(dev).train()
    opts = model.configure_optimizers()
    opts = list(opts) if isinstance(opts, (list, tuple)) else [opts]
    batches = [((torch.randint(0, U, (B,), device=dev, dtype=torch.int32),
                 torch.randint(0, N, (B,), device=dev, dtype=torch.int32)),
                torch.randint(0, N, (B, K), device=dev)) for _ in range(8)]

    def step(i):
        for o in opts:
            o.zero_grad()
        loss = model.calculate_loss(batches[i % 8])
        loss.backward()
        for j, o in enumerate(opts):
            model.optimizer_step(epoch=0, batch_idx=i, optimizer=o, optimizer_idx=j)

    for i in range(200):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    print(f'wall per step: {(time.perf_counter() - t0) / args.steps * 1e6:.1f} us')
    pr = cProfile.Profile()
    pr.enable()
    for i in range(args.steps):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
