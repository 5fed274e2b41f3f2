"""Runs single hot kernels of the path in isolation (for rocprofv3 --pmc passes and A/B timing).

usage: python tools/kernel_bench.py fwd|step|eval [--batch B] [--iters N]
"""
import argparse
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collie_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('what', choices=['fwd', 'fwd_nograd', 'step', 'eval'])
    ap.add_argument('--batch', type=int, default=1048576)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--users', type=int, default=16384)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    model = bench.build_model(dev)
    if args.what in ('fwd', 'fwd_nograd', 'step'):
        batches = bench.synthetic_batches(4, args.batch, dev, 'uniform', 1)
        opts = model.configure_optimizers()
        for i in range(3):
            bench.train_step(model, opts, batches[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.iters):
            if args.what == 'fwd_nograd':
                with torch.no_grad():
                    model.calculate_loss(batches[i % 4])
            elif args.what == 'fwd':
                for o in opts:
                    o.zero_grad()
                model.calculate_loss(batches[i % 4])
            else:
                bench.train_step(model, opts, batches[i % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        print(f'{args.what}: {dt * 1e3:.4f} ms/iter, B={args.batch}')
    else:
        model.eval()
        t = model._tables()
        ws = ops.EvalWorkspace(dev)
        users = torch.arange(args.users, dtype=torch.int64, device=dev)
        for _ in range(2):
            ops.eval_topk(t, users, 0, bench.NUM_ITEMS, 10, None, ws)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            ops.eval_topk(t, users, 0, bench.NUM_ITEMS, 10, None, ws)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        flops = 2.0 * args.users * bench.NUM_ITEMS * bench.DIM
        print(f'eval_topk: {dt * 1e3:.4f} ms/iter, {args.users / dt:.0f} users/s, '
              f'{flops / dt / 1e12:.2f} TFLOP/s')


if __name__ == '__main__':
    main()
