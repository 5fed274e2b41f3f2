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


def sample(args, dev):
    """collie_sample_batch on the ML-20M shape: approximate and exact mode, shuffled order."""
    import numpy as np
    import scipy.sparse as sp
    nnz, U, N, K, B = 20_000_000, bench.NUM_USERS, bench.NUM_ITEMS, bench.NUM_NEG, args.batch
    rng = np.random.default_rng(0)
    keys = np.unique(rng.integers(0, U * N, nnz))
    rows, cols = (keys // N).astype(np.int32), (keys % N).astype(np.int32)
    csr = sp.coo_matrix((np.ones(len(keys), np.int8), (rows, cols)), shape=(U, N)).tocsr()
    csr.sort_indices()
    d = [torch.from_numpy(a).to(dev) for a in (rows, cols, csr.indptr.astype(np.int64),
                                               csr.indices.astype(np.int32))]
    order = torch.randperm(len(keys), device=dev)
    alg = B * (8 + 8 + 4 + 4 + 8 * K)           # order + (row, col) in, users + pos + negatives out
    for name, indptr, indices, tries in (('approximate', None, None, 0), ('exact', d[2], d[3], 200)):
        for _ in range(3):
            ops.sample_batch(d[0], d[1], order, 0, B, indptr, indices, N, K, tries, 1, 0)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for i in range(args.iters):
            ops.sample_batch(d[0], d[1], order, 0, B, indptr, indices, N, K, tries, 1, i * B)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / args.iters
        print(f'sample_batch {name}: B={B} K={K}: {ms * 1e3:.1f} us/launch, '
              f'{B / ms / 1e6:.2f} G samples/s, {alg / ms / 1e6:.0f} GB/s algorithmic '
              f'({alg / ms / 1e6 / 8000:.1%} of 8 TB/s)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('what', choices=['fwd', 'fwd_nograd', 'step', 'eval', 'sample'])
    ap.add_argument('--batch', type=int, default=1048576)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--users', type=int, default=16384)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    if args.what == 'sample':
        return sample(args, dev)
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
