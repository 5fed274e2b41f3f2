"""K1 (collie_mf_train_forward) alone on one of the bench shapes: A/B timing of kernel variants.

usage: python tools/k1_bench.py [--shape c2|c4] [--batch B] [--iters N] [--dist uniform|zipf]
Prints avg launch time (HIP events on the launch stream) and algorithmic GB/s (SURVEY.md 8d
forward bytes: ids + (2 + K) rows of D + 1 floats per positive).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collie_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='c4')
    ap.add_argument('--batch', type=int, default=1048576)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--neg', type=int, default=10)
    ap.add_argument('--kind', default='adaptive_bpr')
    ap.add_argument('--user-range', type=int, default=0, help='draw user ids below this (0 = all)')
    ap.add_argument('--item-range', type=int, default=0)
    ap.add_argument('--sort-users', action='store_true')
    ap.add_argument('--no-grad', action='store_true')
    ap.add_argument('--no-stamps', action='store_true', help='COLLIE_NO_STAMPS: what the stamps cost')
    args = ap.parse_args()
    U, N, D = {'c2': (138_493, 26_744, 64), 'c4': (100_000_000, 10_000_000, 128),
               'c3': (1_000_000, 1_000_000, 128)}[args.shape]
    K, B = args.neg, args.batch
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    ue = torch.empty(U, D, device=dev).normal_(0, 0.1, generator=g)
    ie = torch.empty(N, D, device=dev).normal_(0, 0.1, generator=g)
    ub = torch.zeros(U, 1, device=dev)
    ib = torch.empty(N, 1, device=dev).normal_(0, 0.01, generator=g)
    t = _lib.make_tables(ue, ie, ub, ib)
    batches = []
    for _ in range(4):
        ur, ir = args.user_range or U, args.item_range or N
        users = torch.randint(0, ur, (B,), device=dev, generator=g, dtype=torch.int32)
        if args.sort_users:
            users = users.sort().values.contiguous()
        pos = torch.randint(0, ir, (B,), device=dev, generator=g, dtype=torch.int32)
        neg = torch.randint(0, ir, (B, K), device=dev, generator=g)
        batches.append(ops.make_batch(users, pos, neg))
    ws = ops.TrainWorkspace(dev)
    buf = ws.ensure(B, K, D, U, N)
    cfg = ops.make_loss_cfg(args.kind, N)
    loss = torch.empty(1, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    need = (0 if args.no_grad else 1) | (ops.NO_STAMPS if args.no_stamps else 0)
    for i in range(3):
        ops.train_forward(t, batches[i % 4], cfg, buf, need, ws.next_step(), loss, None, None, status)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(args.iters):
        ops.train_forward(t, batches[i % 4], cfg, buf, need, ws.next_step(), loss, None, None, status)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / args.iters
    alg = B * 8 + B * K * 8 + (2 + K) * B * (D + 1) * 4
    print(f'K1 {args.shape} B={B} K={K} D={D}: {ms * 1e3:.1f} us/launch (incl. loss finalize), '
          f'{alg / ms / 1e6:.0f} GB/s algorithmic = {alg / ms / 1e6 / 8000:.1%} of 8 TB/s, '
          f'loss {loss.item():.6f} status {int(status.item())}')


if __name__ == '__main__':
    main()
