"""Scoring users against the whole catalogue through the hybrid head (``get_preds``,
reference ``collie/metrics.py:77-107``) on the BASELINE configs[4] shape: the users x catalogue grid
with the item side of the first combined layer once per item (``HybridPretrainedModel.
score_catalogue``) against the same pairs pushed through ``model(users, items)``.

    python tools/hybrid_eval_bench.py [--users-scored 16] [--items 1000000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(argv=None):
    """Runs the measurement and returns the result dict (bench.py embeds it under ``hybrid``)."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--users', type=int, default=10_000_000)
    ap.add_argument('--items', type=int, default=1_000_000)
    ap.add_argument('--users-scored', type=int, default=16)
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args(argv)
    from collie_amd.hybrid import HybridPretrainedModel
    from collie_amd.metrics import _get_user_item_pairs, get_preds
    from collie_amd.model import MatrixFactorizationModel
    U, N, D, K, MC = args.users, args.items, 128, 10, 256
    dev = torch.device('cuda:0')

    class Shape:
        num_users, num_items, num_negative_samples = U, N, K

    torch.manual_seed(0)
    with torch.device(dev):
        mf = MatrixFactorizationModel(train=Shape(), embedding_dim=D, lr_scheduler_func=None)
        item_md = torch.randn(N, MC)
    model = HybridPretrainedModel(train=Shape(), trained_model=mf, item_metadata=item_md,
                                  item_metadata_layers_dims=[128], combined_layers_dims=[128, 64, 32],
                                  lr_scheduler_func=None).to(dev)
    del mf
    model.eval()
    user_ids = np.random.default_rng(1).integers(0, U, args.users_scored)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps, out

    def pairwise():
        uu, ii = _get_user_item_pairs(user_ids, N, dev)
        with torch.no_grad():
            return model(uu, ii).view(-1, N)

    t_grid, grid = timed(lambda: get_preds(model, user_ids, N, dev))
    with model.catalogue_scope():       # as inside evaluate_in_batches: item share kept between batches
        t_scoped, scoped = timed(lambda: get_preds(model, user_ids, N, dev))
    assert torch.equal(scoped, grid)
    t_pair, pair = timed(pairwise)
    model.check_status()
    err = float((grid - pair).abs().max() / pair.abs().max())
    pairs = args.users_scored * N
    # flops per pair through the whole head: metadata layer + first combined layer + tail
    f_pair = 2 * MC * 128 + 2 * (2 * D + 128) * 128 + 2 * (128 * 64 + 64 * 32 + 32)
    return {
        'workload': f'HybridPretrainedModel (configs[4] shape) scores of {args.users_scored} users x '
                    f'{N} items through get_preds',
        'grid_ms': t_grid * 1e3, 'grid_in_evaluate_ms': t_scoped * 1e3,
        'pairwise_ms': t_pair * 1e3, 'speedup': t_pair / t_grid,
        'speedup_in_evaluate': t_pair / t_scoped,
        'grid_pairs_per_s': pairs / t_grid, 'grid_in_evaluate_pairs_per_s': pairs / t_scoped,
        'pairwise_pairs_per_s': pairs / t_pair,
        'pairwise_tflops': pairs * f_pair / t_pair / 1e12,
        'max_rel_difference': err}


def main():
    print(json.dumps(run()))


if __name__ == '__main__':
    main()
