"""Wall time per training step at the reference's default batch size (B = 1,024, ML-20M shape):
the step issued launch by launch against the replayed HIP graph (collie_amd/captured.py).
python tools/small_batch_bench.py [--batch 1024] [--steps 3000]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=3000)
    ap.add_argument('--zipf', action='store_true', help='Zipf(1.05) ids: hot rows own long segments')
    ap.add_argument('--same-buffers', action='store_true',
                    help='feed the recording\'s own input buffers (no per-step batch copy)')
    args = ap.parse_args()
    from collie_amd.captured import CapturedTrainStep
    from collie_amd.model import MatrixFactorizationModel
    dev = torch.device('cuda:0')
    U, N, D, K, B = 138493, 26744, 64, 10, args.batch

    class Shape:
        num_users, num_items, num_negative_samples = U, N, K

    out = {'batch': B, 'steps': args.steps}
    for mode in ('stepped', 'replayed', 'one_call'):
        torch.manual_seed(0)
        with torch.device(dev):
            model = MatrixFactorizationModel(train=Shape(), embedding_dim=D, sparse=True,
                                             loss='bpr', optimizer='sparse_adam',
                                             bias_optimizer='sgd', lr_scheduler_func=None)
        model.to(dev).train()
        opts = model.configure_optimizers()
        opts = list(opts) if isinstance(opts, (list, tuple)) else [opts]
        if args.zipf:
            import bench
            batches = bench.synthetic_batches(8, B, dev, 'zipf', 3)
        else:
            batches = [((torch.randint(0, U, (B,), device=dev, dtype=torch.int32),
                         torch.randint(0, N, (B,), device=dev, dtype=torch.int32)),
                        torch.randint(0, N, (B, K), device=dev)) for _ in range(8)]
        replay = CapturedTrainStep(model, opts, mode='call' if mode == 'one_call' else 'graph')

        def step(i):
            if mode != 'stepped':
                loss = replay(batches[i % 8])
                if loss is not None:
                    return
            for o in opts:
                o.zero_grad()
            loss = model.calculate_loss(batches[i % 8])
            loss.backward()
            for j, o in enumerate(opts):
                model.optimizer_step(epoch=0, batch_idx=i, optimizer=o, optimizer_idx=j)

        for i in range(300):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        out[mode] = {'us_per_step': wall / args.steps * 1e6,
                     'host_us_per_step': host / args.steps * 1e6,
                     'interactions_per_s': B * (K + 1) * args.steps / wall}
        if mode != 'stepped':
            out[mode]['replays'] = replay.replays
        if mode == 'replayed':
            # the recording alone, batch already in its buffers: the GPU time of the step
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                replay._graph.replay()
            torch.cuda.synchronize()
            out[mode]['graph_only_us'] = (time.perf_counter() - t0) / args.steps * 1e6
    print(json.dumps(out))


if __name__ == '__main__':
    main()
