"""
BASELINE configs[4]: HybridPretrainedModel-style MF + item-metadata MLP head on
a synthetic 10M users x 1M items, dim 128, item metadata fp32 [N, 256],
item_metadata_layers_dims=[128], combined_layers_dims=[128, 64, 32], adaptive BPR + 10 negatives,
Adam, ``freeze_embeddings`` True or False.  One step = calculate_loss -> backward -> optimizer.step
through the product API on a batch resident in HBM.  Prints one JSON line (bench.py's format) with
the MFMA roofline of the step: algorithmic flops = the GEMMs the reference's autograd performs
(forward, weight gradients, data gradients that reach a trainable tensor) over wall time.

    python tools/hybrid_bench.py [--batch 65536] [--steps 20] [--warmup 5] [--unfrozen]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA


def step_flops(m, d, meta_cols, meta_dims, comb_dims, unfrozen, m_back=None, user_rows=None):
    """2 * rows * n_in * n_out per GEMM: forward + weight gradient for every layer, data gradient
    for every layer whose input needs one (never the raw metadata; the embedding slices of the
    first combined layer only when the embeddings train).  ``m_back``: rows the backward GEMMs run
    on (default m: what the reference's autograd does; the library runs them on the 2 B rows whose
    score gradient can differ from zero).  ``user_rows``: rows the user embedding's share of the first
    combined layer's forward is computed on (default m; the library: once per sample, B)."""
    mb = m if m_back is None else m_back
    total = 0
    dims = [meta_cols] + meta_dims
    for i in range(1, len(dims)):
        total += 2 * (m + mb) * dims[i - 1] * dims[i]              # fwd + dW
        if i > 1:
            total += 2 * mb * dims[i - 1] * dims[i]                # dX
    comb = [2 * d + meta_dims[-1]] + comb_dims + [1]
    for i in range(1, len(comb)):
        total += 2 * (m + mb) * comb[i - 1] * comb[i]
        if i > 1:
            total += 2 * mb * comb[i - 1] * comb[i]
    total += 2 * mb * comb[1] * (meta_dims[-1] + (2 * d if unfrozen else 0))
    if user_rows is not None:
        total -= 2 * (m - user_rows) * d * comb[1]
    return total


def run(argv=None):
    """Runs the benchmark and returns the result dict (bench.py embeds it as its ``hybrid`` leg)."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--users', type=int, default=10_000_000)
    ap.add_argument('--items', type=int, default=1_000_000)
    ap.add_argument('--unfrozen', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=0.0,
                    help='also time the torch CPU op sequence of the reference (oracle/torch_path.py)')
    args = ap.parse_args(argv)
    from collie_amd.hybrid import HybridPretrainedModel
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
                                  freeze_embeddings=not args.unfrozen, loss='bpr', optimizer='adam',
                                  lr=1e-3, lr_scheduler_func=None).to(dev)
    del mf
    model.train()
    opt = model.configure_optimizers()
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    B = args.batch
    batches = [((torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32),
                 torch.randint(0, N, (B,), device=dev, generator=g, dtype=torch.int32)),
                torch.randint(0, N, (B, K), device=dev, generator=g)) for _ in range(4)]

    def step(batch):
        opt.zero_grad()
        loss = model.calculate_loss(batch)
        loss.backward()
        opt.step()
        return loss

    for i in range(args.warmup):
        step(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(batches[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    model.check_status()
    m = B * (1 + K)
    ref_flops = step_flops(m, D, MC, [128], [128, 64, 32], args.unfrozen)
    # executed: the backward runs on the 2 B rows that can carry a score gradient (adaptive BPR),
    # the user embedding's share of the first combined layer once per sample
    flops = step_flops(m, D, MC, [128], [128, 64, 32], args.unfrozen, m_back=2 * B, user_rows=B)
    out = {
        'metric': 'train interactions/sec (pos+negs)', 'value': m / dt, 'unit': 'interactions/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[4]: HybridPretrainedModel head on synthetic {U} x '
                               f'{N}, dim {D}, {MC}-d item metadata -> [128], combined [128, 64, 32], '
                               f'adaptive BPR + {K} negatives, Adam, freeze_embeddings='
                               f'{not args.unfrozen}',
                   'batch_per_gpu': B, 'final_loss': float(loss.item())},
        'roofline': {'bound': 'mfma', 'achieved': flops / dt / 1e12, 'peak': MFMA_F32_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     'traffic': None, 'algorithmic_flops_per_step': flops,
                     'reference_autograd_flops_per_step': ref_flops,
                     'reference_autograd_flops_over_time_frac':
                         ref_flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     'note': 'whole step (all GEMMs + loss + optimizer) over wall time; flops = '
                             'the GEMMs EXECUTED: forward on all (1 + K) B rows (the user share of the first '
                             'combined layer once per sample), backward on the '
                             '2 B rows whose score gradient can differ from zero (the reference\'s '
                             'autograd runs the backward on all rows, 82 % of them exact zeros: '
                             'reference_autograd_flops_per_step)'},
    }
    if args.cpu_seconds > 0:
        from oracle import torch_path as T
        threads = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(threads)
        rng = np.random.default_rng(0)
        cu, cn = 100_000, 50_000          # small tables: the head's cost does not depend on them
        tabs = {'user_embeddings': rng.normal(size=(cu, D)).astype(np.float32),
                'item_embeddings': rng.normal(size=(cn, D)).astype(np.float32),
                'user_biases': np.zeros((cu, 1), np.float32),
                'item_biases': np.zeros((cn, 1), np.float32)}
        ref = T.TorchHybrid(tabs, item_metadata=rng.normal(size=(cn, MC)).astype(np.float32),
                            item_metadata_layers_dims=[128], freeze_embeddings=not args.unfrozen)
        ropt = T.make_optimizers(ref, 'adam', None, 1e-3, None)
        cb = 1024

        def cbatch():
            return ((torch.from_numpy(rng.integers(0, cu, cb).astype(np.int32)),
                     torch.from_numpy(rng.integers(0, cn, cb).astype(np.int32))),
                    torch.from_numpy(rng.integers(0, cn, (cb, K))))
        T.train_step(ref, ropt, cbatch(), T.t_adaptive_bpr)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.cpu_seconds:
            T.train_step(ref, ropt, cbatch(), T.t_adaptive_bpr)
            n += 1
        cdt = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': n * cb * (1 + K) / cdt, 'unit': 'interactions/s',
                               'cores': threads, 'kind': 'port',
                               'sample': f'{n} steps of B={cb}, tables {cu} x {cn} (the head does not '
                                         f'depend on the table sizes), {cdt:.1f} s'}
    del model, opt, batches, item_md
    torch.cuda.empty_cache()
    return out


def main():
    print(json.dumps(run()))


if __name__ == '__main__':
    main()
