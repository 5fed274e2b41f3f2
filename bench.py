"""
bench.py — headline benchmark of the MF training hot path (BASELINE.json: "train interactions/sec
(pos+negs)") on BASELINE configs[1]: synthetic ML-20M shape (138,493 users x 26,744 items,
20,000,263 interactions), MF dim=64, adaptive BPR + 10 negatives, SparseAdam (embeddings) + SGD
(biases), on MI355X.

A "step" is one pass of the hot path over one batch that is already resident in HBM:
``model.calculate_loss(batch)`` (fused gather/dot/loss/backward-staging kernel) ->
``loss.backward()`` -> ``optimizer.step()`` for both optimizers (radix coalesce + fused
segment-reduce/row-update kernels), all through the product API.  Negative sampling is done on the
device before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): BASELINE configs[3], the config
north_star's scaling target is stated on -- 100M users x 10M items x 128 with the four tables
row-sharded over the ranks (``collie_amd/sharded.py``: all-to-all row lookup, fused forward on the
compact tables, all-to-all of the gradient rows, owner-side update).  The GLOBAL batch (``--batch``
positives) is split evenly over the ranks: strong scaling; ``value`` counts all ranks' interactions.
The same tables and batch on ONE GPU are the ``c4`` object of the N = 1 line.

Further objects of the N = 1 line: ``c4`` (above: the step on tables no cache holds, with its own
K1 roofline and whole-step fraction), ``eval`` (the other half of BASELINE.json's metric, "eval
users/sec MAP@10", on configs[2]'s shape: 1M x 1M, dim 128, a sample of the users) and ``hybrid``
(configs[4], the MLP head).

Contract: ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_USERS, NUM_ITEMS, NUM_INTERACTIONS = 138_493, 26_744, 20_000_263
DIM, NUM_NEG = 64, 10
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
WORKLOAD = ('BASELINE configs[1]: synthetic ML-20M shape 138493 users x 26744 items, MF dim=64, '
            'adaptive BPR + 10 negatives, SparseAdam + SGD bias')


C4_WORKLOAD = ('BASELINE configs[3] tables: synthetic 100M users x 10M items, MF dim=128, adaptive '
               'BPR + 10 negatives, SparseAdam + SGD bias')
SHAPES = {'c2': (138_493, 26_744, 20_000_263, 64, WORKLOAD),
          'c4': (100_000_000, 10_000_000, 0, 128, C4_WORKLOAD)}


class _Shape:
    """What the model constructor reads from a train loader (sizes only: data is synthetic)."""
    num_users, num_items, num_negative_samples = NUM_USERS, NUM_ITEMS, NUM_NEG
    num_interactions = NUM_INTERACTIONS


def set_shape(name):
    global NUM_USERS, NUM_ITEMS, NUM_INTERACTIONS, DIM, WORKLOAD
    NUM_USERS, NUM_ITEMS, NUM_INTERACTIONS, DIM, WORKLOAD = SHAPES[name]
    _Shape.num_users, _Shape.num_items = NUM_USERS, NUM_ITEMS
    _Shape.num_interactions = NUM_INTERACTIONS


def build_model(device):
    from collie_amd.model import MatrixFactorizationModel
    torch.manual_seed(0)
    with torch.device(device):                 # tables are created (and initialised) in HBM
        model = MatrixFactorizationModel(train=_Shape(), embedding_dim=DIM, sparse=True, lr=1e-3,
                                         bias_lr=1e-2, optimizer='sparse_adam', bias_optimizer='sgd',
                                         loss='bpr', lr_scheduler_func=None)
    model.to(device)
    model.train()
    return model


def synthetic_batches(n, batch, device, dist, seed):
    """n batches in the loaders' format: users/pos int32 [B], negatives int64 [B, K]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = []
    for _ in range(n):
        if dist == 'zipf':
            # Zipf(1.05) popularity over ranks 1..rows: P(r) ~ r^-1.05, by inverting the CDF of the
            # continuous power law on [1, rows + 1) (id = rank - 1: hot ids are the low ones)
            def zipf(rows):
                a = 0.05                                        # s - 1
                u = torch.rand(batch, device=device, generator=g, dtype=torch.float64)
                r = torch.pow(1.0 - u * (1.0 - (rows + 1.0) ** -a), -1.0 / a)
                return (r.floor().clamp_(1, rows) - 1).to(torch.int32)
            users, pos = zipf(NUM_USERS), zipf(NUM_ITEMS)
        else:
            users = torch.randint(0, NUM_USERS, (batch,), device=device, generator=g, dtype=torch.int32)
            pos = torch.randint(0, NUM_ITEMS, (batch,), device=device, generator=g, dtype=torch.int32)
        neg = torch.randint(0, NUM_ITEMS, (batch, NUM_NEG), device=device, generator=g)
        out.append(((users, pos), neg))
    return out


def train_step(model, optimizers, batch):
    for opt in optimizers:
        opt.zero_grad()
    loss = model.calculate_loss(batch)
    loss.backward()
    for opt in optimizers:
        opt.step()
    return loss


def fwd_algorithmic_bytes(batch):
    """SURVEY.md §8d, forward part: index reads + (1 user + 1+K item) rows of D+1 floats."""
    idx = batch * (4 + 4) + batch * NUM_NEG * 8
    rows = (batch + (1 + NUM_NEG) * batch) * (DIM + 1) * 4
    return idx + rows


def step_algorithmic_bytes(batch_tensors):
    """
    SURVEY.md §8d whole-step figure for one batch: ``idx + fwd + opt`` with
    ``opt = (u* + i*) * D * 4 * 6`` (SparseAdam reads and writes p, m, v of every looked-up row,
    zero-gradient rows included) ``+ (u* + i*) * 4 * 2`` (bias SGD), u* / i* = unique user / item
    rows of the step (items: positives and all K negatives).
    """
    (users, pos), neg = batch_tensors
    B = users.shape[0]
    u_star = int(torch.unique(users).numel())
    i_star = int(torch.unique(torch.cat([pos.long(), neg.reshape(-1)])).numel())
    opt = (u_star + i_star) * DIM * 4 * 6 + (u_star + i_star) * 4 * 2
    return fwd_algorithmic_bytes(B) + opt, u_star, i_star


def pmc_traffic(shape, batch):
    """
    HBM-side bytes per launch of the forward kernel, from the committed rocprofv3 --pmc passes of
    this same command (profiles/rNN_pmc_traffic.json, written by tools/summarize_profiles.py:
    FETCH_SIZE and WRITE_SIZE collected in separate runs, KiB -> bytes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  A PMC pass cannot run inside the timed benchmark,
    so the number is read, not measured live; None when no pass was recorded for this workload.
    """
    for rnd in ('r03', 'r02', 'r01'):              # the newest committed passes of this command
        path = os.path.join(ROOT, 'profiles', f'{rnd}_pmc_traffic.json')
        try:
            with open(path) as f:
                per_kernel = json.load(f).get(f'{shape}_B{batch}', {})
        except (OSError, ValueError):
            continue
        for name, rec in per_kernel.items():
            if name.startswith('mf_train_fwd_lds'):
                return rec['hbm_bytes_per_launch']
    return None


def cpu_baseline(seconds=12.0, batch=1024):
    """The reference's torch op sequence (oracle/torch_path.py) on the host cores: a 'port'."""
    from oracle import torch_path
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = torch_path.TorchMF(NUM_USERS, NUM_ITEMS, DIM, sparse=True)
    opts = torch_path.make_optimizers(model, 'sparse_adam', 'sgd', 1e-3, 1e-2)
    rng = np.random.default_rng(0)

    def batch_():
        return ((torch.from_numpy(rng.integers(0, NUM_USERS, batch).astype(np.int32)),
                 torch.from_numpy(rng.integers(0, NUM_ITEMS, batch).astype(np.int32))),
                torch.from_numpy(rng.integers(0, NUM_ITEMS, (batch, NUM_NEG))))
    for _ in range(5):
        torch_path.train_step(model, opts, batch_(), torch_path.t_adaptive_bpr)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        torch_path.train_step(model, opts, batch_(), torch_path.t_adaptive_bpr)
        steps += 1
    dt = time.perf_counter() - t0
    return {'value': steps * batch * (1 + NUM_NEG) / dt, 'unit': 'interactions/s',
            'cores': threads, 'kind': 'port',
            'sample': f'{steps} steps of B={batch} (reference default batch size), same shape/'
                      f'loss/optimizers, torch CPU op sequence of the reference, {dt:.1f} s'}


EVAL_USERS, EVAL_ITEMS, EVAL_DIM, EVAL_POS = 1_000_000, 1_000_000, 128, 10
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA (exact fp32 products)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)


def small_batch_leg(device, batch=1024, steps=3000):
    """The reference's default batch size (trainer / loaders: 1,024) on configs[1]'s shape: the step
    through the product API launch by launch, and as CollieMinimalTrainer runs it (the whole loop
    body as one library call, collie_amd/captured.py)."""
    from collie_amd.captured import CapturedTrainStep
    set_shape('c2')
    out = {'workload': f'BASELINE configs[1] shape, B = {batch} (the reference\'s default batch size)',
           'steps': steps}
    for mode in ('stepped', 'one_call'):
        model = build_model(device)
        optimizers = model.configure_optimizers()
        batches = synthetic_batches(8, batch, device, 'uniform', seed=99)
        fused = CapturedTrainStep(model, optimizers, mode='call') if mode == 'one_call' else None

        def step(i):
            if fused is None or fused(batches[i % 8]) is None:
                train_step(model, optimizers, batches[i % 8])

        for i in range(300):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[mode] = {'us_per_step': dt * 1e6, 'interactions_per_s': batch * (1 + NUM_NEG) / dt}
        del model, optimizers
    return out


def eval_leg(device, n_users, sharded_world=1, group_rank=0):
    """evaluate_in_batches([mapk], k=10) on configs[2]'s shape for a sample of ``n_users`` users."""
    from collie_amd import metrics as M
    from collie_amd.interactions import Interactions
    rng = np.random.default_rng(0)
    users = np.repeat(np.sort(rng.choice(EVAL_USERS, n_users, replace=False)), EVAL_POS)
    items = rng.integers(0, EVAL_ITEMS, users.shape[0])
    test = Interactions(users=users, items=items, num_users=EVAL_USERS, num_items=EVAL_ITEMS,
                        allow_missing_ids=True, num_negative_samples=1, check_num_negative_samples_is_valid=False,
                        max_number_of_samples_to_consider=0, seed=0)
    if sharded_world == 1:
        from collie_amd.model import MatrixFactorizationModel

        class Shape:
            num_users, num_items, num_negative_samples = EVAL_USERS, EVAL_ITEMS, 1
        torch.manual_seed(0)
        model = MatrixFactorizationModel(train=Shape(), embedding_dim=EVAL_DIM, sparse=True,
                                         optimizer='sparse_adam', loss='bpr', lr_scheduler_func=None)
        with torch.no_grad():
            model.user_biases.weight.normal_(0, 0.01)
            model.item_biases.weight.normal_(0, 0.01)
        model.to(device)

        def run(metrics=(M.mapk,)):
            return M.evaluate_in_batches(list(metrics), test, model, k=10, verbose=False)
    else:
        from collie_amd.sharded import ShardedMatrixFactorizationModel
        model = ShardedMatrixFactorizationModel(EVAL_USERS, EVAL_ITEMS, 1, embedding_dim=EVAL_DIM,
                                                loss='bpr', device=device, seed=0)
        model.bias.normal_(0, 0.01)

        def run(metrics=(M.mapk,)):
            return model.evaluate_in_batches(list(metrics), test, k=10)
    run()                                                      # warm-up (workspace, first launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    score = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops = 2.0 * n_users * EVAL_ITEMS * EVAL_DIM
    # the same users with all three metrics (one more scoring pass for the MRR / AUC rank counts)
    t0 = time.perf_counter()
    all3 = run((M.mapk, M.mrr, M.auc))
    torch.cuda.synchronize()
    dt3 = time.perf_counter() - t0
    bf16 = None
    if sharded_world == 1:
        # the opt-in bf16 x 3 scoring pass (three-way bf16 split on v_mfma_f32_32x32x16_bf16, exact
        # re-scoring of 16 candidates per user): same users, same metric, reported beside the f32 pass
        model.eval_bf16x3 = True
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        score_b = run()
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
        model.eval_bf16x3 = False
        executed = 6.0 * flops          # six bf16 products per f32 product
        bf16 = {'value': n_users / dtb, 'unit': 'users/s', 'seconds': dtb,
                'map_at_10': float(score_b), 'equals_f32_pass': float(score_b) == float(score),
                'mfma': {'bound': 'mfma', 'achieved': executed / dtb / 1e12,
                         'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': executed / dtb / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                         'note': 'executed bf16 flops (6 products per f32 product) over end-to-end '
                                 'time against the dense bf16 MFMA peak'}}
    return {'metric': 'eval users/sec MAP@10', 'value': n_users / dt, 'unit': 'users/s',
            'workload': f'BASELINE configs[2] shape: {EVAL_USERS} users x {EVAL_ITEMS} items, dim '
                        f'{EVAL_DIM}, {EVAL_POS} test items per user, k=10; sample of {n_users} '
                        'users through evaluate_in_batches([mapk]) end to end (CSR slicing on the '
                        'host included)',
            'seconds': dt, 'map_at_10': float(score),
            'mapk_mrr_auc': {'value': n_users / dt3, 'unit': 'users/s', 'seconds': dt3,
                             'scores': [float(x) for x in all3]},
            'mapk_bf16x3_opt_in': bf16,
            'mfma': {'bound': 'mfma', 'achieved': flops / dt / 1e12, 'peak': MFMA_F32_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     'note': 'end-to-end time, not kernel time: a lower bound on the kernel rate'}}


def hybrid_leg(steps, warmup, batch=65536, cpu_seconds=0.0):
    """BASELINE configs[4]: the HybridPretrainedModel head (tools/hybrid_bench.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('hybrid_bench',
                                                  os.path.join(ROOT, 'tools', 'hybrid_bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = ['--steps', str(steps), '--warmup', str(warmup), '--batch', str(batch)]
    if cpu_seconds > 0:
        argv += ['--cpu-seconds', str(cpu_seconds)]
    return mod.run(argv)


def hybrid_get_preds_leg():
    """The hybrid head scoring a whole catalogue through get_preds (tools/hybrid_eval_bench.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'hybrid_eval_bench', os.path.join(ROOT, 'tools', 'hybrid_eval_bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.run([])


def main_sharded(args, world, rank, device):
    """N > 1: BASELINE configs[3] -- the 100M x 10M x 128 tables row-sharded over the ranks, one
    rank per GPU, RCCL all-to-all each way; the GLOBAL batch is fixed and split evenly over the
    ranks (strong scaling)."""
    import torch.distributed as dist
    from collie_amd.sharded import ShardedMatrixFactorizationModel
    set_shape(args.sharded_shape)
    backend = os.environ.get('COLLIE_BENCH_BACKEND', 'nccl')   # 'gloo': rehearsal on one GPU
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=device)
    else:
        dist.init_process_group(backend)
    local_batch = args.batch // world
    model = ShardedMatrixFactorizationModel(NUM_USERS, NUM_ITEMS, NUM_NEG, embedding_dim=DIM,
                                            lr=1e-3, bias_lr=1e-2, optimizer='sparse_adam',
                                            bias_optimizer='sgd', loss='bpr', device=device, seed=0)
    n_distinct = min(args.steps + args.warmup, 16)
    batches = synthetic_batches(n_distinct, local_batch, device, args.dist, seed=1234 + rank)
    ul = args.users_local
    if ul:
        # every rank trains the interactions of the users it owns (what a loader that partitions
        # the interactions by user % world hands out; same id distribution): user rows stay local
        last = (NUM_USERS - 1 - rank) // world * world + rank
        batches = [((torch.clamp(u // world * world + rank, max=last), p), n)
                   for (u, p), n in batches]
    pre_warm = max(0, 8 - args.warmup)     # as train_leg: the chip is brought to temperature first
    for i in range(pre_warm + args.warmup):
        model.train_step(batches[i % n_distinct], users_local=ul)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # the next step's exchanges are sized meanwhile: its plan is queued inside this step
        # (next_batch=) or, with --plan-side-stream, on a second stream after it (plan_ahead())
        nxt = (batches[(args.warmup + i + 1) % n_distinct]
               if args.plan_ahead and i + 1 < args.steps else None)
        loss = model.train_step(batches[(args.warmup + i) % n_distinct], users_local=ul,
                                next_batch=None if args.plan_side_stream else nxt)
        if nxt is not None and args.plan_side_stream:
            model.plan_ahead(nxt, users_local=ul)
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # a few more, untimed steps with an event at every phase boundary: where the sharded step spends
    # its time on this rank (plan / row exchange / compute / gradient exchange / owner update)
    model.profile_phases(True)
    n_ph = min(args.steps, 10)
    for i in range(n_ph):
        nxt = batches[(i + 1) % n_distinct] if args.plan_ahead and i + 1 < n_ph else None
        model.train_step(batches[i % n_distinct], users_local=ul,
                         next_batch=None if args.plan_side_stream else nxt)
        if nxt is not None and args.plan_side_stream:
            model.plan_ahead(nxt, users_local=ul)
    phases = model.phase_times()
    model.profile_phases(False)
    rccl = rccl_selftest(device, world, backend)
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        per_step = local_batch * world * (1 + NUM_NEG)
        result = {
            'metric': 'train interactions/sec (pos+negs)',
            'value': per_step * args.steps / elapsed, 'unit': 'interactions/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': WORKLOAD + f'; global batch {local_batch * world} positives split evenly '
                            f'over {world} GPUs (strong scaling)',
                'batch_per_gpu': local_batch, 'global_batch': local_batch * world,
                'id_distribution': args.dist,
                'parallelism': f'tables row-sharded over {world} GPUs (owner = row mod {world}), '
                               'RCCL all-to-all lookup + gradient exchange'
                               + ('; samples sit on the rank that owns their user (user rows are '
                                  'not exchanged)' if ul else ''),
                'bytes_per_rank_per_step': {k: model.last_exchange.get(k)
                                            for k in ('bytes_in', 'bytes_out')},
                'single_gpu_same_workload': "the 'c4' object of the N=1 line (python bench.py): "
                                            'same tables and global batch on one GPU, fused route',
                'single_gpu_same_workload_recorded': (recorded_c4() if args.sharded_shape == 'c4'
                                                      else None),
                'final_loss': float(loss.item()),
                'rows_exchanged_per_rank_per_step': model.last_exchange.get('rows_requested'),
                'gradient_rows_sent_per_rank_per_step': model.last_exchange.get('gradient_rows_sent'),
                'phase_ms_rank0': phases,
                'pre_warm_steps': pre_warm,
                'look_ahead': ('none' if not args.plan_ahead else
                               'plan_ahead() on a side stream' if args.plan_side_stream else
                               "train_step(next_batch=): the next step's plan is queued inside the "
                               'step, in front of the owner update'),
            },
            'roofline': None,
            'rccl': rccl,
        }
        fwd_ms = phases.get('forward')
        if fwd_ms:
            # rank 0's K1 launch on its compact table (the phase between the row exchange and the
            # gradient export: bias unpack + K1 + loss finalize), same algorithmic bytes per
            # positive as the single-GPU line
            alg = fwd_algorithmic_bytes(local_batch)
            achieved = alg / (fwd_ms * 1e-3) / 1e9
            result['roofline'] = {
                'kernel': 'mf_train_fwd_lds on rank 0 (sharded route: gathers from the compact row '
                          'tables the lookup exchange filled), timed as the forward phase',
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                'algorithmic_bytes_per_launch': alg, 'avg_launch_ms': fwd_ms}
        print(json.dumps(result))
    dist.destroy_process_group()


def train_leg(shape, args, device, steps, warmup):
    """One single-GPU training measurement on ``shape``: the step through the product API with HIP
    events around calculate_loss (K1 + loss finalize) on torch's current stream (the stream the
    library launches on)."""
    set_shape(shape)
    model = build_model(device)
    optimizers = model.configure_optimizers()
    n_distinct = min(steps + warmup, int(os.environ.get('COLLIE_BENCH_DISTINCT', 32)) if shape == 'c2' else 8)
    batches = synthetic_batches(n_distinct, args.batch, device, args.dist, seed=1234)
    # The chip needs ~30 ms of work after the idle table set-up before a step takes its steady time
    # (measured: 0.878 ms per step with 5 warm-up steps, 0.861 with 30, whichever batches are used:
    # clocks, not data).  The run is therefore brought to temperature by untimed steps BEFORE the W
    # warm-up steps the caller asked for; the K timed steps are whole steps like any other.
    pre_warm = max(0, min(40, int(50.0 / (7.0 if shape == 'c4' else 0.9))) - warmup)
    for i in range(pre_warm + warmup):
        train_step(model, optimizers, batches[i % n_distinct])
    fwd_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        batch = batches[(warmup + i) % n_distinct]
        for opt in optimizers:
            opt.zero_grad()
        fwd_events[i][0].record()
        loss = model.calculate_loss(batch)
        fwd_events[i][1].record()
        loss.backward()
        for opt in optimizers:
            opt.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model.check_status()
    final_loss = float(loss.item())
    step_bytes, u_star, i_star = step_algorithmic_bytes(batches[0])
    fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in fwd_events]))
    per_step = args.batch * (1 + NUM_NEG)
    alg = fwd_algorithmic_bytes(args.batch)
    achieved = alg / (fwd_ms * 1e-3) / 1e9
    lds_variant = {64: 'CH=4,LPR=4,NP=1', 128: 'CH=4,LPR=8,NP=2'}.get(DIM, '?')
    out = {
        'value': per_step * steps / elapsed, 'unit': 'interactions/s', 'steps': steps,
        'warmup': warmup, 'ms_per_step': elapsed / steps * 1e3,
        'config': {'workload': WORKLOAD, 'batch_per_gpu': args.batch, 'global_batch': args.batch,
                   'id_distribution': args.dist, 'parallelism': 'single GPU',
                   'final_loss': final_loss, 'pre_warm_steps': pre_warm},
        'roofline': {
            'kernel': f'mf_train_fwd_lds<{lds_variant},KIND=bpr,META=0> (+ loss_finalize_kernel, '
                      '5 us): calculate_loss = fused gather + dot + loss + backward staging',
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': pmc_traffic(shape, args.batch),
            'algorithmic_bytes_per_launch': alg, 'avg_launch_ms': fwd_ms,
            'residency': ('tables + SparseAdam moments = %.0f MB: L2 / Infinity-Cache resident, frac is '
                          'NOT a fraction of HBM traffic (traffic < algorithmic bytes); the HBM-'
                          'resident number is hbm_resident' % ((NUM_USERS + NUM_ITEMS) * (DIM + 1) * 12 / 1e6)
                          if shape == 'c2' else 'tables exceed every cache: HBM-resident'),
            'whole_step': {
                'note': 'all kernels of the step (forward + coalesce + optimizer row updates): '
                        'SURVEY.md 8d bytes_step = idx + fwd + opt over wall time per step',
                'algorithmic_bytes_per_step': step_bytes, 'unique_user_rows': u_star,
                'unique_item_rows': i_star,
                'achieved': step_bytes / (elapsed / steps) / 1e9, 'unit': 'GB/s',
                'frac': step_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS,
            },
        },
    }
    del model, optimizers, batches
    torch.cuda.empty_cache()
    return out


def launch_ranks(n):
    """``python bench.py --gpus N`` run plainly: start the N ranks as a CHILD process
    (torch.distributed.run, rendezvous on 127.0.0.1), relay rank 0's JSON line and exit with the
    child's code.  This parent never touches the GPU (no torch.cuda call before or after)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)]
    cmd += sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    for ln in child.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    raise SystemExit(child.returncode if child.returncode or lines else 1)


def recorded_c4():
    """The single-GPU number of the N > 1 workload as the last committed N = 1 line has it
    (profiles/rNN_bench_default.json, 'c4' object): what `value` of an N > 1 line is to be compared
    with -- the N = 1 line's own `value` is a different workload (configs[1])."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_default.json')), reverse=True):
        try:
            with open(path) as f:
                c4 = json.loads(f.read().strip().splitlines()[-1]).get('c4') or {}
            if 'value' in c4:
                return {'value': c4['value'], 'unit': c4.get('unit'), 'ms_per_step': c4.get('ms_per_step'),
                        'source': os.path.relpath(path, ROOT)}
        except (OSError, ValueError):
            continue
    return None


def rccl_selftest(device, world, backend):
    """What the collectives of this run cost, measured on the process group the step uses: the
    latency of a small all-to-all (the count exchanges) and the rate of a large one (the row
    exchanges), so the N > 1 line shows that the backend saw `world` ranks and what it delivered."""
    import torch.distributed as dist
    staged = backend == 'gloo'
    dev = 'cpu' if staged else device

    def a2a(n_floats, iters):
        src = torch.ones(world * n_floats, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        for _ in range(3):
            dist.all_to_all_single(dst, src)
        if not staged:
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            dist.all_to_all_single(dst, src)
        if not staged:
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    small = a2a(8, 50)
    big_floats = (32 << 20) // 4                      # 32 MiB to every peer
    big = a2a(big_floats, 5)
    return {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
            'all_to_all_us_small': small * 1e6,
            'all_to_all_GBps_large': (world - 1) * big_floats * 4 / big / 1e9,
            'note': 'per rank: 64-byte messages (latency) / 32 MiB per peer, bytes sent to the '
                    'other ranks per second'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=1048576,
                    help='positives per step (SURVEY.md 8d lists 1024, 65536, 1048576); with '
                         '--gpus N > 1 the GLOBAL batch, split evenly over the ranks')
    ap.add_argument('--dist', choices=['uniform', 'zipf'], default='uniform')
    ap.add_argument('--shape', choices=['c2', 'c4', 'c5'], default='c2',
                    help="c2 = BASELINE configs[1] (the headline line); c4 = configs[3]'s tables "
                         '(100M users x 10M items, dim 128) on ONE GPU: the shape no cache holds; '
                         'c5 = configs[4], the hybrid MLP head (tools/hybrid_bench.py)')
    ap.add_argument('--sharded-shape', choices=['c2', 'c4'], default='c4',
                    help="tables of the N > 1 run (default: configs[3], north_star's scaling config)")
    ap.add_argument('--no-plan-ahead', dest='plan_ahead', action='store_false',
                    help='N > 1: do NOT plan step t + 1 while step t computes (train_step(next_batch=) '
                         'queues the next plan in front of the owner update, so the host reads of a '
                         'step never idle the GPU; on by default, see DESIGN.md 6)')
    ap.add_argument('--plan-side-stream', action='store_true',
                    help='N > 1: plan ahead with plan_ahead() on a second stream instead')
    ap.add_argument('--no-users-local', dest='users_local', action='store_false',
                    help='N > 1: draw every rank\'s users from the whole table (user rows then take '
                         'the exchange like item rows) instead of from the users the rank owns')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--zipf-steps', type=int, default=20,
                    help='steps of the Zipf(1.05)-ids leg of the default line (0 = skip it)')
    ap.add_argument('--eval-users', type=int, default=65536,
                    help='users in the eval leg (0 = skip it)')
    ap.add_argument('--hybrid-steps', type=int, default=20,
                    help="steps of the configs[4] leg (hybrid MLP head, reported under 'hybrid'; "
                         '0 = skip it; single-GPU runs only)')
    ap.add_argument('--small-batch-steps', type=int, default=3000,
                    help="steps of the B = 1,024 leg reported under 'b1024' (0 = skip it)")
    ap.add_argument('--c4-steps', type=int, default=10,
                    help="steps of the configs[3]-tables-on-one-GPU leg reported under 'c4' (the "
                         'HBM-resident shape; 0 = skip it; default shape only)')
    args = ap.parse_args()

    if args.shape == 'c5':
        print(json.dumps(hybrid_leg(args.steps, args.warmup,
                                    args.batch if args.batch != 1048576 else 65536,
                                    0.0 if args.no_cpu_baseline else 12.0)))
        return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        return launch_ranks(args.gpus)
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1 or os.environ.get('COLLIE_BENCH_SHARDED_WORLD1') == '1':
        # (COLLIE_BENCH_SHARDED_WORLD1=1 under torch.distributed.run --nproc-per-node 1: the whole
        # N > 1 code path on one rank, every collective issued to RCCL -- a rehearsal of the line the
        # driver's multi-GPU run prints, for boxes with one GPU)
        return main_sharded(args, world, rank, device)

    leg = train_leg(args.shape, args, device, args.steps, args.warmup)
    result = {
        'metric': 'train interactions/sec (pos+negs)', 'value': leg['value'],
        'unit': 'interactions/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': leg['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'config': leg['config'],
        'roofline': leg['roofline'],
    }
    if not args.no_cpu_baseline:
        set_shape(args.shape)
        result['cpu_baseline'] = cpu_baseline()
    if args.c4_steps > 0 and args.shape == 'c2':
        free, _ = torch.cuda.mem_get_info(device)
        if free >= 190 * 2 ** 30:
            # the same step on tables no cache holds (169 GB with the SparseAdam moments): the
            # headline's 42 MB tables live in L2 / the Infinity Cache, these in HBM
            c4 = train_leg('c4', args, device, args.c4_steps, 3)
            result['c4'] = {k: c4[k] for k in ('value', 'unit', 'steps', 'warmup', 'ms_per_step',
                                               'config', 'roofline')}
            # the headline tables are cache-resident: the same kernel's fraction on tables that are
            # not, beside it in the top-level roofline object
            result['roofline']['hbm_resident'] = {
                k: c4['roofline'][k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac',
                                               'traffic', 'algorithmic_bytes_per_launch',
                                               'avg_launch_ms')}
            result['roofline']['hbm_resident']['workload'] = c4['config']['workload']
        else:
            result['c4'] = {'skipped': f'{free / 2 ** 30:.0f} GiB of HBM free, the leg needs 190'}
    if args.zipf_steps > 0 and args.shape == 'c2' and args.dist == 'uniform':
        # "ids drawn uniform and Zipf(1.05) (report both)" -- the same step on skewed
        # ids (hot rows own long segments: the piecewise segment sums of K3a / K4)
        args.dist = 'zipf'
        z = train_leg('c2', args, device, args.zipf_steps, 5)
        args.dist = 'uniform'
        result['zipf'] = {'id_distribution': 'Zipf(1.05)', 'value': z['value'], 'unit': z['unit'],
                          'ms_per_step': z['ms_per_step'], 'steps': args.zipf_steps,
                          'forward_avg_launch_ms': z['roofline']['avg_launch_ms'],
                          'forward_frac_of_8TBs': z['roofline']['frac']}
    if args.small_batch_steps > 0 and args.shape == 'c2':
        result['b1024'] = small_batch_leg(device, 1024, args.small_batch_steps)
    if args.eval_users > 0:
        result['eval'] = eval_leg(device, args.eval_users)
    if args.hybrid_steps > 0 and args.shape == 'c2':
        torch.cuda.empty_cache()
        leg = hybrid_leg(args.hybrid_steps, 8)  # (3 warm-up steps left the first timed ones 5 % slow)
        result['hybrid'] = {k: leg[k] for k in ('metric', 'value', 'unit', 'ms_per_step',
                                                'config', 'roofline')}
        del leg
        torch.cuda.empty_cache()
        result['hybrid']['get_preds'] = hybrid_get_preds_leg()
    print(json.dumps(result))


if __name__ == '__main__':
    main()
