"""
GPU test (``-m gpu``) of the allocation-free entry point ``collie_mf_eval_topk_static``: called
through the C ABI, its top-k ids and scores must equal ``collie_mf_eval_topk``'s bit for bit, on
two calls in a row that share the library's static scratch (the second with a smaller input, so
stale candidates of the first call would show), and it must reject sizes beyond its maximum.
Collected last (file name) so that a failure here cannot hide other tests under ``pytest -x``.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tables(dev, U, N, D, seed):
    from collie_amd import _lib
    g = torch.Generator(device='cpu').manual_seed(seed)
    ue = torch.randn(U, D, generator=g).to(dev)
    ie = torch.randn(N, D, generator=g).to(dev)
    ub = (0.1 * torch.randn(U, 1, generator=g)).to(dev)
    ib = (0.1 * torch.randn(N, 1, generator=g)).to(dev)
    # engineered exact ties: duplicate item rows (ties must resolve by ascending id in both paths)
    ie[N // 2] = ie[3]
    ib[N // 2] = ib[3]
    return _lib.make_tables(ue, ie, ub, ib), (ue, ie, ub, ib)


@pytest.mark.parametrize('D', [64, 20])         # LDS-DMA kernel / register-staged kernel
def test_static_topk_equals_workspace_topk_two_calls_in_a_row(D):
    from collie_amd import ops
    dev = torch.device('cuda:0')
    U, N = 3000, 40000
    tables, keep = _tables(dev, U, N, D, seed=D)
    rng = np.random.default_rng(0)
    # first call: the maximum batch and the whole catalogue; second: fewer users, fewer items, other k
    for n_users, (lo, hi), k, y in ((1024, (0, N), 10, None), (37, (100, 9000), 16, (1.0, 5.0))):
        uids = torch.tensor(rng.integers(0, U, n_users), dtype=torch.int64, device=dev)
        ws = ops.EvalWorkspace(dev)
        ref_s, ref_i = ops.eval_topk(tables, uids, lo, hi, k, y, ws)
        out_s = torch.full((n_users, k), float('nan'), dtype=torch.float32, device=dev)
        out_i = torch.full((n_users, k), -7, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(dev)
        ops.eval_topk_static(tables, uids, lo, hi, k, y, out_s, out_i)
        assert torch.cuda.memory_allocated(dev) == before     # torch's allocator was not touched
        torch.cuda.synchronize()
        assert torch.equal(out_i, ref_i)
        assert torch.equal(out_s, ref_s)
        assert int(out_i.min()) >= lo and int(out_i.max()) < hi
    del keep


def test_static_topk_rejects_sizes_beyond_its_maximum():
    from collie_amd import _lib, ops
    dev = torch.device('cuda:0')
    tables, keep = _tables(dev, 2000, 500, 32, seed=1)
    uids = torch.zeros(1025, dtype=torch.int64, device=dev)
    out_s = torch.empty(1025, 5, dtype=torch.float32, device=dev)
    out_i = torch.empty(1025, 5, dtype=torch.int64, device=dev)
    with pytest.raises(_lib.CollieHipError, match='at most 1024 users'):
        ops.eval_topk_static(tables, uids, 0, 500, 5, None, out_s, out_i)
    del keep
