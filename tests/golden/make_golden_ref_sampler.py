"""
Generates ``tests/golden/ref_sampler.npz`` from the REFERENCE'S OWN ``Interactions.__getitem__`` /
``_negative_sample`` (``collie/interactions/datasets.py:378-445``), imported verbatim through
``tests/golden/ref_harness.py``.  Build container only.  Usage:
``python tests/golden/make_golden_ref_sampler.py``.

The reference draws from host generators (``random.choice`` / ``random.choices`` in the exact
rejection loop, ``np.random.randint`` in approximate mode).  Only those three calls are redirected
-- in the namespace of the reference's ``datasets`` module -- to the counter-based Philox stream the
device sampler uses (``draw_item(seed, position, draw, num_items)``, oracle/sampler_oracle.py, itself
pinned to the Random123 vectors): every decision of the sampling RULES (what is rejected, how the
``samples_checked`` budget is counted, what the fill after an exhausted budget looks like, the
order of the negatives) is then made by the reference's code, and the outputs pin both
``oracle/sampler_oracle.py`` and ``collie_sample_batch`` to it bit for bit.
"""
import os
import sys

import numpy as real_np
from scipy.sparse import coo_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness  # noqa: E402

ref_harness.install()
import collie.interactions.datasets as D            # noqa: E402  (reference)
from oracle import sampler_oracle as S              # noqa: E402  (only draw_item: the stream)


class Stream:
    """The draws of ONE sample: (seed, position) fixed, the draw number counts up."""
    seed = position = draw = 0
    rows = None                     # batched approximate mode: positions of the rows of one call

    @classmethod
    def start(cls, seed, position):
        cls.seed, cls.position, cls.draw, cls.rows = seed, position, 0, None

    @classmethod
    def next(cls, n):
        v = S.draw_item(cls.seed, cls.position, cls.draw, n)
        cls.draw += 1
        return v


class FakeRandom:
    """Stands where the reference's module-level name ``random`` stood (datasets.py:405-421)."""
    @staticmethod
    def seed(_):
        pass

    @staticmethod
    def choice(seq):
        return seq[Stream.next(len(seq))]

    @staticmethod
    def choices(seq, k):
        return [seq[Stream.next(len(seq))] for _ in range(k)]


class FakeNumpyRandom:
    @staticmethod
    def randint(low, high, size):
        if len(size) == 1:
            return real_np.array([low + Stream.next(high - low) for _ in range(size[0])])
        out = real_np.empty(size, real_np.int64)
        positions = Stream.rows
        for t in range(size[0]):                    # one sample per row, each its own stream
            Stream.start(Stream.seed, positions[t])
            out[t] = [low + Stream.next(high - low) for _ in range(size[1])]
        return out


class NumpyProxy:
    """``np`` of the reference's module: numpy, except ``np.random.randint`` (datasets.py:439)."""
    random = FakeNumpyRandom()

    def __getattr__(self, name):
        return getattr(real_np, name)


def dataset(U, N, nnz, seed, dense_users=0):
    rng = real_np.random.default_rng(seed)
    keys = rng.choice(U * N, nnz, replace=False)
    rows, cols = keys // N, keys % N
    for u in range(dense_users):                    # users that interacted with all but 1 + u items
        have = set(cols[rows == u].tolist())
        extra = [i for i in range(N) if i not in have][: max(0, N - len(have) - 1 - u)]
        rows = real_np.concatenate([rows, real_np.full(len(extra), u)])
        cols = real_np.concatenate([cols, real_np.array(extra, dtype=cols.dtype)])
    return rows.astype(real_np.int32), cols.astype(real_np.int32)


def case(out, tag, U, N, nnz, K, max_tries, seed, base, data_seed, dense_users=0, shuffle=False,
         batched=False):
    rows, cols = dataset(U, N, nnz, data_seed, dense_users)
    mat = coo_matrix((real_np.ones(len(rows)), (rows, cols)), shape=(U, N))
    D.random, D.np = SAVED                          # the constructor runs on the real modules
    inter = D.Interactions(mat=mat, num_negative_samples=K, allow_missing_ids=True,
                           remove_duplicate_user_item_pairs=False,
                           check_num_negative_samples_is_valid=False,
                           max_number_of_samples_to_consider=max_tries, seed=1)
    D.random, D.np = FakeRandom(), NumpyProxy()
    n = len(rows)
    order = real_np.random.default_rng(data_seed + 1).permutation(n) if shuffle else real_np.arange(n)
    users = real_np.empty(n, real_np.int32)
    pos = real_np.empty(n, real_np.int32)
    neg = real_np.empty((n, K), real_np.int64)
    if batched:                                     # one __getitem__ call with an index array
        assert max_tries == 0
        Stream.start(seed, base)
        Stream.rows = [base + t for t in range(n)]
        (u, p), ng = inter[order]
        users[:], pos[:], neg[:] = u, p, ng
    else:
        for t in range(n):                          # sample t = dataset entry order[t]
            Stream.start(seed, base + t)
            (u, p), ng = inter[int(order[t])]
            users[t], pos[t], neg[t] = u, p, ng
    D.random, D.np = SAVED
    out[f'{tag}|rows'], out[f'{tag}|cols'] = rows, cols
    out[f'{tag}|order'] = order.astype(real_np.int64)
    out[f'{tag}|params'] = real_np.array([U, N, K, max_tries, seed, base, int(shuffle)],
                                         dtype=real_np.int64)
    out[f'{tag}|users'], out[f'{tag}|pos'], out[f'{tag}|neg'] = users, pos, neg
    print(tag, 'samples', n, 'K', K, 'max_tries', max_tries)


SAVED = (D.random, D.np)


def main():
    out = {}
    big = (1 << 40) + 12345
    # exact mode, budget never hit / K = 1, 4, 10
    case(out, 'exact_k1', 50, 40, 600, 1, 200, 7, 0, 1)
    case(out, 'exact_k4', 50, 40, 600, 4, 200, big, 1000, 2, shuffle=True)
    case(out, 'exact_k10', 60, 200, 3000, 10, 200, 99, 5, 3)
    # tiny catalogue, dense users, small budget: the budget is hit, fills hold positives / repeats
    case(out, 'exact_budget', 20, 12, 120, 4, 3, big + 1, 0, 4, dense_users=6, shuffle=True)
    case(out, 'exact_budget_k10', 20, 14, 150, 10, 5, 3, 77, 5, dense_users=4)
    case(out, 'exact_budget1', 30, 10, 200, 3, 1, 11, 0, 6, dense_users=5)
    # approximate mode (max_number_of_samples_to_consider = 0): scalar and batched __getitem__
    case(out, 'approx_k10', 50, 40, 600, 10, 0, 5, 0, 7)
    case(out, 'approx_batched', 50, 40, 600, 7, 0, big, 123456789, 8, shuffle=True, batched=True)
    real_np.savez_compressed(os.path.join(HERE, 'ref_sampler.npz'), **out)


if __name__ == '__main__':
    main()
