"""
TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by collie_amd/).

CPU restatement of the device-side batch assembly + negative sampler (``collie_sample_batch``).
The ALGORITHM is the reference's ``Interactions.__getitem__`` / ``_negative_sample``
(collie/interactions/datasets.py:375-445): approximate mode draws K uniform item ids per sample
(``np.random.randint``, :439-443); exact mode draws candidates one at a time, rejects a candidate
that the user has interacted with or that the sample already holds (:407-410), counts rejections
against ``max_number_of_samples_to_consider`` per sample (:411, :423) and, once that budget is
spent, fills the remaining slots with unconditional draws and keeps the current candidate as the
last negative (:412-419, :425).

The random STREAM is necessarily different from the reference's (CPython ``random`` / numpy's
global generator are host generators): draws come from the counter-based Philox4x32-10 generator
(Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), counter = (global sample
position, draw number), key = seed, so a batch does not depend on how the work is split.

Pinned: (a) the generator against the Random123 known-answer vectors; (b) the sampling RULES against
the reference's own ``Interactions.__getitem__`` / ``_negative_sample`` run in the build container
with its three RNG calls (``random.choice``, ``random.choices``, ``np.random.randint``) reading this
stream -- ``tests/golden/ref_sampler.npz``, written by ``tests/golden/make_golden_ref_sampler.py``:
exact mode with and without an exhausted budget, K = 1 / 4 / 10, tiny catalogues, approximate mode
scalar and batched -- bit for bit (tests/test_sampler_cpu.py); (c) the device kernel against that
fixture and against this file bit for bit (tests/test_gpu_sampler.py); (d) the invariants the
reference's own tests assert (tests/test_interactions.py:356-457).
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """ctr: 4 uint32, key: 2 uint32 (python ints) -> 4 uint32."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK32, p1 & MASK32, \
                         ((p0 >> 32) ^ c3 ^ k1) & MASK32, p0 & MASK32
        k0 = (k0 + W0) & MASK32
        k1 = (k1 + W1) & MASK32
    return c0, c1, c2, c3


def draw_item(seed, position, draw, num_items):
    """Uniform item id in [0, num_items): high 64 bits of (64 random bits) x num_items."""
    r = philox4x32_10((position & MASK32, (position >> 32) & MASK32, draw & MASK32, 0),
                      (seed & MASK32, (seed >> 32) & MASK32))
    r64 = r[0] | (r[1] << 32)
    return (r64 * num_items) >> 64


def sample_batch(rows, cols, order, start, batch, indptr, indices, num_items, num_neg, max_tries,
                 seed, position_base):
    """
    -> users int32 [batch], pos int32 [batch], neg int64 [batch, num_neg]: the loaders' batch
    format (datasets.py:382-388).  ``order`` None = dataset order; ``indptr`` None = approximate.
    Sample t of the batch is dataset entry order[start + t]; its stream position is
    position_base + start + t.
    """
    users = np.empty(batch, np.int32)
    pos = np.empty(batch, np.int32)
    neg = np.empty((batch, num_neg), np.int64)
    for t in range(batch):
        idx = start + t if order is None else int(order[start + t])
        u = int(rows[idx])
        users[t], pos[t] = u, cols[idx]
        position = position_base + start + t
        d = 0
        if indptr is None or max_tries <= 0:
            for k in range(num_neg):
                neg[t, k] = draw_item(seed, position, d, num_items)
                d += 1
            continue
        mine = set(int(x) for x in indices[indptr[u]:indptr[u + 1]])
        got = []
        checked = 0
        while len(got) < num_neg:
            cand = draw_item(seed, position, d, num_items)
            d += 1
            while cand in mine or cand in got:
                if checked >= max_tries:
                    for _ in range(num_neg - len(got) - 1):
                        got.append(draw_item(seed, position, d, num_items))
                        d += 1
                    break
                cand = draw_item(seed, position, d, num_items)
                d += 1
                checked += 1
            got.append(cand)
        neg[t] = got
    return users, pos, neg
