"""
CPU tests of the sampler oracle (oracle/sampler_oracle.py): the Philox4x32-10 generator against
the Random123 known-answer vectors, and the sampling rules against the invariants the reference's
own tests assert (tests/test_interactions.py:356-457 of the reference).
"""
import numpy as np
import scipy.sparse as sp

from oracle import sampler_oracle as S


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    assert S.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    f = 0xffffffff
    assert S.philox4x32_10((f, f, f, f), (f, f)) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert S.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
                           (0xa4093822, 0x299f31d0)) == (0xd16cfe09, 0x94fdcceb, 0x5001e420,
                                                         0x24126ea1)


def make_data(U=40, N=60, nnz=500, seed=0):
    rng = np.random.default_rng(seed)
    keys = rng.choice(U * N, nnz, replace=False)
    rows, cols = (keys // N).astype(np.int32), (keys % N).astype(np.int32)
    csr = sp.coo_matrix((np.ones(nnz), (rows, cols)), shape=(U, N)).tocsr()
    csr.sort_indices()
    return rows, cols, csr


def test_exact_negatives_are_true_negatives_and_distinct():
    rows, cols, csr = make_data()
    N, K = 60, 5
    users, pos, neg = S.sample_batch(rows, cols, None, 0, len(rows), csr.indptr, csr.indices, N, K,
                                     200, seed=42, position_base=0)
    np.testing.assert_array_equal(users, rows)
    np.testing.assert_array_equal(pos, cols)
    assert neg.shape == (len(rows), K) and neg.dtype == np.int64
    assert users.dtype == np.int32 and pos.dtype == np.int32
    for t in range(len(rows)):
        mine = set(csr.indices[csr.indptr[users[t]]:csr.indptr[users[t] + 1]])
        assert not (set(neg[t]) & mine)                 # test_interactions.py:421-424, :443-447
        assert len(set(neg[t])) == K                    # datasets.py:409
    assert len(set(neg.ravel())) > K                    # test_interactions.py:428


def test_approximate_is_uniform_with_replacement():
    rows, cols, _ = make_data()
    _, _, neg = S.sample_batch(rows, cols, None, 0, len(rows), None, None, 60, 8, 0, 1, 0)
    assert neg.min() >= 0 and neg.max() < 60
    counts = np.bincount(neg.ravel(), minlength=60)
    expect = neg.size / 60
    assert ((counts - expect) ** 2 / expect).sum() < 120      # chi^2, 59 dof (p ~ 1e-5)


def test_budget_exhaustion_follows_the_reference_fill_rule():
    # one user who has interacted with every item but one: rejections run out at once
    N, K = 12, 4
    rows = np.zeros(N - 1, np.int32)
    cols = np.arange(N - 1, dtype=np.int32)
    indptr, indices = np.array([0, N - 1]), cols.copy()
    _, _, neg = S.sample_batch(rows, cols, None, 0, 6, indptr, indices, N, K, 1, 3, 0)
    assert neg.shape == (6, K) and neg.min() >= 0 and neg.max() < N
    # with a budget of one rejection most samples must contain positives (fill is unconditional)
    assert np.isin(neg, cols).any()


def test_stream_does_not_depend_on_batch_split_or_order_argument():
    rows, cols, csr = make_data()
    N, K = 60, 3
    whole = S.sample_batch(rows, cols, None, 0, 50, csr.indptr, csr.indices, N, K, 200, 9, 1000)
    a = S.sample_batch(rows, cols, None, 0, 20, csr.indptr, csr.indices, N, K, 200, 9, 1000)
    b = S.sample_batch(rows, cols, None, 20, 30, csr.indptr, csr.indices, N, K, 200, 9, 1000)
    np.testing.assert_array_equal(whole[2], np.concatenate([a[2], b[2]]))
    ident = np.arange(len(rows), dtype=np.int64)
    again = S.sample_batch(rows, cols, ident, 0, 50, csr.indptr, csr.indices, N, K, 200, 9, 1000)
    np.testing.assert_array_equal(whole[2], again[2])
    other_epoch = S.sample_batch(rows, cols, None, 0, 50, csr.indptr, csr.indices, N, K, 200, 9,
                                 1000 + len(rows))
    assert not np.array_equal(whole[2], other_epoch[2])


def test_host_loader_rejects_positives_and_within_row_repeats():
    """``Interactions._negative_sample`` follows ``datasets.py:399-426``: a candidate is rejected if
    it is one of the user's positives OR already in the sample's negative list (the device sampler
    follows the same rule, tests/test_gpu_sampler.py)."""
    from collie_amd.interactions import Interactions
    rng = np.random.default_rng(0)
    it = Interactions(users=rng.integers(0, 20, 200), items=rng.integers(0, 40, 200), num_users=20,
                      num_items=40, allow_missing_ids=True, num_negative_samples=8,
                      max_number_of_samples_to_consider=200, seed=1)
    (u, _), neg = it[np.arange(len(it))]
    assert neg.shape == (len(it), 8)
    pos = set(zip(it.mat.row.tolist(), it.mat.col.tolist()))
    assert all(len(set(r.tolist())) == len(r) for r in neg)
    assert not any((uu, n) in pos for uu, r in zip(u.tolist(), neg.tolist()) for n in r)


# ---- the sampling RULES pinned to the reference's own _negative_sample -----------------------------
GOLDEN = __import__('os').path.join(__import__('os').path.dirname(__file__), 'golden',
                                    'ref_sampler.npz')


def ref_sampler_cases():
    """tests/golden/ref_sampler.npz: outputs of the reference's ``Interactions.__getitem__``
    (datasets.py:378-445) with its three RNG calls reading the Philox stream
    (tests/golden/make_golden_ref_sampler.py)."""
    z = np.load(GOLDEN)
    for tag in sorted({k.split('|')[0] for k in z.files}):
        U, N, K, max_tries, seed, base, shuffled = (int(v) for v in z[f'{tag}|params'])
        rows, cols = z[f'{tag}|rows'], z[f'{tag}|cols']
        csr = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(U, N)).tocsr()
        csr.sort_indices()
        yield dict(tag=tag, U=U, N=N, K=K, max_tries=max_tries, seed=seed, base=base,
                   rows=rows, cols=cols, order=z[f'{tag}|order'] if shuffled else None, csr=csr,
                   want=(z[f'{tag}|users'], z[f'{tag}|pos'], z[f'{tag}|neg']))


def test_oracle_equals_the_references_negative_sample_bit_for_bit():
    seen_budget_hit = False
    for c in ref_sampler_cases():
        exact = c['max_tries'] > 0
        got = S.sample_batch(c['rows'], c['cols'], c['order'], 0, len(c['rows']),
                             c['csr'].indptr if exact else None,
                             c['csr'].indices if exact else None, c['N'], c['K'], c['max_tries'],
                             seed=c['seed'], position_base=c['base'])
        for g, w, name in zip(got, c['want'], ('users', 'pos', 'neg')):
            np.testing.assert_array_equal(g, w, err_msg=f"{c['tag']} {name}")
            assert g.dtype == w.dtype, (c['tag'], name)
        if exact:       # does the fixture exercise the exhausted-budget fill (datasets.py:412-419)?
            neg = c['want'][2]
            for t in range(len(neg)):
                mine = set(c['csr'].indices[c['csr'].indptr[c['want'][0][t]]:
                                            c['csr'].indptr[c['want'][0][t] + 1]])
                if len(set(neg[t])) < c['K'] or set(neg[t]) & mine:
                    seen_budget_hit = True
    assert seen_budget_hit
