"""CPU: the oracle against the committed golden vectors and against independent float64 numpy."""
import json
import os

import numpy as np
import pytest

from oracle import flat as oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flat_search.json")


def _cases():
    with open(GOLDEN) as fh:
        return json.load(fh)


@pytest.mark.parametrize("name", sorted(_cases().keys()))
def test_oracle_reproduces_golden(name):
    c = _cases()[name]
    X = oracle.synth_rows(c["corpus_seed"], 0, c["n"], c["d"])
    Q = oracle.synth_rows(c["query_seed"], 0, c["nq"], c["d"])
    D, I = oracle.search(X, Q, c["k"], c["metric"])
    np.testing.assert_array_equal(I, np.array(c["ids"], dtype=np.int64))
    np.testing.assert_array_equal(D.view(np.uint32), np.array(c["scores_bits"], dtype=np.uint32))


def test_oracle_is_thread_count_invariant():
    X = oracle.synth_rows(3, 0, 30_000, 128)
    Q = oracle.synth_rows(4, 0, 32, 128)
    D1, I1 = oracle.search(X, Q, 10, nthreads=1)
    D8, I8 = oracle.search(X, Q, 10, nthreads=5)
    np.testing.assert_array_equal(I1, I8)
    np.testing.assert_array_equal(D1, D8)


def test_oracle_scores_within_1e4_of_float64_and_ids_match_outside_near_ties():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((20_000, 384), dtype=np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((32, 384), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    D, I = oracle.search(X, Q, 10)
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    assert np.abs(D - np.take_along_axis(S, I, 1)).max() < 1e-4  # north-star score tolerance
    truth = np.argsort(-S, axis=1, kind="stable")[:, :11]
    for b in range(32):
        gaps = -np.diff(S[b, truth[b]])
        if gaps.min() > 1e-6:  # fp32 summation noise is ~1e-7 on unit vectors
            np.testing.assert_array_equal(I[b], truth[b, :10])


def test_oracle_l2_matches_float64():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((3000, 64), dtype=np.float32)
    Q = rng.standard_normal((5, 64), dtype=np.float32)
    D, I = oracle.search(X, Q, 7, oracle.METRIC_L2)
    L = ((Q[:, None, :].astype(np.float64) - X[None].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_array_equal(I, np.argsort(L, axis=1, kind="stable")[:, :7])
    np.testing.assert_allclose(D, np.take_along_axis(L, I, 1), rtol=1e-5)
    assert (np.diff(D, axis=1) >= 0).all()


def test_oracle_tie_rule_and_padding():
    base = oracle.synth_rows(8, 0, 40, 32)
    X = np.concatenate([base, base])
    Q = oracle.synth_rows(9, 0, 4, 32)
    D, I = oracle.search(X, Q, 6)
    # duplicates: each score appears twice, smaller id first
    assert (I[:, 0::2] + 40 == I[:, 1::2]).all()
    assert (D[:, 0::2] == D[:, 1::2]).all()
    D, I = oracle.search(X[:3], Q, 5)
    assert (I[:, 3:] == -1).all() and (D[:, 3:] == -np.finfo(np.float32).max).all()
    D, I = oracle.search(X[:3], Q, 5, oracle.METRIC_L2)
    assert (I[:, 3:] == -1).all() and (D[:, 3:] == np.finfo(np.float32).max).all()


def test_oracle_merge_equals_search_on_union():
    X = oracle.synth_rows(21, 0, 9000, 96)
    Q = oracle.synth_rows(22, 0, 16, 96)
    for metric in (0, 1):
        parts = [oracle.search(X[lo:lo + 3000], Q, 10, metric, id_offset=lo) for lo in (0, 3000, 6000)]
        Dm, Im = oracle.merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), metric)
        D, I = oracle.search(X, Q, 10, metric)
        np.testing.assert_array_equal(Im, I)
        np.testing.assert_array_equal(Dm, D)


def test_synthetic_rows_are_unit_norm_and_position_independent():
    a = oracle.synth_rows(1234, 0, 64, 768)
    b = oracle.synth_rows(1234, 32, 32, 768)
    np.testing.assert_array_equal(a[32:], b)
    assert np.abs(np.linalg.norm(a.astype(np.float64), axis=1) - 1).max() < 1e-6


def test_oracle_nonfinite_scores_follow_the_faiss_heap_rule():
    """faiss IndexFlat.search selects with a heap that starts at -FLT_MAX (IP) / +FLT_MAX (L2) and admits a
    candidate only if it compares strictly better than the heap top (published behaviour of
    faiss/utils/Heap.h + the flat result handlers; call site faiss_store.py:152).  Restated literally
    here in numpy over the oracle's own score matrix: NaN, -inf and -FLT_MAX never enter, +inf ranks
    first, the rest is ordinary ordering."""
    rng = np.random.default_rng(0)
    N, d, nq, k = 500, 24, 6, 8
    X = rng.standard_normal((N, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    X[3, 2] = np.nan
    X[4, :] = np.inf
    X[5, 0] = -np.inf
    X[6, :] = 3.0e38
    Q[0, :] = 0.0
    Q[1, 1] = np.nan
    Q[2, :] = np.abs(Q[2, :])
    fmax = np.finfo(np.float32).max
    for metric in (0, 1):
        S = oracle.scores(X, Q, metric)                     # ranking scores (larger is better)
        D, I = oracle.search(X, Q, k, metric)
        qn = np.array([oracle.dot(q, q) for q in Q], dtype=np.float32)
        for b in range(nq):
            with np.errstate(invalid="ignore"):
                admitted = [r for r in range(N) if S[b, r] > -fmax]          # NaN compares false
            order = sorted(admitted, key=lambda r: (-S[b, r], r))[:k]
            want_i = order + [-1] * (k - len(order))
            np.testing.assert_array_equal(I[b], np.array(want_i))
            for j, r in enumerate(order):
                with np.errstate(invalid="ignore", over="ignore"):
                    want = S[b, r] if metric == 0 else max(np.float32(0), np.float32(qn[b] - S[b, r]))
                assert D[b, j] == want or (np.isnan(D[b, j]) and np.isnan(want))
            assert (D[b, len(order):] == (-fmax if metric == 0 else fmax)).all()
        assert (I[1] == -1).all() and 3 not in I
