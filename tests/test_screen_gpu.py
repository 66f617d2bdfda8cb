"""GPU parity of the two-stage search (fp16 screening scan + exact fp32 second stage + certificate,
include/rag_amd.h rag_index_set_screening): results must be what the one-pass fp32 search and the CPU
oracle return, bit for bit, whether the certificate holds or the fallback runs."""
import os

import numpy as np
import pytest

from oracle import flat as oracle

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def _screened(X, metric=0):
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16, FlatIndex
    idx = FlatIndex(X.shape[1], metric)
    idx.add(X)
    idx.set_screening(SCREEN_FP16)
    return idx


def _check(idx, X, Q, k, metric=0):
    D, I = idx.search(Q, k)
    Do, Io = oracle.search(X, Q, k, metric)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))


CASES = [
    # (N, d, nq, k, metric)
    (200_000, 384, 32, 10, 0),
    (100_000, 768, 32, 10, 0),
    (100_000, 768, 32, 1, 0),
    (60_000, 768, 7, 16, 0),
    (60_000, 768, 32, 17, 0),     # k' = 128
    (60_000, 384, 45, 48, 0),     # two passes of queries
    (60_000, 384, 32, 49, 0),     # k' = 240
    (80_000, 768, 32, 100, 0),    # the rerank profile's k
    (50_000, 768, 32, 10, 1),     # L2
    (50_000, 384, 5, 100, 1),
    (30_000, 100, 32, 10, 0),     # d padded to 128 in the fp16 copy
    (30_000, 200, 3, 20, 1),      # padded to 256, ring of 4
    (20_000, 1024, 32, 10, 0),
    (20_000, 1536, 32, 10, 0),    # d > 1024: the verify kernel takes its rows in two passes
    (8_000, 1536, 5, 48, 1),      # largest k the LDS budget leaves at d = 1536
    (10_000, 2048, 9, 16, 0),     # largest d the screen covers
    (20_000, 1000, 32, 60, 0),    # fallback needs rounds (k > max_k(1000) = 48) if it ever runs
    (33, 64, 32, 10, 0),          # fewer rows than candidates
    (600_000, 64, 32, 100, 0),    # sample pass seeds the screening pass's band thresholds
    (600_000, 64, 7, 33, 1),
    (5, 64, 2, 10, 0),            # k > N: -1 padding
]


@pytest.mark.parametrize("N,d,nq,k,metric", CASES)
def test_screened_search_matches_oracle_bit_exact(gpu_required, N, d, nq, k, metric):
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16
    rng = np.random.default_rng(99 + N + d + k)
    X, Q = _unit(rng, N, d), _unit(rng, nq, d)
    idx = _screened(X, metric)
    assert idx.screening == SCREEN_FP16
    _check(idx, X, Q, k, metric)
    st = idx.screen_stats()
    assert st["queries"] == nq
    assert st["max_err_ratio"] < 1.0          # observed error inside the worst-case bound
    idx.close()


def test_error_bound_has_headroom_and_random_data_rarely_falls_back(gpu_required):
    rng = np.random.default_rng(7)
    X = _unit(rng, 300_000, 768)
    idx = _screened(X)
    for seed in range(4):
        Q = _unit(np.random.default_rng(seed), 32, 768)
        _check(idx, X, Q, 10)
    st = idx.screen_stats()
    assert st["queries"] == 128
    assert st["fallbacks"] <= 2, st
    assert 0.0 < st["max_err_ratio"] < 0.25, st   # fp16 rounding errors do not line up in practice
    idx.close()


def test_dense_near_ties_fall_back_and_stay_exact(gpu_required):
    # 400 rows within ~1e-6 of each other at the top of every query's ranking: the band cannot be
    # closed, the certificate fails, the fp32 fallback answers.
    rng = np.random.default_rng(11)
    d = 384
    X = _unit(rng, 40_000, d)
    centre = _unit(rng, 1, d)[0]
    clones = centre[None, :] + 1e-6 * rng.standard_normal((400, d)).astype(np.float32)
    clones /= np.linalg.norm(clones, axis=1, keepdims=True)
    where = rng.choice(len(X), size=400, replace=False)
    X[where] = clones.astype(np.float32)
    Q = np.vstack([centre[None, :], _unit(rng, 31, d)]).astype(np.float32)
    idx = _screened(X)
    _check(idx, X, Q, 10)
    st = idx.screen_stats()
    assert st["fallbacks"] >= 1               # query 0 at least
    _check(idx, X, Q, 100)
    idx.close()


def test_exact_duplicates_rank_by_ascending_id(gpu_required):
    rng = np.random.default_rng(12)
    X = _unit(rng, 20_000, 128)
    X[[5, 77, 4000, 19_999]] = X[123]
    Q = X[[123]].copy()
    idx = _screened(X)
    D, I = idx.search(Q, 8)
    assert I[0, :5].tolist() == [5, 77, 123, 4000, 19_999]
    _check(idx, X, Q, 8)
    idx.close()


@pytest.mark.parametrize("scale", [1e-4, 37.0, 2.5e4])
def test_unnormalised_magnitudes(gpu_required, scale):
    rng = np.random.default_rng(13)
    X = (rng.standard_normal((50_000, 256)) * scale).astype(np.float32)
    Q = (rng.standard_normal((16, 256)) * (1.0 / scale if scale > 1 else 3.0)).astype(np.float32)
    for metric in (0, 1):
        idx = _screened(X, metric)
        _check(idx, X, Q, 10, metric)
        assert idx.screen_stats()["max_err_ratio"] < 1.0
        idx.close()


def test_wide_dynamic_range_inside_rows(gpu_required):
    # a few large coordinates and many tiny ones: elements below the fp16 range after scaling are
    # covered by the bound's absolute term
    rng = np.random.default_rng(14)
    X = (rng.standard_normal((40_000, 128)) * 1e-6).astype(np.float32)
    X[:, :4] = rng.standard_normal((40_000, 4)).astype(np.float32) * 50.0
    Q = rng.standard_normal((8, 128)).astype(np.float32)
    idx = _screened(X)
    _check(idx, X, Q, 10)
    assert idx.screen_stats()["max_err_ratio"] < 1.0
    idx.close()


def test_incremental_add_rescales_the_copy(gpu_required):
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16, FlatIndex
    rng = np.random.default_rng(15)
    A = _unit(rng, 30_000, 384)
    B = (_unit(rng, 20_000, 384) * 300.0).astype(np.float32)   # needs a new scale: everything is converted again
    Cc = _unit(rng, 10_000, 384)
    idx = FlatIndex(384)
    idx.set_screening(SCREEN_FP16)            # enabled on an empty index
    Q = _unit(rng, 32, 384)
    idx.add(A)
    _check(idx, A, Q, 10)
    idx.add(B)
    _check(idx, np.vstack([A, B]), Q, 10)
    idx.add(Cc)
    _check(idx, np.vstack([A, B, Cc]), Q, 10)
    assert idx.screening == SCREEN_FP16
    idx.close()


def test_out_of_range_corpus_disables_screening_not_results(gpu_required):
    from rag_inference_pipeline_amd.flat_index import SCREEN_INACTIVE
    rng = np.random.default_rng(16)
    X = _unit(rng, 10_000, 64)
    X[17, 3] = 3e20                            # beyond the magnitudes the bound covers
    Q = _unit(rng, 4, 64)
    idx = _screened(X)
    assert idx.screening == SCREEN_INACTIVE
    _check(idx, X, Q, 10)
    assert idx.screen_stats()["queries"] == 0  # answered by the fp32 scan
    idx.close()


def test_bad_queries_fall_back_per_query(gpu_required):
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    rng = np.random.default_rng(17)
    X = _unit(rng, 30_000, 128)
    Q = _unit(rng, 6, 128)
    Q[1] *= 1e15                               # out of range: that query alone goes to the fp32 scan
    Q[4] = 0.0                                 # all-zero query: every score 0, ids 0..k-1
    plain = FlatIndex(128)
    plain.add(X)
    want = plain.search(Q, 10)
    idx = _screened(X)
    got = idx.search(Q, 10)
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    assert got[1][4].tolist() == list(range(10))
    st = idx.screen_stats()
    assert st["fallbacks"] >= 1
    plain.close()
    idx.close()


def test_screening_limits_and_switching_off(gpu_required):
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16, SCREEN_OFF, FlatIndex
    wide = FlatIndex(2100)
    with pytest.raises(RuntimeError, match="two-stage search covers"):
        wide.set_screening(SCREEN_FP16)
    wide.close()
    rng = np.random.default_rng(180)
    X, Q = _unit(rng, 6_000, 1536), _unit(rng, 8, 1536)
    idx = _screened(X)
    _check(idx, X, Q, 100)                     # k = 100 does not fit beside a 1536-wide query image: fp32 scan
    assert idx.screen_stats()["queries"] == 0
    _check(idx, X, Q, 20)
    assert idx.screen_stats()["queries"] == 8
    idx.close()
    rng = np.random.default_rng(18)
    X, Q = _unit(rng, 20_000, 384), _unit(rng, 32, 384)
    idx = _screened(X)
    _check(idx, X, Q, 150)                     # k beyond the two-stage range: plain fp32 rounds
    assert idx.screen_stats()["queries"] == 0
    _check(idx, X, Q, 10)
    idx.set_screening(SCREEN_OFF)
    assert idx.screening == SCREEN_OFF
    _check(idx, X, Q, 10)
    idx.close()


def test_screened_device_api_and_id_offset(gpu_required):
    import torch
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16, FlatIndex
    rng = np.random.default_rng(19)
    X, Q = _unit(rng, 50_000, 768), _unit(rng, 32, 768)
    idx = FlatIndex(768)
    idx.add_device(torch.from_numpy(X).cuda().data_ptr(), len(X))
    idx.set_id_offset(1_000_000)
    idx.set_screening(SCREEN_FP16)
    q = torch.from_numpy(Q).cuda()
    D = torch.empty((32, 10), dtype=torch.float32, device="cuda")
    I = torch.empty((32, 10), dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):                         # back-to-back searches share the workspace
        idx.search_device(q.data_ptr(), 32, 10, D.data_ptr(), I.data_ptr(), st)
    torch.cuda.synchronize()
    Do, Io = oracle.search(X, Q, 10, 0, 1_000_000)
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))
    idx.close()


def test_deferred_fallback_reports_through_the_flag_word(gpu_required):
    """rag_index_search_device_ex: RAG_SEARCH_DEFER_FALLBACK leaves out the fallback launches and says in a device
    word whether the result is final; RAG_SEARCH_EXACT_ONE_PASS is the repeat.  Dense near-ties (certificate
    fails) and ordinary queries, one and two blocks of queries, and the host-output entry point."""
    import torch
    from rag_inference_pipeline_amd.flat_index import (SEARCH_DEFAULT, SEARCH_DEFER_FALLBACK, SEARCH_EXACT_ONE_PASS,
                                                       merge_topk_packed_flagged_device)
    from rag_inference_pipeline_amd.sharded import pack_layout
    rng = np.random.default_rng(23)
    d, k = 384, 10
    X = _unit(rng, 40_000, d)
    centre = _unit(rng, 1, d)[0]
    clones = centre[None, :] + 1e-6 * rng.standard_normal((400, d)).astype(np.float32)
    clones /= np.linalg.norm(clones, axis=1, keepdims=True)
    X[rng.choice(len(X), size=400, replace=False)] = clones.astype(np.float32)
    idx = _screened(X)
    st = torch.cuda.current_stream().cuda_stream
    flag = torch.full((1,), 77, dtype=torch.int32, device="cuda")
    for nq, hard in [(32, False), (32, True), (40, True), (40, False), (5, True)]:
        Q = _unit(rng, nq, d)
        if hard:
            Q[nq - 1] = centre     # in the second block when nq > 32: that block must not clear the word... and the
            #                        first block's clear must not hide it either
        q = torch.from_numpy(Q).cuda()
        D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        Do, Io = oracle.search(X, Q, k)
        flag.fill_(77)
        idx.search_device_ex(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), SEARCH_DEFER_FALLBACK, flag.data_ptr(), st)
        torch.cuda.synchronize()
        assert int(flag.item()) == (1 if hard else 0)
        if not hard:   # final: identical to the oracle
            np.testing.assert_array_equal(I.cpu().numpy(), Io)
            np.testing.assert_array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))
        for mode in (SEARCH_EXACT_ONE_PASS, SEARCH_DEFAULT):   # the repeat, and the self-contained mode: flag cleared
            flag.fill_(77)
            idx.search_device_ex(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), mode, flag.data_ptr(), st)
            torch.cuda.synchronize()
            assert int(flag.item()) == 0
            np.testing.assert_array_equal(I.cpu().numpy(), Io)
            np.testing.assert_array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))
        Dh, Ih = idx.search_from_device(q.data_ptr(), nq, k, st)   # device queries in, host results out
        np.testing.assert_array_equal(Ih, Io)
        np.testing.assert_array_equal(Dh.view(np.uint32), Do.view(np.uint32))
    # screening off: nothing to defer, the word reads 0
    idx.set_screening(0)
    flag.fill_(77)
    idx.search_device_ex(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), SEARCH_DEFER_FALLBACK, flag.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    idx.close()

    # the shard merge ORs the ranks' words
    nq, world = 4, 3
    s_off, f_off, nbytes = pack_layout(nq, k)
    for raised in ([0, 0, 0], [0, 1, 0], [1, 0, 1]):
        host = np.zeros((world, nbytes), dtype=np.uint8)
        for g in range(world):
            host[g, :s_off].view(np.int64)[:] = np.arange(nq * k) + 1000 * g
            host[g, s_off:f_off].view(np.float32)[:] = np.sort(rng.standard_normal((nq, k)).astype(np.float32))[:, ::-1].ravel()
            host[g, f_off:f_off + 4].view(np.uint32)[:] = raised[g]
        buf = torch.from_numpy(host).cuda()
        os_, oi_ = torch.empty((nq, k), dtype=torch.float32, device="cuda"), torch.empty((nq, k), dtype=torch.int64, device="cuda")
        any_ = torch.full((1,), 55, dtype=torch.int32, device="cuda")
        mirror = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        merge_topk_packed_flagged_device(0, 0, world, nq, k, buf.data_ptr(), nbytes, s_off, f_off, os_.data_ptr(),
                                         oi_.data_ptr(), any_.data_ptr(), mirror.data_ptr(), st)
        torch.cuda.synchronize()
        assert int(any_.item()) == int(any(raised))
        m = mirror.numpy()   # the kernel's own copy of the result in pinned host memory
        np.testing.assert_array_equal(m[:s_off].view(np.int64).reshape(nq, k), oi_.cpu().numpy())
        np.testing.assert_array_equal(m[s_off:f_off].view(np.float32).reshape(nq, k), os_.cpu().numpy())
        assert int(m[f_off:f_off + 4].view(np.uint32)[0]) == int(any(raised))
        all_s = np.stack([host[g, s_off:f_off].view(np.float32).reshape(nq, k) for g in range(world)])
        all_i = np.stack([host[g, :s_off].view(np.int64).reshape(nq, k) for g in range(world)])
        Dm, Im = oracle.merge(all_s, all_i, 0)
        np.testing.assert_array_equal(oi_.cpu().numpy(), Im)
        np.testing.assert_array_equal(os_.cpu().numpy(), Dm)


def test_randomised_corpora_two_stage_matches_oracle(gpu_required):
    """40 seeded random draws: shapes, metrics, and the data the certificate finds hard — tight
    clusters (dense near-ties around rank k), exact duplicates, quantised coordinates, skewed scales."""
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16
    rng = np.random.default_rng(20261005)
    dims = [8, 24, 64, 100, 200, 384, 520, 768, 1024]
    fallbacks = queries = 0
    for trial in range(int(os.environ.get("RAG_AMD_TEST_TRIALS", "40"))):
        d = int(rng.choice(dims))
        N = int(rng.integers(1, 20_000 if d <= 384 else 8_000))
        nq = int(rng.integers(1, 50))
        k = int(rng.choice([1, 3, 10, 16, 17, 48, 49, 100]))
        metric = int(rng.integers(0, 2))
        kind = trial % 5
        if kind == 0:    # plain Gaussian, random overall scale
            X = rng.standard_normal((N, d), dtype=np.float32) * np.float32(rng.choice([1e-3, 1.0, 300.0]))
        elif kind == 1:  # tight clusters: many rows within the band of each other
            centres = rng.standard_normal((max(1, N // 200), d), dtype=np.float32)
            X = centres[rng.integers(0, len(centres), size=N)] + np.float32(1e-3) * rng.standard_normal((N, d), dtype=np.float32)
        elif kind == 2:  # exact duplicates
            X = rng.standard_normal((N, d), dtype=np.float32)
            X[N // 2:] = X[: N - N // 2]
        elif kind == 3:  # coordinates on a coarse grid: exact score ties between different rows
            X = rng.integers(-3, 4, size=(N, d)).astype(np.float32)
        else:            # one dominant coordinate
            X = rng.standard_normal((N, d), dtype=np.float32)
            X[:, 0] *= 100.0
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        if kind == 1:
            Q = X[rng.integers(0, N, size=nq)] + np.float32(1e-2) * Q   # queries inside the clusters
        idx = _screened(np.ascontiguousarray(X, dtype=np.float32), metric)
        assert idx.screening == SCREEN_FP16
        D, I = idx.search(np.ascontiguousarray(Q, dtype=np.float32), k)
        Do, Io = oracle.search(X, Q, k, metric)
        msg = f"trial {trial}: kind={kind} N={N} d={d} nq={nq} k={k} metric={metric}"
        np.testing.assert_array_equal(I, Io, err_msg=msg)
        np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32), err_msg=msg)
        st = idx.screen_stats()
        assert st["max_err_ratio"] < 1.0, msg
        fallbacks += st["fallbacks"]
        queries += st["queries"]
        idx.close()
    assert queries > 0 and fallbacks > 0      # the hard corpora did exercise the fallback ...
    assert fallbacks < queries                # ... and the easy ones did not


def test_per_search_flags_do_not_leak_between_searches(gpu_required):
    """The certificate / fallback words are epoch-valued (flat_kernels.hip.h ScreenQueryState): a flag is set for
    the search whose id it holds, nothing clears them.  Alternate batches that force fallbacks (NaN and huge
    queries, dense duplicates) with batches that need none, through one handle: every result exact, and the
    fallback counter moves only for the batches that asked for it."""
    rng = np.random.default_rng(42)
    X = _unit(rng, 70_000, 128)
    X[30_000:30_400] = X[7]                       # 400 exact duplicates of one row: a band that cannot fit
    idx = _screened(X)
    good = _unit(rng, 32, 128)
    bad = good.copy()
    bad[3, 5] = np.nan
    bad[11, :] *= 1.0e30
    bad[20] = X[7]                                # lands in the duplicate cluster
    idx.screen_stats(reset=True)
    seen = 0
    for rnd in range(6):
        Q = bad if rnd % 2 == 0 else good
        _check(idx, X, Q, 10)
        st = idx.screen_stats()
        if rnd % 2 == 0:
            assert st["fallbacks"] >= seen + 3, (rnd, st)
        else:
            assert st["fallbacks"] == seen, (rnd, st)   # a clean batch right after a dirty one: no stale flag
        seen = st["fallbacks"]
    # different batch sizes and k on the same handle, dirty then clean
    _check(idx, X, bad[:5], 16)
    seen = idx.screen_stats()["fallbacks"]
    _check(idx, X, good[:5], 16)
    _check(idx, X, good[:1], 1)
    assert idx.screen_stats()["fallbacks"] == seen
    idx.close()


def test_randomised_large_shapes_full_rounds_and_tail_round(gpu_required):
    """Seeded random shapes large enough for full rounds of the tile walk plus a random leftover (the
    one-per-workgroup tail round), k on both sides of the warm-start limit (16), both metrics, one-pass and
    two-stage on the same handle, ragged batches."""
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16, SCREEN_OFF, FlatIndex
    rng = np.random.default_rng(20261005)
    for trial in range(int(os.environ.get("RAG_AMD_TEST_TRIALS_LARGE", "10"))):
        d = int(rng.choice([8, 24, 40, 64]))
        N = int(rng.integers(66_000, 330_000))
        nq = int(rng.integers(1, 45))
        k = int(rng.choice([1, 3, 10, 16, 17, 40, 100]))
        metric = int(rng.integers(0, 2))
        X = rng.standard_normal((N, d), dtype=np.float32)
        if trial % 3 == 0:
            X[N // 3: N // 3 + 5000] = X[:5000]          # ties across distant tiles
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        idx = FlatIndex(d, metric)
        idx.add(X)
        Do, Io = oracle.search(X, Q, k, metric)
        for mode in (SCREEN_OFF, SCREEN_FP16):
            idx.set_screening(mode)
            D, I = idx.search(Q, k)
            msg = f"trial {trial}: N={N} d={d} nq={nq} k={k} metric={metric} mode={mode}"
            np.testing.assert_array_equal(I, Io, err_msg=msg)
            np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32), err_msg=msg)
        idx.close()
