"""GPU parity: HIP flat index (through the C ABI) vs the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

from oracle import flat as oracle

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def _index(X, metric=0):
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    idx = FlatIndex(X.shape[1], metric)
    idx.add(X)
    return idx


CASES = [
    # (N, d, nq, k, metric)
    (10_000, 384, 1, 10, 0),      # BASELINE config A shape
    (4096, 768, 32, 10, 0),
    (4096, 768, 32, 100, 0),
    (20_000, 384, 32, 100, 0),
    (5000, 384, 7, 10, 1),
    (4096, 768, 32, 10, 1),
    (1000, 64, 32, 5, 0),
    (777, 100, 3, 10, 0),         # d not a multiple of 8, ragged N
    (33, 8, 32, 10, 0),
    (70_001, 128, 40, 10, 0),     # nq > 32 -> two passes
    (9_000, 1536, 32, 10, 0),     # d > 1024 -> two column chunks with parked partial sums
    (5_000, 1100, 9, 20, 1),      # ragged chunks (768 + 336 padded), L2
    (3_000, 3072, 32, 100, 0),    # four chunks, k = 100
    (30_000, 384, 32, 300, 0),    # k > max_k(384) = 240 -> two rounds with a key ceiling
    (6_000, 768, 5, 250, 1),      # k > max_k(768) = 112 -> three rounds, L2
    (200, 768, 3, 150, 0),        # rounds run past the end of a tiny corpus (-1 padding)
    (600_000, 64, 32, 100, 0),    # k >= 32 on a large corpus: thresholds seeded by the sample pass
    (700_001, 40, 9, 40, 1),      # same, L2, ragged
    (600_000, 64, 5, 300, 0),     # sample pass inside key-ceiling rounds
    (65_536 + 32 * 151, 64, 32, 10, 0),   # 1 full round + 151 leftover tiles: the one-per-workgroup tail round
    (65_536 * 2 + 32 * 700 + 5, 32, 9, 10, 1),  # 700 leftover tiles: waves 0..2 of the tail round, ragged last tile
    (30_000, 384, 32, 250, 0),    # second key-ceiling round has k = 10: warm start under a ceiling
    (4096, 768, 32, 16, 0),       # k = 16: the largest warm-start k (16 half-tile maxima)
    (300, 768, 32, 10, 0),        # fewer tiles than waves: inactive waves post -inf maxima
]


@pytest.mark.parametrize("N,d,nq,k,metric", CASES)
def test_search_matches_oracle_bit_exact(gpu_required, N, d, nq, k, metric):
    rng = np.random.default_rng(1234 + N + d)
    X, Q = _unit(rng, N, d), _unit(rng, nq, d)
    idx = _index(X, metric)
    assert idx.ntotal == N
    D, I = idx.search(Q, k)
    Do, Io = oracle.search(X, Q, k, metric)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))  # bit-exact scores
    idx.close()


def test_k_greater_than_n_pads_like_faiss(gpu_required):
    rng = np.random.default_rng(5)
    X, Q = _unit(rng, 5, 64), _unit(rng, 2, 64)
    D, I = _index(X).search(Q, 8)
    Do, Io = oracle.search(X, Q, 8)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert (I[:, 5:] == -1).all() and (D[:, 5:] == -np.finfo(np.float32).max).all()


def test_empty_index(gpu_required):
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    idx = FlatIndex(32)
    D, I = idx.search(np.zeros((3, 32), np.float32), 4)
    assert (I == -1).all() and (D == -np.finfo(np.float32).max).all()


def test_ties_break_by_ascending_id(gpu_required):
    rng = np.random.default_rng(9)
    base = _unit(rng, 50, 128)
    X = np.concatenate([base, base, base[:25]])  # every row duplicated: exact score ties
    Q = _unit(rng, 32, 128)
    D, I = _index(X).search(Q, 10)
    Do, Io = oracle.search(X, Q, 10)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    # all-equal scores (zero queries): ids 0..k-1
    D0, I0 = _index(X).search(np.zeros((2, 128), np.float32), 10)
    np.testing.assert_array_equal(I0, np.tile(np.arange(10), (2, 1)))


def test_adversarial_ascending_scores(gpu_required):
    # scores increase with the row number: every row beats the running threshold
    d, N = 64, 50_000
    q = np.zeros((1, d), np.float32); q[0, 0] = 1.0
    X = np.zeros((N, d), np.float32); X[:, 0] = np.linspace(-1, 1, N, dtype=np.float32)
    Q = np.repeat(q, 32, axis=0)
    for k in (10, 100):
        D, I = _index(X).search(Q, k)
        Do, Io = oracle.search(X, Q, k)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)


def test_synthetic_corpus_matches_oracle_generator(gpu_required):
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    idx = FlatIndex(768)
    idx.add_synthetic(3000, seed=1234)
    got = idx.get_rows(0, 3000)
    want = oracle.synth_rows(1234, 0, 3000, 768)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    Q = oracle.synth_rows(4321, 0, 32, 768)
    D, I = idx.search(Q, 10)
    Do, Io = oracle.search(want, Q, 10)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def test_incremental_add_and_id_offset(gpu_required):
    rng = np.random.default_rng(11)
    X, Q = _unit(rng, 3000, 384), _unit(rng, 8, 384)
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    idx = FlatIndex(384)
    for part in np.array_split(X, 7):
        idx.add(part)
    idx.set_id_offset(1_000_000)
    D, I = idx.search(Q, 10)
    Do, Io = oracle.search(X, Q, 10, id_offset=1_000_000)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)


def _nonfinite_case(metric):
    rng = np.random.default_rng(77 + metric)
    N, d, nq = 6000, 72, 12
    X = rng.standard_normal((N, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    X[5, 3] = np.nan                     # NaN row: every score NaN -> never a result
    X[6, :] = np.inf                     # +inf / NaN scores depending on the query's signs
    X[7, 0] = -np.inf
    X[8, :] = 3.0e38                     # overflows to +-inf (or inf - inf = NaN) in the sum
    X[9, 10] = np.inf; X[9, 11] = -np.inf
    X[4000:4004, 1] = np.inf             # equal +inf scores: ties by id
    Q[0, :] = 0.0                        # 0 * inf = NaN for the inf rows, exact zeros elsewhere
    Q[1, 5] = np.nan                     # NaN query: no result at all
    Q[2, 0] = np.inf
    Q[3, :] = np.abs(Q[3, :])            # all positive: row 6 scores +inf and must rank first
    Q[4, :] = 3.0e38
    return X, Q


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("two_stage", [False, True])
def test_nonfinite_rows_and_queries_follow_the_faiss_heap_rule(gpu_required, metric, two_stage):
    """include/rag_amd.h, "Non-finite values": a row whose ranking score is NaN, -inf or -FLT_MAX is never
    returned (faiss's heap admits only strictly-better-than-neutral candidates), +inf ranks first, ties by
    id; a NaN query gets k padding slots.  Bit-exact against the oracle, with the two-stage search
    requested as well (a non-finite corpus switches it to inactive, so the fp32 scan answers)."""
    from rag_inference_pipeline_amd.flat_index import SCREEN_INACTIVE
    X, Q = _nonfinite_case(metric)
    idx = _index(X, metric)
    if two_stage:
        idx.set_screening(True)
        assert idx.screening == SCREEN_INACTIVE
    for k in (10, 100):
        D, I = idx.search(Q, k)
        Do, Io = oracle.search(X, Q, k, metric)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
        assert not np.isnan(D).any() and 5 not in I
        assert (I[1] == -1).all()                                  # NaN query
    idx.close()


def test_nonfinite_queries_on_a_finite_corpus_through_the_two_stage_search(gpu_required):
    """A finite corpus keeps the fp16 screening copy; NaN / inf / huge queries fail the screening
    certificate by construction and are answered by the fp32 fallback — same bits as the oracle."""
    from rag_inference_pipeline_amd.flat_index import SCREEN_FP16
    rng = np.random.default_rng(5)
    X, Q = _unit(rng, 40_000, 128), _unit(rng, 32, 128)
    Q[3, 7] = np.nan
    Q[9, 0] = np.inf
    Q[17, :] = -np.inf
    Q[21, :] *= 1.0e30
    idx = _index(X)
    idx.set_screening(True)
    assert idx.screening == SCREEN_FP16
    D, I = idx.search(Q, 10)
    Do, Io = oracle.search(X, Q, 10)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    st = idx.screen_stats()
    assert st["fallbacks"] >= 3
    idx.close()


def test_global_ids_must_fit_32_bits(gpu_required):
    """the shard merge carries the global id in 32 bits — an offset that could wrap is refused."""
    from rag_inference_pipeline_amd import _native
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    idx = FlatIndex(16)
    idx.add(np.zeros((100, 16), np.float32))
    idx.set_id_offset(2**32 - 101)                      # last id = 2^32 - 2: fine
    for bad in (2**32 - 50, 2**32, -1, 2**40):
        with pytest.raises(_native.RagAmdError):
            idx.set_id_offset(bad)
    with pytest.raises(_native.RagAmdError):            # growing past the limit under a set offset
        idx.add(np.zeros((5, 16), np.float32))
    idx.close()


def test_hip_reproduces_committed_golden(gpu_required):
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flat_search.json")) as fh:
        cases = json.load(fh)
    for name, c in cases.items():
        X = oracle.synth_rows(c["corpus_seed"], 0, c["n"], c["d"])
        Q = oracle.synth_rows(c["query_seed"], 0, c["nq"], c["d"])
        D, I = _index(X, c["metric"]).search(Q, c["k"])
        np.testing.assert_array_equal(I, np.array(c["ids"], dtype=np.int64), err_msg=name)
        np.testing.assert_array_equal(D.view(np.uint32), np.array(c["scores_bits"], dtype=np.uint32), err_msg=name)


def test_full_size_properties_10m_768(gpu_required):
    """BASELINE size (10M x 768, batch 32, k 10): size-independent checks against the oracle.
    (1) returned scores are bit-exact canonical dot products of the returned rows;
    (2) lists are sorted (score desc, id asc) with distinct ids;
    (3) no row of a 200k-row random sample beats the k-th result (completeness on a sample);
    (4) planted queries (a corpus row itself) come back as their own top-1."""
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    N, d, B, k, seed = 10_000_000, 768, 32, 10, 1234
    idx = FlatIndex(d)
    idx.add_synthetic(N, seed)
    Q = oracle.synth_rows(4321, 0, B, d)
    planted = [17, 5_000_001, 9_999_999]
    for j, r in enumerate(planted):
        Q[j] = oracle.synth_rows(seed, r, 1, d)[0]
    D, I = idx.search(Q, k)
    for j, r in enumerate(planted):
        assert I[j, 0] == r
    assert ((I >= 0) & (I < N)).all()
    for b in range(B):
        assert len(set(I[b].tolist())) == k
        rows = np.concatenate([oracle.synth_rows(seed, int(r), 1, d) for r in I[b]])
        np.testing.assert_array_equal(idx.get_rows(int(I[b, 0]), 1), rows[:1])
        want = oracle.scores(rows, Q[b:b + 1])[0]
        np.testing.assert_array_equal(D[b].view(np.uint32), want.view(np.uint32))
        key = list(zip((-D[b]).tolist(), I[b].tolist()))
        assert key == sorted(key)
    rng = np.random.default_rng(0)
    starts = rng.integers(0, N - 2000, size=100)
    for s in starts:
        Xs = oracle.synth_rows(seed, int(s), 2000, d)
        Ds, Is = oracle.search(Xs, Q, 1, id_offset=int(s))
        # the best row of the sample must not beat the k-th result unless it is in the result
        for b in range(B):
            if Ds[b, 0] > D[b, k - 1] or (Ds[b, 0] == D[b, k - 1] and Is[b, 0] < I[b, k - 1]):
                assert Is[b, 0] in I[b]
    idx.close()


def test_randomised_shapes_match_oracle(gpu_required):
    """40 seeded random (N, d, nq, k, metric) draws, including column-chunked d, multi-round k,
    multi-pass nq, duplicated rows and un-normalised data."""
    rng = np.random.default_rng(20261004)
    dims = [8, 24, 64, 100, 200, 384, 520, 768, 1032, 1536]
    for trial in range(int(os.environ.get("RAG_AMD_TEST_TRIALS", "40"))):
        d = int(rng.choice(dims))
        N = int(rng.integers(1, 12_000 if d <= 768 else 3_000))
        nq = int(rng.integers(1, 71))
        k = int(rng.choice([1, 2, 7, 10, 33, 100, 130, 260]))
        metric = int(rng.integers(0, 2))
        X = rng.standard_normal((N, d), dtype=np.float32) * np.float32(rng.choice([0.1, 1.0, 30.0]))
        if trial % 4 == 0 and N > 10:  # exact duplicates -> ties
            X[N // 2:] = X[: N - N // 2]
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        idx = _index(X, metric)
        D, I = idx.search(Q, k)
        Do, Io = oracle.search(X, Q, k, metric)
        np.testing.assert_array_equal(I, Io, err_msg=f"trial {trial}: N={N} d={d} nq={nq} k={k} metric={metric}")
        np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32), err_msg=f"trial {trial}")
        idx.close()


def test_concurrent_searches_on_one_handle(gpu_required):
    """The reference's scheduler runs batches concurrently on worker threads
    (batch_scheduler.py:286-288): one handle, many threads, every answer still exact."""
    import threading
    rng = np.random.default_rng(3)
    X = _unit(rng, 30_000, 384)
    idx = _index(X)
    queries = [_unit(rng, int(n), 384) for n in (1, 32, 7, 40, 16, 32, 3, 25)]
    want = [oracle.search(X, q, 10) for q in queries]
    errors = []

    def work(i):
        try:
            for _ in range(5):
                D, I = idx.search(queries[i], 10)
                np.testing.assert_array_equal(I, want[i][1])
                np.testing.assert_array_equal(D, want[i][0])
        except Exception as exc:  # noqa: BLE001
            errors.append((i, exc))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(queries))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    idx.close()


def test_device_search_accepts_unaligned_query_pointer(gpu_required):
    """search_device on a query buffer that starts 4 bytes into an allocation (a tensor slice)."""
    import torch
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    rng = np.random.default_rng(8)
    X, Q = _unit(rng, 5000, 64), _unit(rng, 8, 64)
    idx = FlatIndex(64)
    idx.add(X)
    buf = torch.zeros(1 + Q.size, dtype=torch.float32, device="cuda")
    buf[1:] = torch.from_numpy(Q.ravel()).cuda()
    out_s = torch.empty((8, 10), dtype=torch.float32, device="cuda")
    out_i = torch.empty((8, 10), dtype=torch.int64, device="cuda")
    idx.search_device(buf[1:].data_ptr(), 8, 10, out_s.data_ptr(), out_i.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Do, Io = oracle.search(X, Q, 10)
    np.testing.assert_array_equal(out_i.cpu().numpy(), Io)
    np.testing.assert_array_equal(out_s.cpu().numpy(), Do)


def test_ids_match_blas_sgemm_order_outside_near_ties(gpu_required):
    """Independent of the oracle's summation order: numpy fp32 `Q @ X.T` (OpenBLAS sgemm — the algorithm family
    faiss IndexFlat uses for nq >= 20) + stable (score desc, id asc) sort.  Every query's id list must equal
    the HIP list unless float64 shows the contested ranks closer than fp32 summation noise (2e-6 here; the
    full-size runs over 2016 queries are profiles/r02_parity_vs_sgemm_*.json, scripts/parity_vs_sgemm.py)."""
    N, d, nq, k = 200_000, 384, 256, 10
    X = oracle.synth_rows(1234, 0, N, d)
    Q = oracle.synth_rows(777, 0, nq, d)
    Q[1::2] = X[np.random.default_rng(1).integers(0, N, nq // 2)] + 0.05 * Q[1::2]   # dense neighbourhoods
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q = Q.astype(np.float32)
    idx = _index(X)
    idx.set_screening(True)                      # the default product path; same bits as the one-pass scan
    D, I = idx.search(Q, k)
    S = Q @ X.T
    part = np.argpartition(S, -(k + 2), axis=1)[:, -(k + 2):]
    real = 0
    for b in range(nq):
        cand = part[b]
        order = cand[np.lexsort((cand, -S[b, cand].astype(np.float64)))]
        if order[:k].tolist() == I[b].tolist():
            continue
        union = sorted(set(order.tolist()) | set(I[b].tolist()))
        f64 = dict(zip(union, (X[union].astype(np.float64) @ Q[b].astype(np.float64)).tolist()))
        for lst in (I[b].tolist(), order[:k].tolist()):
            s = np.array([f64[r] for r in lst])
            tol = max(0.0, float(np.max(s[1:] - s[:-1])), max([f64[r] for r in union if r not in lst], default=-1) - s[-1])
            real += tol > 2e-6
    assert real == 0
    assert np.abs(D - np.take_along_axis(S, I, 1)).max() < 1e-4     # north-star score tolerance vs the sgemm scores
    idx.close()


@pytest.mark.parametrize("N,d,nq,k", [(10_000, 384, 1, 10), (1_000_000, 384, 32, 10), (4096, 768, 32, 100)])
def test_against_faiss_when_the_box_has_it(gpu_required, N, d, nq, k, tmp_path):
    """probe, never assume.  When `import faiss` works, IndexFlatIP / IndexFlatL2 are the pin:
    ids must match outside float64-classified near-ties, scores within 1e-4, and index files WRITTEN BY faiss
    (IxFI flat, IwFl IVF-flat: the kinds scripts/create_test_docs.py:83-104 produces) must load through
    index_io with the same rows.  Skipped, with the reason in the report, when faiss is absent (it is not in
    this image: uv.lock pins faiss-cpu 1.13.1 for the reference, nothing installs it here)."""
    faiss = pytest.importorskip("faiss", reason="faiss is not importable on this box: the flat oracle stays pinned by the "
                                                 "sgemm / float64 cross-checks only (parity unpinned vs faiss itself)")
    from rag_inference_pipeline_amd import index_io
    X = oracle.synth_rows(1234, 0, N, d)
    Q = oracle.synth_rows(4321, 0, nq, d)
    for metric, cls in ((0, faiss.IndexFlatIP), (1, faiss.IndexFlatL2)):
        fidx = cls(d)
        fidx.add(X)
        Df, If = fidx.search(Q, k)
        idx = _index(X, metric)
        D, I = idx.search(Q, k)
        idx.close()
        assert np.abs(D - Df).max() < 1e-4
        X64, Q64 = X.astype(np.float64), Q.astype(np.float64)
        for b in range(nq):
            if I[b].tolist() == If[b].tolist():
                continue
            union = sorted(set(I[b].tolist()) | set(If[b].tolist()))
            s = X64[union] @ Q64[b] if metric == 0 else -((X64[union] - Q64[b]) ** 2).sum(1)
            f64 = dict(zip(union, s.tolist()))
            for lst in (I[b].tolist(), If[b].tolist()):
                v = np.array([f64[r] for r in lst])
                assert max(0.0, float(np.max(v[1:] - v[:-1]))) < 2e-6 and max([f64[r] for r in union if r not in lst], default=-9) - v[-1] < 2e-6
        path = str(tmp_path / f"flat{metric}.index")
        faiss.write_index(fidx, path)
        rows, m = index_io.read_index_file(path, 0)
        assert m == metric
        np.testing.assert_array_equal(np.asarray(rows), X)
    if N >= 10_000:
        quant = faiss.IndexFlatL2(d)
        ivf = faiss.IndexIVFFlat(quant, d, 64)
        ivf.train(X[:20_000])
        ivf.add(X)
        path = str(tmp_path / "ivf.index")
        faiss.write_index(ivf, path)
        rows, m = index_io.read_index_file(path, 0)
        np.testing.assert_array_equal(np.asarray(rows), X)      # unpacked back into insertion (id) order
