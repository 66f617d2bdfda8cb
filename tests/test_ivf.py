"""IVFFlat `nprobe` mode: the oracle's restatement (CPU) and the HIP path against it (gpu).

The reference's own generator writes a faiss IndexIVFFlat (scripts/create_test_docs.py:83-104) and FAISSStore.load sets
index.nprobe (faiss_store.py:84-92): on that file the reference looks only at the nprobe lists nearest to a query.
faiss is absent here, so parity is unpinned as for the flat search; what is held: oracle == a literal float64 numpy
restatement of "nearest lists, then exact top-k of their rows"; HIP == oracle (ids and fp32 score bits); nprobe = nlist
== the exhaustive flat search."""
import os

import numpy as np
import pytest

from oracle import flat as oracle
from rag_inference_pipeline_amd import index_io


def _unit(rng, n, d):
    x = rng.standard_normal((n, d), dtype=np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _file(tmp_path, n, d, nlist, metric, seed=0, sparse=False, nprobe=7):
    rng = np.random.default_rng(seed)
    X = _unit(rng, n, d)
    path = tmp_path / f"ivf_{n}_{d}_{nlist}_{metric}.index"
    index_io.write_ivfflat_index(path, X, nlist, metric, seed=seed, sparse=sparse, nprobe=nprobe, all_centroids=True)
    return X, path


def test_lists_roundtrip_and_exhaustive_view(tmp_path):
    X, path = _file(tmp_path, 3000, 24, 50, 1, sparse=True, nprobe=9)
    lists = index_io.read_ivfflat_lists(path)
    assert (lists.nlist, lists.ntotal, lists.nprobe, lists.metric, lists.quantizer_metric) == (50, 3000, 9, 1, 1)
    assert lists.offsets[0] == 0 and lists.offsets[-1] == 3000 and (np.diff(lists.offsets) >= 0).all()
    np.testing.assert_array_equal(np.sort(lists.ids), np.arange(3000))
    np.testing.assert_array_equal(lists.rows, X[lists.ids])                 # list order, stored ids alongside
    for l in (0, 17, 49):                                                   # members of a list are nearest to ITS centroid
        rows = lists.rows[lists.offsets[l]:lists.offsets[l + 1]]
        if len(rows):
            d2 = ((rows[:, None, :].astype(np.float64) - lists.centroids[None].astype(np.float64)) ** 2).sum(-1)
            assert (np.argmin(d2, axis=1) == l).all()
    rows, metric = index_io.read_index_file(path)
    np.testing.assert_array_equal(rows, X)
    assert metric == 1


@pytest.mark.parametrize("metric", [0, 1])
def test_oracle_is_nearest_lists_then_exact_topk(tmp_path, metric):
    X, path = _file(tmp_path, 4000, 32, 64, metric, seed=3)
    lists = index_io.read_ivfflat_lists(path)
    Q = _unit(np.random.default_rng(5), 9, 32)
    k, nprobe = 10, 5
    D, I = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, nprobe, metric)
    X64, Q64, C64 = X.astype(np.float64), Q.astype(np.float64), lists.centroids.astype(np.float64)
    assign = np.empty(4000, dtype=np.int64)
    for l in range(64):
        assign[lists.ids[lists.offsets[l]:lists.offsets[l + 1]]] = l
    for qi in range(9):
        near = np.argsort(((C64 - Q64[qi]) ** 2).sum(1), kind="stable")[:nprobe]      # the quantizer is L2
        cand = np.nonzero(np.isin(assign, near))[0]
        score = X64[cand] @ Q64[qi] if metric == 0 else -((X64[cand] - Q64[qi]) ** 2).sum(1)
        want = cand[np.argsort(-score, kind="stable")[:k]]
        np.testing.assert_array_equal(I[qi], want)
        assert set(np.unique(assign[I[qi]])) <= set(near.tolist())
    # every list probed: the exhaustive flat search
    Df, If = oracle.search(X, Q, k, metric)
    Da, Ia = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, 64, metric)
    np.testing.assert_array_equal(Ia, If)
    np.testing.assert_array_equal(Da.view(np.uint32), Df.view(np.uint32))


def test_list_shards_partition_every_list_and_merge_to_the_unsharded_result(tmp_path):
    """A corpus-sharded deployment in the nprobe mode: rank r keeps the centroids and rows [len r / W, len (r + 1) / W) of
    EVERY list (IVFFlatLists.shard).  The shards partition each list, and the oracle's per-shard top-k lists merged by
    (score, ascending id) — what the device merge does with the all-gathered lists — are the unsharded oracle's result."""
    X, path = _file(tmp_path, 4000, 16, 40, 1, seed=3)
    lists = index_io.read_ivfflat_lists(path)
    world = 3
    shards = [lists.shard(r, world) for r in range(world)]
    assert sum(s.ntotal for s in shards) == lists.ntotal
    for l in range(lists.nlist):
        want = lists.ids[lists.offsets[l]:lists.offsets[l + 1]]
        got = np.concatenate([s.ids[s.offsets[l]:s.offsets[l + 1]] for s in shards])
        np.testing.assert_array_equal(got, want)
        assert max(s.offsets[l + 1] - s.offsets[l] for s in shards) - min(s.offsets[l + 1] - s.offsets[l] for s in shards) <= 1
    Q = _unit(np.random.default_rng(8), 6, 16)
    k, nprobe = 7, 5
    Dw, Iw = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, nprobe, 1)
    parts = [oracle.ivf_search(s.centroids, s.quantizer_metric, s.rows, s.ids, s.offsets, Q, k, nprobe, 1) for s in shards]
    D, I = oracle.merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), 1)
    np.testing.assert_array_equal(I, Iw)
    np.testing.assert_array_equal(D, Dw)
    with pytest.raises(ValueError):
        lists.shard(3, 3)


def test_build_ivf_tool_writes_a_file_both_modes_can_load(tmp_path):
    """tools/build_ivf.py (host side; the assignment injected — on a GPU box it is this build's flat search): k-means on a
    sample, every row in exactly one list, the `IwFl` file read back as lists (nprobe mode) and as the original rows in their
    original order (exhaustive mode); training moves the centroids towards the data (lower quantisation error than the
    initial pick)."""
    from rag_inference_pipeline_amd.tools.build_ivf import build_ivf, train_centroids
    rng = np.random.default_rng(3)
    centres = rng.standard_normal((12, 20)).astype(np.float32)
    X = (centres[rng.integers(0, 12, size=4000)] + 0.1 * rng.standard_normal((4000, 20))).astype(np.float32)

    def nearest(rows, cent):
        return np.argmin((cent.astype(np.float64) ** 2).sum(1)[None, :] - 2.0 * rows.astype(np.float64) @ cent.astype(np.float64).T, axis=1)

    def qerr(cent):
        a = nearest(X, cent)
        return float(((X - cent[a]) ** 2).sum())

    assert qerr(train_centroids(X[:1000], 12, iters=25, seed=5)) < 0.7 * qerr(train_centroids(X[:1000], 12, iters=0, seed=5))
    out = tmp_path / "built.ivf"
    info = build_ivf(X, 12, 1, str(out), nprobe=3, train_rows=1000, iters=25, seed=5, assign=nearest)
    assert info["rows"] == 4000 and info["nlist"] == 12 and info["nprobe"] == 3
    lists = index_io.read_ivfflat_lists(out)
    assert (lists.nlist, lists.ntotal, lists.nprobe, lists.metric, lists.quantizer_metric) == (12, 4000, 3, 1, 1)
    np.testing.assert_array_equal(np.sort(lists.ids), np.arange(4000))
    np.testing.assert_array_equal(lists.rows, X[lists.ids])
    for l in range(12):   # every row sits in the list of its nearest centroid, lists in ascending id order
        ids = lists.ids[lists.offsets[l]:lists.offsets[l + 1]]
        assert (np.diff(ids) > 0).all() and (nearest(X[ids], lists.centroids) == l).all()
    rows, metric = index_io.read_index_file(out, 0)
    assert metric == 1
    np.testing.assert_array_equal(rows, X)


def test_settings_carry_the_ivf_mode(monkeypatch):
    from rag_inference_pipeline_amd.config import PipelineSettings
    assert PipelineSettings().faiss_ivf_mode == "exhaustive"
    assert PipelineSettings(RAG_AMD_IVF_MODE="nprobe", FAISS_NPROBE=12).faiss_nprobe == 12


# ---- gpu ----------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("n,d,nlist,metric,nq,k", [
    (60_000, 64, 4096, 1, 32, 10),      # the generator's shape of index (L2, nlist 4096), scaled down
    (60_000, 64, 4096, 0, 32, 100),
    (20_000, 384, 256, 1, 9, 10),
    (5_000, 100, 300, 0, 3, 256),       # d not a multiple of 8 (ring depth 1), lists of a dozen rows, k in two rounds
    (700, 32, 1000, 1, 5, 10),          # more lists than rows: most lists are empty
    (40_000, 96, 200, 1, 70, 10),       # three passes of <= 32 queries (ring depth 4), lists of 200 rows = 7 tiles
    (9_000, 768, 64, 0, 33, 120),       # the reference's dimension: k above one pass's 112 -> rounds; two passes
])
def test_hip_nprobe_search_matches_the_oracle(gpu_required, tmp_path, n, d, nlist, metric, nq, k):
    from rag_inference_pipeline_amd.flat_index import FlatIndex
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    X, path = _file(tmp_path, n, d, nlist, metric, seed=n + d)
    lists = index_io.read_ivfflat_lists(path)
    Q = _unit(np.random.default_rng(11), nq, d)
    Q[0] = X[123 % n]                                           # a query that is a corpus row
    idx = IVFFlatIndex(lists, nprobe=8)
    assert (idx.ntotal, idx.nlist, idx.nprobe) == (n, nlist, 8)
    for nprobe in (1, 8, 64, 100, 200):   # (the coarse selection keeps 64, 128 or 256 keys per wave)
        D, I = idx.search(Q, k, nprobe=nprobe)
        Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, nprobe, metric)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    if nlist <= 2048:   # every list probed == the exhaustive search, bit for bit
        flat = FlatIndex(d, metric)
        flat.add(X)
        Df, If = flat.search(Q, k)
        Da, Ia = idx.search(Q, k, nprobe=nlist)
        np.testing.assert_array_equal(Ia, If)
        np.testing.assert_array_equal(Da.view(np.uint32), Df.view(np.uint32))
        flat.close()
    # queries that share their lists (one row, asked 8 times, among others): a list is read once for all of them
    Qs = np.concatenate([np.repeat(X[7:8], 8, axis=0), Q[: max(1, nq // 2)]])
    D, I = idx.search(Qs, k, nprobe=5)
    Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Qs, k, 5, metric)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    with pytest.raises(Exception, match="2048"):
        idx.search(Q, 3000)
    idx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [1, 0])
def test_two_stage_list_scan_is_identical_and_certified(gpu_required, tmp_path, monkeypatch, metric):
    """The two-stage form (fp16 screening pass over the tile map -> canonical fp32 scores of the band -> certificate, exact
    scan as the fallback) is what k <= 100 searches run by default: same ids and score bits as the one-stage scan
    (RAG_AMD_IVF_TWO_STAGE=0) and the oracle; random data passes its certificates; 400 near-identical rows at the top of
    a query's ranking cannot be certified and are answered by the exact fallback — only that query; duplicates rank by
    ascending stored id."""
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    rng = np.random.default_rng(31)
    n, d, nlist = 60_000, 384, 96
    X = _unit(rng, n, d)
    centre = _unit(rng, 1, d)[0]
    clones = centre[None, :] + 1e-6 * rng.standard_normal((400, d)).astype(np.float32)
    clones /= np.linalg.norm(clones, axis=1, keepdims=True)
    X[rng.choice(n, size=400, replace=False)] = clones.astype(np.float32)
    X[[5, 77, 4000, 19_999]] = X[123]
    path = tmp_path / "two_stage.index"
    index_io.write_ivfflat_index(path, X, nlist, metric, seed=31, nprobe=8, all_centroids=True)
    lists = index_io.read_ivfflat_lists(path)
    idx = IVFFlatIndex(lists, nprobe=8)
    assert idx.two_stage
    Q = np.vstack([_unit(rng, 30, d), X[[123]], centre[None, :]]).astype(np.float32)
    for k, nprobe in ((10, 8), (100, 8), (10, 1), (48, 20)):
        idx.screen_stats(reset=True)
        D, I = idx.search(Q, k, nprobe=nprobe)
        st = idx.screen_stats()
        Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, nprobe, metric)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
        assert st["queries"] == len(Q) and 1 <= st["fallbacks"] <= 3 and st["max_err_ratio"] < 0.5, st
        monkeypatch.setenv("RAG_AMD_IVF_TWO_STAGE", "0")
        D1, I1 = idx.search(Q, k, nprobe=nprobe)
        monkeypatch.delenv("RAG_AMD_IVF_TWO_STAGE")
        assert idx.screen_stats()["queries"] == len(Q)          # the one-stage search certifies nothing
        np.testing.assert_array_equal(I, I1)
        np.testing.assert_array_equal(D.view(np.uint32), D1.view(np.uint32))
    D, I = idx.search(X[[123]], 8, nprobe=nlist)
    assert I[0, :5].tolist() == [5, 77, 123, 4000, 19_999]
    idx.close()
    monkeypatch.setenv("RAG_AMD_IVF_TWO_STAGE", "0")     # at load: no fp16 copy at all
    plain = IVFFlatIndex(lists, nprobe=8)
    monkeypatch.delenv("RAG_AMD_IVF_TWO_STAGE")
    assert not plain.two_stage
    D2, I2 = plain.search(Q, 10)
    Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, 10, 8, metric)
    np.testing.assert_array_equal(I2, Io)
    plain.close()


@pytest.mark.gpu
def test_randomised_ivf_searches_match_the_oracle(gpu_required):
    """30 seeded random draws of everything at once: dimension (ring depths 8 / 4 / 1, fp16 copies padded to 64), list count
    and lengths (empty lists, lists of one tile, one dominant list), metric of the rows and of the quantizer, k (every buffer
    size, above 100 -> one-stage, above one pass -> rounds), nprobe (all selection widths, > 256 -> the flat search as the
    quantizer), query counts over several passes, and the data the certificate finds hard — tight clusters, exact duplicates,
    coordinates on a coarse grid, a dominant coordinate.  Lists are ANY partition of the rows (the mode searches the lists it
    is given); ids a random permutation.  Ids and fp32 score bits against the oracle every time."""
    from rag_inference_pipeline_amd.index_io import IVFFlatLists
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    rng = np.random.default_rng(20261006)
    dims = [8, 24, 64, 100, 200, 384, 520, 768, 1024]
    fallbacks = queries = two_stage_runs = 0
    for trial in range(int(os.environ.get("RAG_AMD_TEST_TRIALS", "30"))):
        d = int(rng.choice(dims))
        n = int(rng.integers(1, 15_000 if d <= 384 else 6_000))
        nlist = int(rng.choice([1, 7, 64, 300, 2000]))
        nq = int(rng.integers(1, 70))
        k = int(rng.choice([1, 3, 10, 16, 17, 48, 49, 100, 130, 260]))
        nprobe = int(rng.choice([1, 4, 64, 100, 200, 400]))
        metric, qmetric = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        kind = trial % 5
        if kind == 0:
            X = rng.standard_normal((n, d), dtype=np.float32) * np.float32(rng.choice([1e-3, 1.0, 300.0]))
        elif kind == 1:
            centres = rng.standard_normal((max(1, n // 200), d), dtype=np.float32)
            X = centres[rng.integers(0, len(centres), size=n)] + np.float32(1e-3) * rng.standard_normal((n, d), dtype=np.float32)
        elif kind == 2:
            X = rng.standard_normal((n, d), dtype=np.float32)
            X[n // 2:] = X[: n - n // 2]
        elif kind == 3:
            X = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
        else:
            X = rng.standard_normal((n, d), dtype=np.float32)
            X[:, 0] *= 100.0
        X = np.ascontiguousarray(X, dtype=np.float32)
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        if kind == 1:
            Q = X[rng.integers(0, n, size=nq)] + np.float32(1e-2) * Q
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        cent = np.ascontiguousarray(X[rng.integers(0, n, size=nlist)] + np.float32(0.1) * rng.standard_normal((nlist, d), dtype=np.float32))
        weights = rng.random(nlist) ** (3 if trial % 2 else 1)       # every other draw: a few lists take most rows
        assign = np.sort(rng.choice(nlist, size=n, p=weights / weights.sum()))
        offsets = np.zeros(nlist + 1, dtype=np.int64)
        np.cumsum(np.bincount(assign, minlength=nlist), out=offsets[1:])
        ids = rng.permutation(n).astype(np.int64)
        lists = IVFFlatLists(cent, qmetric, X, ids, offsets, metric, nprobe)
        idx = IVFFlatIndex(lists)
        D, I = idx.search(Q, k)
        Do, Io = oracle.ivf_search(cent, qmetric, X, ids, offsets, Q, k, nprobe, metric)
        msg = f"trial {trial}: kind={kind} n={n} d={d} nlist={nlist} nq={nq} k={k} nprobe={nprobe} metric={metric}/{qmetric}"
        np.testing.assert_array_equal(I, Io, err_msg=msg)
        np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32), err_msg=msg)
        st = idx.screen_stats()
        assert st["max_err_ratio"] < 1.0, msg
        if idx.two_stage and k <= 100:
            two_stage_runs += 1
            fallbacks += st["fallbacks"]
            queries += st["queries"]
        idx.close()
    assert two_stage_runs >= 10 and queries > 0 and 0 < fallbacks < queries


@pytest.mark.gpu
def test_ivf_error_contract_and_edge_cases(gpu_required, tmp_path):
    """What rag_ivf_* refuses and what it answers at the edges: bad shapes, a closed handle, dimensions the mode does not
    take, queries with no reachable row (-1 / FLT_MAX padding, as rag_index_search pads), non-finite queries (no results
    under L2: every distance is inf or NaN), an empty batch."""
    from rag_inference_pipeline_amd import _native
    from rag_inference_pipeline_amd.index_io import IVFFlatLists
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    X, path = _file(tmp_path, 3000, 40, 30, 1, seed=9)
    lists = index_io.read_ivfflat_lists(path)
    idx = IVFFlatIndex(lists, nprobe=4)
    Q = _unit(np.random.default_rng(1), 3, 40)
    with pytest.raises(ValueError):
        idx.search(Q[:, :8], 5)
    for bad_k, bad_np in ((0, 4), (5, 0)):
        with pytest.raises(_native.RagAmdError):
            idx.search(Q, bad_k, nprobe=bad_np)
    D, I = idx.search(Q[:0], 5)
    assert D.shape == (0, 5) and I.shape == (0, 5)
    # a query with a NaN / an overflowing norm: no results (ids -1, distances FLT_MAX), the others unaffected
    Qb = Q.copy()
    Qb[1, 3] = np.nan
    Qb[2, :] = 3e19
    D, I = idx.search(Qb, 5)
    Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Qb, 5, 4, 1)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    assert (I[1:] == -1).all() and (I[0] >= 0).all()
    idx.close()
    with pytest.raises(RuntimeError):
        idx.search(Q, 5)
    # only empty lists near the query: k results cannot be reached -> padding
    cent = np.eye(4, 40, dtype=np.float32)
    rows = np.tile(cent[3], (3, 1)).astype(np.float32)
    sparse = IVFFlatLists(cent, 1, rows, np.array([7, 8, 9]), np.array([0, 0, 0, 0, 3]), 1, 1)
    idx = IVFFlatIndex(sparse, nprobe=1)
    D, I = idx.search(cent[:1], 4)           # nearest list (0) is empty
    assert (I == -1).all() and (D == np.finfo(np.float32).max).all()
    D, I = idx.search(cent[3:4], 4)          # list 3 holds three rows
    assert sorted(I[0, :3].tolist()) == [7, 8, 9] and I[0, 3] == -1
    idx.close()
    with pytest.raises(_native.RagAmdError, match="1024"):
        IVFFlatIndex(IVFFlatLists(np.zeros((2, 1032), np.float32), 1, np.zeros((0, 1032), np.float32), np.zeros(0, np.int64),
                                  np.zeros(3, np.int64), 1, 1))


@pytest.mark.gpu
def test_build_ivf_tool_on_the_gpu_and_search_what_it_built(gpu_required, tmp_path):
    """The whole path without faiss: a flat index file -> tools/build_ivf.py (rows assigned by this build's flat search over
    the trained centroids) -> FAISSStore in the nprobe mode.  A row's list is the first list its own vector probes, so every
    row finds itself with nprobe = 1; results equal the oracle's on the file the tool wrote."""
    from rag_inference_pipeline_amd.components.faiss_store import FAISSStore
    from rag_inference_pipeline_amd.config import PipelineSettings
    from rag_inference_pipeline_amd.tools.build_ivf import build_ivf
    rng = np.random.default_rng(9)
    n, d, nlist = 30_000, 96, 256
    centres = _unit(rng, 300, d)
    X = centres[rng.integers(0, 300, size=n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d)
    X = np.ascontiguousarray(X / np.linalg.norm(X, axis=1, keepdims=True), dtype=np.float32)
    flat = tmp_path / "corpus.npy"
    np.save(flat, X)
    out = tmp_path / "corpus.ivf"
    info = build_ivf(X, nlist, 1, str(out), nprobe=8, train_rows=10_000, iters=10)
    assert info["rows"] == n and info["empty_lists"] < nlist // 2
    lists = index_io.read_ivfflat_lists(out)
    store = FAISSStore(PipelineSettings(FAISS_INDEX_PATH=str(out), faiss_dim=d, RAG_AMD_IVF_MODE="nprobe", FAISS_NPROBE=1))
    store.load()
    probe = rng.choice(n, size=64, replace=False)
    D, I = store.search(X[probe], 1)
    np.testing.assert_array_equal(I[:, 0], probe)            # nprobe = 1: the row's own list is the one probed
    store.unload()
    store = FAISSStore(PipelineSettings(FAISS_INDEX_PATH=str(out), faiss_dim=d, RAG_AMD_IVF_MODE="nprobe", FAISS_NPROBE=8))
    store.load()
    Q = _unit(rng, 32, d)
    D, I = store.search(Q, 10)
    Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, 10, 8, 1)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    store.unload()


@pytest.mark.gpu
def test_concurrent_host_searches_share_one_handle(gpu_required, tmp_path):
    """The reference's scheduler runs batches on pool threads (gateway/batch_scheduler.py:286-288): four threads search one
    IVFFlatIndex at once, host pointers and device pointers mixed, shapes mixed — every result is the oracle's."""
    import threading
    import torch
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    n, d, nlist = 40_000, 192, 128
    X, path = _file(tmp_path, n, d, nlist, 1, seed=13)
    lists = index_io.read_ivfflat_lists(path)
    idx = IVFFlatIndex(lists, nprobe=6)
    errors = []

    def client(t: int) -> None:
        try:
            rng = np.random.default_rng(500 + t)
            stream = torch.cuda.Stream()
            for it in range(6):
                nq, k = ((32, 10), (3, 100), (1, 5), (40, 10))[(t + it) % 4]
                Q = _unit(rng, nq, d)
                if (t + it) % 2:
                    D, I = idx.search(Q, k)
                else:
                    with torch.cuda.stream(stream):
                        q = torch.from_numpy(Q).cuda()
                        s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
                        i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
                        idx.search_device(q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), stream.cuda_stream)
                    stream.synchronize()
                    D, I = s.cpu().numpy(), i.cpu().numpy()
                Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, 6, 1)
                if not (np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))):
                    errors.append(f"thread {t} round {it}: differs")
        except Exception as exc:  # noqa: BLE001
            errors.append(f"thread {t}: {type(exc).__name__}: {exc}")

    pool = [threading.Thread(target=client, args=(t,)) for t in range(4)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    idx.close()
    assert not errors, errors


@pytest.mark.gpu
def test_hip_nprobe_search_with_the_most_lists_the_mode_takes(gpu_required):
    """nlist = 32768 (the plan kernel's 128 KB of list masks in LDS): lists of a row or two, most of them one tile; one more
    list is refused."""
    from rag_inference_pipeline_amd import _native
    from rag_inference_pipeline_amd.index_io import IVFFlatLists
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    rng = np.random.default_rng(41)
    nlist, n, d = 32768, 50_000, 24
    cent = _unit(rng, nlist, d)
    rows = _unit(rng, n, d)
    assign = np.sort(rng.integers(0, nlist, size=n))
    offsets = np.zeros(nlist + 1, dtype=np.int64)
    np.cumsum(np.bincount(assign, minlength=nlist), out=offsets[1:])
    ids = rng.permutation(n).astype(np.int64)
    lists = IVFFlatLists(cent, 1, rows, ids, offsets, 1, 64)
    idx = IVFFlatIndex(lists)
    Q = _unit(rng, 7, d)
    for nprobe in (64, 300):
        D, I = idx.search(Q, 10, nprobe=nprobe)
        Do, Io = oracle.ivf_search(cent, 1, rows, ids, offsets, Q, 10, nprobe, 1)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    idx.close()
    with pytest.raises(_native.RagAmdError, match="32768"):
        IVFFlatIndex(IVFFlatLists(_unit(rng, nlist + 1, d), 1, rows[:1], ids[:1], np.r_[np.zeros(nlist + 1, np.int64), 1], 1, 1))


@pytest.mark.gpu
def test_hip_nprobe_search_on_device_pointers(gpu_required, tmp_path):
    """rag_ivf_search_device: queries and results in device memory, enqueued on the caller's stream — the same bits as
    the host-pointer entry point, batch after batch without a host wait, alternating between two streams (the ticket counter
    is left at zero by every scan; a search on another stream first waits for the previous one's hand-over event)."""
    import torch
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    n, d, nlist, k = 30_000, 128, 512, 10
    X, path = _file(tmp_path, n, d, nlist, 1, seed=5)
    lists = index_io.read_ivfflat_lists(path)
    idx = IVFFlatIndex(lists, nprobe=16)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]   # two callers: the handle's workspace is handed over by an event
    outs = []
    for rep, nq in enumerate((32, 1, 8, 32, 32, 5)):
        stream = streams[rep % 2]
        with torch.cuda.stream(stream):
            Q = _unit(np.random.default_rng(100 + rep), nq, d)
            q = torch.from_numpy(Q).cuda()
            s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
            i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
            idx.search_device(q.data_ptr(), nq, k, s.data_ptr(), i.data_ptr(), stream.cuda_stream)
            outs.append((Q, q, s, i))
    for stream in streams:
        stream.synchronize()
    for Q, _, s, i in outs:
        Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, 16, 1)
        np.testing.assert_array_equal(i.cpu().numpy(), Io)
        np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), Do.view(np.uint32))
    idx.close()


@pytest.mark.gpu
def test_faiss_store_nprobe_mode_end_to_end(gpu_required, tmp_path):
    from rag_inference_pipeline_amd.components.faiss_store import FAISSStore
    from rag_inference_pipeline_amd.config import PipelineSettings
    n, d, nlist = 30_000, 96, 512
    X, path = _file(tmp_path, n, d, nlist, 1, seed=2, nprobe=64)
    lists = index_io.read_ivfflat_lists(path)
    Q = _unit(np.random.default_rng(4), 32, d)
    store = FAISSStore(PipelineSettings(FAISS_INDEX_PATH=str(path), faiss_dim=d, RAG_AMD_IVF_MODE="nprobe", FAISS_NPROBE=4))
    store.load()
    assert store.is_loaded and store.index_size == n
    D, I = store.search(Q, 10)
    Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, 10, 4, 1)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    # the embedder's device hand-off: queries that never left HBM, on the producer's stream (rag_ivf_search_device_host_out)
    import torch
    from rag_inference_pipeline_amd.device_embeddings import DeviceEmbeddings
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        qt = torch.from_numpy(Q).cuda()
    Dd, Id = store.search(DeviceEmbeddings(qt, stream.cuda_stream, 0), 10)
    np.testing.assert_array_equal(Id, Io)
    np.testing.assert_array_equal(Dd.view(np.uint32), Do.view(np.uint32))
    store.unload()
    exhaustive = FAISSStore(PipelineSettings(FAISS_INDEX_PATH=str(path), faiss_dim=d))   # the default: every list
    exhaustive.load()
    De, Ie = exhaustive.search(Q, 10)
    Df, If = oracle.search(X, Q, 10, 1)
    np.testing.assert_array_equal(Ie, If)
    assert (I != Ie).any()          # nprobe = 4 of 512 lists misses true neighbours: the two modes differ, as documented
    exhaustive.unload()


@pytest.mark.gpu
def test_hip_nprobe_search_at_the_generators_full_size(gpu_required):
    """The generator's shape at full size (reference scripts/create_test_docs.py:12,83-104: 4.5M x 768, L2, nlist 4096,
    nprobe 64) on a clustered corpus built on the GPU (scripts/bench_ivf.py: setup only): 32 + 1 queries against the
    oracle — ids and fp32 score bits — and the size-independent properties: every result row lies in one of the query's
    probed lists, distances ascend, a query that IS a corpus row finds it first at distance ~0."""
    import argparse
    import torch
    import scripts.bench_ivf as bivf
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex
    if torch.cuda.mem_get_info()[1] < 100e9:
        pytest.skip("needs an MI355X-class device (the 13.8 GB corpus is built on the GPU)")
    n, d, nlist, nprobe, k = 4_500_000, 768, 4096, 64, 10
    lists, cent = bivf.build_lists(n, d, nlist, nprobe, unit=True, clustered=True)
    idx = IVFFlatIndex(lists, nprobe=nprobe)
    g = torch.Generator(device="cuda").manual_seed(5)
    Q = bivf.draw((33, d), g, True).cpu().numpy()
    Q[32] = lists.rows[1234567]
    D, I = idx.search(Q, k)
    idx.close()
    bivf._CENTERS = None
    del cent
    torch.cuda.empty_cache()
    Do, Io = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, nprobe, 1)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    assert (np.diff(D, axis=1) >= 0).all() and (I >= 0).all()
    assert I[32, 0] == lists.ids[1234567] and abs(D[32, 0]) < 1e-5
    # membership: the list of every returned row is among the 64 lists nearest to the query
    pos = np.empty(n, dtype=np.int64)
    pos[lists.ids] = np.arange(n)
    list_of = np.searchsorted(lists.offsets, pos[I], side="right") - 1
    cn = (lists.centroids.astype(np.float64) ** 2).sum(1)
    dist = cn[None, :] - 2.0 * Q.astype(np.float64) @ lists.centroids.astype(np.float64).T
    kth = np.sort(dist, axis=1)[:, nprobe - 1]
    assert (np.take_along_axis(dist, list_of, axis=1) <= kth[:, None] + 1e-6).all()
