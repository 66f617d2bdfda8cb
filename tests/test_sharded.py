"""Multi-rank corpus sharding: world_size-2/3 gloo runs, CPU rehearsal and (gpu) the real kernels."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import flat as oracle
from rag_inference_pipeline_amd.sharded import pack_layout, shard_range

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, world, tmp_path, n, d, nq, k, metric, env=None):
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"rank{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_sharded_worker.py"), mode, str(r), str(world), str(port), out,
             str(n), str(d), str(nq), str(k), str(metric)],
            env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **(env or {}))))
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [np.load(o) for o in outs]


def _check(results, n, d, nq, k, metric, world):
    X = oracle.synth_rows(1234, 0, n, d)
    Q = oracle.synth_rows(4321, 0, nq, d)
    D, I = oracle.search(X, Q, k, metric)
    D2, I2 = oracle.search(X, Q[: max(1, nq // 2)], k, metric)
    covered = 0
    for r, res in enumerate(results):
        np.testing.assert_array_equal(res["I"], I)       # every rank holds the full merged answer
        np.testing.assert_array_equal(res["D"], D)
        np.testing.assert_array_equal(res["I2"], I2)
        np.testing.assert_array_equal(res["D2"], D2)
        assert (int(res["lo"]), int(res["hi"])) == shard_range(n, r, world)
        covered += int(res["hi"]) - int(res["lo"])
    assert covered == n


def test_shard_ranges_partition_the_corpus():
    for n, w in [(10_000_000, 8), (10, 3), (7, 8), (0, 2)]:
        edges = [shard_range(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        sizes = [hi - lo for lo, hi in edges]
        assert max(sizes) - min(sizes) <= 1
    # 3.84 KB of ids and scores per rank per batch + the "not final" word, padded to 8 bytes
    assert pack_layout(32, 10) == (2560, 3840, 3848)
    assert pack_layout(1, 1) == (8, 12, 16) and pack_layout(3, 5)[2] % 8 == 0


@pytest.mark.parametrize("world,metric", [(2, 0), (3, 1)])
def test_sharded_search_gloo_cpu_rehearsal(tmp_path, world, metric):
    n, d, nq, k = 5001, 64, 9, 10
    _check(_launch("cpu", world, tmp_path, n, d, nq, k, metric), n, d, nq, k, metric, world)


def test_flagged_batches_are_repeated_through_the_exact_scan_on_every_rank(tmp_path):
    """The "not final" word of a rank's deferred two-stage search travels in the all-gather: when the last rank
    raises it, every rank sees the same OR-ed word, repeats the batch in RAG_SEARCH_EXACT_ONE_PASS mode and
    gathers again — with two searches in flight (submit / collect) as well as one at a time."""
    world, n, d, nq, k, metric = 3, 4001, 48, 7, 10, 0
    results = _launch("cpuflag", world, tmp_path, n, d, nq, k, metric)
    _check(results, n, d, nq, k, metric, world)   # the later, unflagged searches
    X, Q = oracle.synth_rows(1234, 0, n, d), oracle.synth_rows(4321, 0, nq, d)
    Da, Ia = oracle.search(X, Q, k, metric)
    Db, Ib = oracle.search(X, Q[::-1].copy(), k, metric)
    for res in results:
        np.testing.assert_array_equal(res["Ia"], Ia)
        np.testing.assert_array_equal(res["Da"], Da)
        np.testing.assert_array_equal(res["Ib"], Ib)
        np.testing.assert_array_equal(res["Db"], Db)
        assert int(res["repeats_pipelined"]) == 2 and int(res["repeats_total"]) == 2   # exactly the flagged ones


@pytest.mark.parametrize("flagging", [False, True])
def test_serving_channel_gloo_cpu_rehearsal(tmp_path, flagging):
    """Leader / follower protocol on CPU (world 3, gloo): searches and query-sharded rerank passes
    interleave on the channel, the sharded rerank equals the one-process result exactly (stub model),
    shutdown releases the followers.  flagging: the last rank's two-stage searches report "not final" — the leader
    re-sends each such batch as an OP_SEARCH_EXACT request (followers never look at a flag) and the answers are the
    exact ones."""
    world, n, d = 3, 3001, 32
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"serve{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_sharded_serve_worker.py"), str(r), str(world), str(port), out, "0",
             "flag" if flagging else "plain"],
            env=dict(os.environ, OMP_NUM_THREADS="2")))
    for p in procs:
        assert p.wait(timeout=300) == 0
    lead = np.load(outs[0])
    assert bool(lead["same"]) and int(lead["repeats"]) == (2 if flagging else 0)
    X, Q = oracle.synth_rows(1234, 0, n, d), oracle.synth_rows(4321, 0, 6, d)
    D0, I0 = oracle.search(X, Q, 5)
    D1, I1 = oracle.search(X, Q[:2], 3)
    np.testing.assert_array_equal(lead["I0"], I0)
    np.testing.assert_array_equal(lead["D0"], D0)
    np.testing.assert_array_equal(lead["I1"], I1)
    np.testing.assert_array_equal(lead["D1"], D1)
    assert all(int(np.load(o)["served"]) == (5 if flagging else 3) for o in outs[1:])      # search, rerank, search (+ 2 repeats)


@pytest.mark.parametrize("flagging", [False, True])
def test_concurrent_batches_on_the_serving_channel(tmp_path, flagging):
    """The scheduler runs batches concurrently on pool threads (reference batch_scheduler.py:286-288,
    retrieval/api.py:348-349): four threads on rank 0 issue searches (same and different shapes) and
    query-sharded rerank passes at once.  Each request's collectives must stay together — every answer
    equals the oracle's, nothing hangs, and the followers serve exactly the number of requests issued."""
    world, threads = 3, 4
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"conc{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_sharded_serve_worker.py"), str(r), str(world), str(port), out,
             str(threads), "flag" if flagging else "plain"], env=dict(os.environ, OMP_NUM_THREADS="2")))
    try:
        for p in procs:
            assert p.wait(timeout=120) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    lead = np.load(outs[0])
    assert lead["errors"].size == 0, lead["errors"].tolist()
    assert (int(lead["repeats"]) > 0) == flagging
    assert all(int(np.load(o)["served"]) == int(lead["requests"]) for o in outs[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("world,metric,k", [(2, 0, 10), (3, 0, 100), (2, 1, 10)])
def test_sharded_search_real_kernels_two_ranks_one_gpu(gpu_required, tmp_path, world, metric, k):
    n, d, nq = 40_003, 384, 32
    _check(_launch("gpu", world, tmp_path, n, d, nq, k, metric), n, d, nq, k, metric, world)


def _nccl_worlds():
    """World sizes the RCCL test runs at: 1 always (the whole nccl code path on the one-GPU box), plus one
    rank per GPU when the box has more (capped at 4: RCCL needs a GPU per rank, and the box allows few
    processes on a card)."""
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 0
    return [1] + ([min(n, 4)] if n > 1 else [])


@pytest.mark.gpu
@pytest.mark.parametrize("world", _nccl_worlds())
@pytest.mark.parametrize("metric,k,own", [(0, 10, "1"), (1, 100, "1"), (0, 10, "0")])
def test_sharded_search_over_rccl(gpu_required, tmp_path, world, metric, k, own):
    """BASELINE config D's collective leg: backend "nccl".  own = "1" (the product): the whole step — local search,
    ncclAllGather on the library's own communicator, flagged merge — is ONE C-ABI call on one stream
    (rag_index_search_gather_device), the request broadcast another (rag_comm_request_device) with the followers
    polling the posted head; own = "0": torch.distributed's collectives in the same places.  Either way ids and score
    bits equal the oracle's on every rank — one-pass and two-stage, two batches in flight, with the collective on
    its own stream, and through the leader/follower protocol."""
    n, d, nq = 60_007, 768, 32
    results = _launch("nccl", world, tmp_path, n, d, nq, k, metric, env={"RAG_AMD_OWN_RCCL": own})
    _check(results, n, d, nq, k, metric, world)
    lead = results[0]
    assert int(lead["own"]) == int(own)
    np.testing.assert_array_equal(lead["I3"], lead["I"])
    np.testing.assert_array_equal(lead["D3"], lead["D"])
    np.testing.assert_array_equal(lead["I8"], lead["I"][:3])
    np.testing.assert_array_equal(lead["D8"], lead["D"][:3])
    Qr = np.ascontiguousarray(oracle.synth_rows(4321, 0, nq, d)[::-1])
    Dr, Ir = oracle.search(oracle.synth_rows(1234, 0, n, d), Qr, k, metric)
    for res in results:   # two-stage, two in flight; then the same with all-gather + merge on a second stream
        for a, b in (("4", "5"), ("6", "7")):
            np.testing.assert_array_equal(res["I" + a], lead["I"])
            np.testing.assert_array_equal(res["D" + a], lead["D"])
            np.testing.assert_array_equal(res["I" + b], Ir)
            np.testing.assert_array_equal(res["D" + b], Dr)
    assert all(int(r["served"]) == 2 for r in results[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("world", _nccl_worlds())
@pytest.mark.parametrize("metric,k,own", [(1, 10, "1"), (0, 100, "1"), (1, 10, "0")])
def test_sharded_ivf_search_over_rccl(gpu_required, tmp_path, world, metric, k, own):
    """A sharded IVFFlat index in the nprobe mode over backend "nccl": every rank holds its share of every list; own = "1":
    the step — local nprobe search (two-stage), ncclAllGather on the library's own communicator, merge — is ONE C-ABI
    call (rag_ivf_search_gather_device), also with the collective on a second stream and through the leader / follower
    protocol; own = "0": torch's collectives.  The unsharded oracle's ids and score bits either way."""
    sys.path.insert(0, HERE)
    from _sharded_ivf_worker import build_lists
    n, d, nlist, nprobe = 40_000, 128, 200, 9
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"ivfrccl{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_sharded_ivf_worker.py"), str(r), str(world), str(port), out, str(n), str(d), str(nlist),
             str(nprobe), str(k), str(metric)],
            env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", RAG_AMD_OWN_RCCL=own)))
    for p in procs:
        assert p.wait(timeout=300) == 0
    lists, Q = build_lists(n, d, nlist, metric, nprobe)
    Dw, Iw = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets, Q, k, nprobe, metric)
    Dr, Ir = Dw[::-1], Iw[::-1]
    for o in outs:
        res = np.load(o)
        assert int(res["own"]) == int(own) and int(res["two_stage"]) == 1
        for a, (Dx, Ix) in (("", (Dw, Iw)), ("4", (Dw, Iw)), ("5", (Dr, Ir))):
            np.testing.assert_array_equal(res["I" + a], Ix)
            np.testing.assert_array_equal(res["D" + a], Dx)
    lead = np.load(outs[0])
    np.testing.assert_array_equal(lead["I3"], Iw[:5])
    np.testing.assert_array_equal(lead["D3"], Dw[:5])
    assert all(int(np.load(o)["served"]) == 1 for o in outs[1:])


@pytest.mark.gpu
def test_faiss_store_sharded_serving_mode(gpu_required, tmp_path):
    """FAISSStore under torch.distributed: every rank loads its row range of the same file, rank 0
    serves search() as usual, the other ranks follow until it unloads."""
    from rag_inference_pipeline_amd import index_io
    n, d, world = 30_001, 384, 3
    X = oracle.synth_rows(1234, 0, n, d)
    path = tmp_path / "faiss_index.bin"
    index_io.write_flat_index(path, X, 0)
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"store{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_sharded_store_worker.py"), str(r), str(world), str(port), str(path), out, str(d)],
            env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")))
    for p in procs:
        assert p.wait(timeout=300) == 0
    lead = np.load(outs[0])
    for i, (nq, k) in enumerate([(1, 10), (32, 10), (5, 100)]):
        D, I = oracle.search(X, oracle.synth_rows(4321 + i, 0, nq, d), k)
        np.testing.assert_array_equal(lead[f"I{i}"], I)
        np.testing.assert_array_equal(lead[f"D{i}"], D)
    assert all(int(np.load(o)["served"]) == 3 for o in outs[1:])


@pytest.mark.gpu
def test_faiss_store_sharded_serving_in_the_ivf_nprobe_mode(gpu_required, tmp_path):
    """RAG_AMD_IVF_MODE=nprobe under torch.distributed: every rank keeps the centroids and its share of EVERY inverted list,
    probes the same lists, and rank 0's search() returns the unsharded nprobe result — the oracle's, bit for bit."""
    from rag_inference_pipeline_amd import index_io
    n, d, world, nlist, nprobe = 20_000, 96, 3, 128, 6
    rng = np.random.default_rng(17)
    X = rng.standard_normal((n, d), dtype=np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    path = tmp_path / "ivf_index.bin"
    index_io.write_ivfflat_index(path, X, nlist, 1, seed=17, nprobe=nprobe, all_centroids=True)
    lists = index_io.read_ivfflat_lists(path)
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"ivfstore{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_sharded_store_worker.py"), str(r), str(world), str(port), str(path), out, str(d),
             f"nprobe:{nprobe}"],
            env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")))
    for p in procs:
        assert p.wait(timeout=300) == 0
    lead = np.load(outs[0])
    for i, (nq, k) in enumerate([(1, 10), (32, 10), (5, 100)]):
        D, I = oracle.ivf_search(lists.centroids, lists.quantizer_metric, lists.rows, lists.ids, lists.offsets,
                                 oracle.synth_rows(4321 + i, 0, nq, d), k, nprobe, 1)
        np.testing.assert_array_equal(lead[f"I{i}"], I)
        np.testing.assert_array_equal(lead[f"D{i}"], D)
    assert all(int(np.load(o)["served"]) == 3 for o in outs[1:])


@pytest.mark.gpu
def test_reranker_shards_batches_by_query(gpu_required, tmp_path):
    """SURVEY §8e, rerank row: with one process per GPU the rerank batch is split by query over the
    ranks that shard the index; order and scores (to fp32 rounding) equal the one-process result, and rerank requests
    interleave with searches on the same serving channel."""
    from rag_inference_pipeline_amd import index_io
    n, d, world = 5_000, 64, 3
    path = tmp_path / "faiss_index.bin"
    index_io.write_flat_index(path, oracle.synth_rows(1234, 0, n, d), 0)
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"rr{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_sharded_rerank_worker.py"), str(r), str(world), str(port), str(path), out, str(d)],
            env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")))
    for p in procs:
        assert p.wait(timeout=300) == 0
    lead = np.load(outs[0])
    assert bool(lead["same_order"]) and float(lead["max_diff"]) < 1e-5, (lead["same_order"], lead["max_diff"])
    assert bool(lead["search_same"]) and int(lead["n_lists"]) == 7
    assert all(s <= 5 for s in lead["sizes"])
    assert all(int(np.load(o)["served"]) == 3 for o in outs[1:])      # search, rerank, search
