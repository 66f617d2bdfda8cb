"""Headline benchmark: queries/sec of the retrieval hot path on a synthetic 10M x 768 corpus,
batch 32, k = 10 (BASELINE.json `metric`), on N GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one batch of 32 query embeddings (already resident in HBM) through
scan + top-k on this rank's corpus shard, then (N > 1) one RCCL all-gather of the per-shard
top-k lists and a device merge.  The corpus is split over the ranks (strong scaling: 10M rows
in total whatever N is), as the north star defines the workload.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6.3 TB/s
TRAFFIC_FILE = "r04_scan_hbm_traffic.json"  # PMC pass of this workload (profiles/README.md says how it was collected)

# xor of the returned ids for the default workload (rows, dim, batch, k, seed), as produced by the
# 1-GPU run that the full-size parity test checks against the oracle; sharded runs must reproduce it
EXPECTED_ID_CHECKSUM = {(10_000_000, 768, 32, 10, 1234): 5031347}


def parse_args() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total corpus rows over all ranks")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-steps", type=int, default=20)
    ap.add_argument("--no-encoder-leg", action="store_true",
                    help="skip the extra text->ids leg (query encoder in front of the scan)")
    ap.add_argument("--no-rerank-leg", action="store_true",
                    help="skip the extra cross-encoder stage leg (32 x 100 synthetic (query, document) pairs)")
    ap.add_argument("--no-two-stage-leg", action="store_true",
                    help="skip the extra leg through the two-stage exact search (fp16 screen + fp32 second stage)")
    ap.add_argument("--no-ivf-leg", action="store_true",
                    help="skip the extra leg through the IVFFlat nprobe mode (4.5M x 768, nlist 4096, nprobe 64)")
    return ap.parse_args()


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_cpu(fn, budget_s: float, max_reps: int) -> tuple[int, float]:
    fn()  # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= max_reps:
            return reps, el


def cpu_baseline(args: argparse.Namespace) -> dict:
    """The CPU path timed on this box's host cores on a bounded sample of the same workload; queries/sec
    scaled linearly to the full corpus.  `import faiss` is PROBED: when it imports, the real
    IndexFlatIP.search is the baseline (`kind: "reference"`) and the oracle's figure rides along; when it
    does not, the baseline is the oracle — a C/OpenMP port of the IndexFlat algorithm (`kind: "port"`) —
    and the line says why."""
    from oracle import flat as oracle

    n = min(args.cpu_sample_rows, args.rows)
    # thread policy of the reference: faiss_threads = min(16, cores) (config/__init__.py:32-46,
    # applied at runtime.py:76); cores = what this process may actually run on
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(16, avail, oracle.num_threads()))
    X = oracle.synth_rows(args.seed, 0, n, args.dim)
    Q = oracle.synth_rows(args.seed + 3087, 0, args.batch, args.dim)
    scale = n / float(args.rows)
    reps, el = _time_cpu(lambda: oracle.search(X, Q, args.k, nthreads=cores), 8.0, 20)
    port = {"value": args.batch * reps / el * scale, "cores": cores, "ms_per_batch_on_sample": el / reps * 1e3}
    out = {
        "value": port["value"],
        "unit": "queries/s",
        "cores": cores,
        "kind": "port",
        "cpu_model": _cpu_model(),
        "logical_cpus_visible": avail,
        "sample": f"oracle/flat_oracle.c (AVX2+FMA, OpenMP x{cores}) on the first {n} of {args.rows} rows, "
                  f"{reps} batches of {args.batch} at {el / reps * 1e3:.1f} ms/batch; "
                  f"queries/s scaled x{scale:g} (scan cost is linear in rows)",
    }
    all_threads = min(avail, oracle.num_threads())
    if all_threads > cores:  # every core the process may use, next to the reference's 16-thread policy
        r2, e2 = _time_cpu(lambda: oracle.search(X, Q, args.k, nthreads=all_threads), 6.0, 20)
        out["all_cores"] = {"value": args.batch * r2 / e2 * scale, "unit": "queries/s", "cores": all_threads,
                            "ms_per_batch_on_sample": e2 / r2 * 1e3}
    # The same sample through an optimised BLAS: numpy's sgemm (OpenBLAS) on the same 16 threads + argpartition / sort
    # top-k — the algorithm family faiss IndexFlat uses for batches of 20 queries and more.  Reported beside the port
    # (which fixes a summation order for bit-exact parity and is slower for it); neither is the optimisation target.
    try:
        from threadpoolctl import threadpool_limits

        def sgemm_search():
            S = Q @ X.T
            part = np.argpartition(-S, args.k - 1, axis=1)[:, :args.k]
            top = np.take_along_axis(S, part, axis=1)
            order = np.argsort(-top, axis=1, kind="stable")
            return np.take_along_axis(part, order, axis=1)

        with threadpool_limits(limits=cores):
            r4, e4 = _time_cpu(sgemm_search, 5.0, 20)
        out["sgemm"] = {"value": args.batch * r4 / e4 * scale, "unit": "queries/s", "cores": cores,
                        "ms_per_batch_on_sample": e4 / r4 * 1e3,
                        "how": "numpy (OpenBLAS sgemm) Q @ X.T + argpartition + sort on the same sample and thread count"}
    except Exception as exc:  # noqa: BLE001
        out["sgemm"] = f"not measured ({type(exc).__name__}: {exc})"
    try:
        import faiss  # noqa: F401  (probe: the GPU box receives only this repo, faiss may or may not be there)
    except Exception as exc:  # noqa: BLE001
        out["faiss"] = f"not importable on this box ({type(exc).__name__}): baseline is the port"
    else:
        faiss.omp_set_num_threads(cores)  # reference runtime.py:76
        fidx = faiss.IndexFlatIP(args.dim)
        fidx.add(X)
        r3, e3 = _time_cpu(lambda: fidx.search(Q, args.k), 8.0, 20)
        out.update({"value": args.batch * r3 / e3 * scale, "kind": "reference",
                    "sample": f"faiss {faiss.__version__} IndexFlatIP.search, {cores} OpenMP threads, on the first {n} of "
                              f"{args.rows} rows, {r3} batches at {e3 / r3 * 1e3:.1f} ms/batch; scaled x{scale:g}",
                    "port": port, "faiss": faiss.__version__})
    return out


# ---- MFMA rooflines of the transformer legs ------------------------------------------------------------------------------
# achieved TF/s is computed live (GEMM + attention flops the pass executes / measured time); the matrix-pipe busy
# fraction and the HBM floor come from counter passes of the same workload (separate rocprofv3 --pmc runs, as the guide
# prescribes) committed under profiles/ — the `*_source` fields name the file, nothing is collected inside this run.
FP32_MFMA_PEAK_TF = 157.3            # dense v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TF = 2500.0            # dense fp16 / bf16 (the 5 PF headline figure includes 2:1 sparsity)


def _profile_json(name: str):
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def transformer_flops(hidden: int, inter: int, layers: int, lens, head: bool, first_only: bool) -> float:
    """GEMM + attention flops a forward pass over sequences of `lens` tokens executes (rag_bert.hip: with an output
    that reads first tokens only, the last layer's projections after attention run on one row per sequence)."""
    T, nseq = float(sum(int(x) for x in lens)), float(len(lens))
    per_token = 2.0 * (3 * hidden * hidden + hidden * hidden + 2 * hidden * inter)
    total = per_token * layers * T + (2.0 * hidden * hidden * nseq if head else 0.0)
    if first_only and nseq < T:
        total -= 2.0 * (hidden * hidden + 2 * hidden * inter) * (T - nseq)
    return total + sum(4.0 * int(L) * int(L) * hidden for L in lens) * layers


def mfma_roofline(flops: float, ms: float, peak_tf: float, basis: str, kernel: str, busy_file: str | None, busy_key=None,
                  floor_file: str | None = None) -> dict:
    achieved = flops / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak_tf, "unit": "TFLOP/s",
           "frac": achieved / peak_tf, "peak_basis": basis, "flops_per_pass": flops}
    prof = _profile_json(busy_file) if busy_file else None
    busy = None
    if prof is not None:
        try:
            busy = busy_key(prof) if busy_key else prof["derived"]["mfma_utilisation"]
        except (KeyError, TypeError, IndexError):
            busy = None
    out["mfma_busy"] = busy
    out["mfma_busy_source"] = (f"profiles/{busy_file}: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x GRBM_GUI_ACTIVE) from a separate "
                               "rocprofv3 --pmc pass of this workload; not collected inside this run") if busy is not None else None
    fl = _profile_json(floor_file) if floor_file else None
    try:
        out["hbm_floor_ms"] = fl["per_pass"]["hbm_floor_ms_at_6.3TBps"] if fl else None
    except (KeyError, TypeError):
        out["hbm_floor_ms"] = None
    out["hbm_floor_source"] = (f"profiles/{floor_file}: (FETCH_SIZE x 2 + WRITE_SIZE) of one pass / 6.3 TB/s (the guide's "
                               "achievable HBM rate)") if out["hbm_floor_ms"] is not None else None
    return out


def main() -> None:
    args = parse_args()
    # Exactly ONE line on stdout, the JSON: libraries write to file descriptor 1 on their own (RCCL prints a version
    # banner when its first communicator comes up, gloo its rank lines), so everything but the result goes to stderr:
    # fd 1 is pointed at stderr for the run and the line is written to the original stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the retrieval path has no CPU fallback)")
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsal) ranks share devices
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RAG_AMD_DIST_BACKEND", "nccl")  # "gloo": host-staged rehearsal
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend)

    from rag_inference_pipeline_amd.flat_index import FlatIndex
    from rag_inference_pipeline_amd.sharded import ShardedFlatIndex, shard_range
    from oracle import flat as oracle  # query generator + cpu_baseline only (checker side)

    N, d, B, k = args.rows, args.dim, args.batch, args.k
    row_lo, row_hi = shard_range(N, rank, world)
    n_local = row_hi - row_lo

    index = FlatIndex(d, device=dev)
    index.reserve(n_local)
    index.add_synthetic(n_local, seed=args.seed, row_number_offset=row_lo)
    index.set_id_offset(row_lo)
    sharded = ShardedFlatIndex(index, metric=0, device=dev) if world > 1 else None

    sptr = torch.cuda.current_stream().cuda_stream
    Q = torch.from_numpy(oracle.synth_rows(args.seed + 3087, 0, B, d)).cuda()
    out_s = torch.empty((B, k), dtype=torch.float32, device="cuda")
    out_i = torch.empty((B, k), dtype=torch.int64, device="cuda")
    fin = {}

    # N > 1: two batches in flight (submit / collect).  A rank's two-stage search defers its fp32 fallback: the
    # "not final" word rides in the all-gather and the host reads it back one batch later, while the next batch
    # is already queued on the GPU (sharded.py); every batch is collected — checked, repeated if flagged —
    # inside the timed region (barrier() drains).
    pend: list = []

    def submit(q) -> None:
        pend.append(sharded.submit(q, k))
        if len(pend) > 1:
            fin["s"], fin["i"] = sharded.collect(pend.pop(0))

    def drain() -> None:
        while pend:
            fin["s"], fin["i"] = sharded.collect(pend.pop(0))

    def step() -> None:
        if sharded is None:
            index.search_device(Q.data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr(), sptr)
        else:
            submit(Q)

    def barrier() -> None:
        drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    index.profile_enable(True)
    index.profile(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    scan_ms_total, scan_launches = index.profile(reset=True)
    index.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # p50 of per-batch latency: submit -> results visible to the host, one batch at a time
    lat = []
    for _ in range(args.latency_steps):
        barrier()
        t1 = time.perf_counter()
        step()
        drain()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    p50_ms = float(np.median(lat) * 1e3) if lat else None

    res_s = (fin["s"] if world > 1 else out_s).cpu().numpy()
    res_i = (fin["i"] if world > 1 else out_i).cpu().numpy()

    # SURVEY 8(d) timing protocol, literally: host submit -> results on host.  One batch of host-resident
    # queries through rag_index_search (pinned staging, H2D of B*d*4 bytes, scan + merge, D2H of 12*B*k bytes,
    # stream sync) per step; p50 over the steps.  PCIe-inclusive, so never `value`.
    def host_leg() -> dict | None:
        if world > 1:
            return serving_leg()
        Qh = np.ascontiguousarray(Q.cpu().numpy())
        for _ in range(max(2, args.warmup)):
            index.search(Qh, k)
        per = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            t1 = time.perf_counter()
            Dh, Ih = index.search(Qh, k)
            per.append(time.perf_counter() - t1)
        el = time.perf_counter() - t0
        return {"value": B * args.steps / el, "unit": "queries/s", "ms_per_step": el / args.steps * 1e3,
                "p50_latency_ms": float(np.median(per) * 1e3), "p95_latency_ms": float(np.percentile(per, 95) * 1e3),
                "steps": args.steps,
                "identical_to_device_path": bool(np.array_equal(Ih, res_i) and np.array_equal(Dh.view(np.uint32), res_s.view(np.uint32))),
                "path": f"rag_index_search: host queries ({B * d * 4} B H2D) -> results on host ({12 * B * k} B D2H), one sync per batch"}

    # N > 1: the step the PRODUCT serves (FAISSStore.search on rank 0 -> ShardedFlatIndex.leader_search, the other
    # ranks in follower_loop): host queries -> one pinned message -> one upload -> ONE broadcast -> local search
    # on the query device pointer inside the message -> all-gather -> merge -> ids, scores and the flag word
    # back in one copy.  Never `value` (PCIe- and host-inclusive).
    def serving_leg() -> dict | None:
        Qh = np.ascontiguousarray(Q.cpu().numpy())
        barrier()
        out = None
        if rank == 0:
            for _ in range(max(2, args.warmup)):
                sharded.leader_search(Qh, k)
            per = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                t1 = time.perf_counter()
                Dh, Ih = sharded.leader_search(Qh, k)
                per.append(time.perf_counter() - t1)
            el = time.perf_counter() - t0
            # the scheduler's situation (reference batch_scheduler.py:286-288): two batches in flight on pool threads of
            # rank 0, each calling FAISSStore.search -> leader_search back to back
            import threading
            per_thread = max(1, args.steps // 2)

            def client() -> None:
                for _ in range(per_thread):
                    sharded.leader_search(Qh, k)

            pool = [threading.Thread(target=client) for _ in range(2)]
            t2 = time.perf_counter()
            for th in pool:
                th.start()
            for th in pool:
                th.join()
            el2 = time.perf_counter() - t2
            sharded.shutdown()
            out = {"value": B * args.steps / el, "unit": "queries/s", "ms_per_step": el / args.steps * 1e3,
                   "p50_latency_ms": float(np.median(per) * 1e3), "p95_latency_ms": float(np.percentile(per, 95) * 1e3),
                   "two_batches_in_flight": {"value": B * 2 * per_thread / el2, "unit": "queries/s",
                                             "ms_per_step": el2 / (2 * per_thread) * 1e3, "threads_on_rank_0": 2},
                   "collective": ("own RCCL communicator inside the C ABI" if sharded.own_rccl else "torch.distributed")
                                 + (", all-gather + merge and requests on their own streams" if sharded._comm_stream is not None else ""),
                   "steps": args.steps, "_D": Dh, "_I": Ih,
                   "path": f"ShardedFlatIndex.leader_search on rank 0, {world - 1} follower rank(s) in follower_loop: one "
                           f"{sharded._msg_bytes} B message per batch (one H2D, one broadcast), local search, one "
                           f"all-gather of {world} x {12 * B * k + 8} B, device merge, one D2H of ids + scores + flag"}
        else:
            sharded.follower_loop()
        barrier()
        return out

    host_one_pass = host_leg()
    if host_one_pass is not None and "_I" in host_one_pass:
        Dh, Ih = host_one_pass.pop("_D"), host_one_pass.pop("_I")
        host_one_pass["identical_to_device_path"] = bool(np.array_equal(Ih, res_i) and
                                                         np.array_equal(Dh.view(np.uint32), res_s.view(np.uint32)))

    # Extra leg (reported beside the headline, never as `value`): the same step with the query encoder
    # in front — token ids resident in HBM -> bge-base-architecture encoder (fp32 MFMA, seeded random
    # weights: no checkpoint exists offline) -> CLS pooling + L2 norm -> scan + top-k.  Every rank
    # encodes the 32 queries itself (cheaper than a broadcast of the embeddings).
    legs_failed: dict[str, str] = {}
    enc_leg = None
    model = None
    enc_step = None
    try:
        if not args.no_encoder_leg and d == 768:
            from rag_inference_pipeline_amd import _native
            from rag_inference_pipeline_amd.bert import BertConfig, BertModel, pack_sequences, random_weights

            ecfg = BertConfig.bge_base()
            model = BertModel(ecfg, random_weights(ecfg, 0), device=dev)
            rng = np.random.default_rng(4321)
            lens = rng.integers(8, 21, size=B)
            seqs = [rng.integers(1000, 30000, size=int(n)).tolist() for n in lens]
            ids_np, _, cu_np = pack_sequences(seqs)
            ids_t, cu_t = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
            Qe2 = [torch.empty((B, d), dtype=torch.float32, device="cuda") for _ in range(3)]
            enc_n = [0]

            def enc_step() -> None:
                Qe = Qe2[enc_n[0] % 3]   # a batch's embeddings stay put until that batch has been collected
                enc_n[0] += 1
                model.forward_device(ids_t.data_ptr(), 0, cu_t.data_ptr(), B, int(cu_np[-1]), int(lens.max()),
                                     _native.BERT_OUT_CLS, True, Qe.data_ptr(), sptr)
                if sharded is None:
                    index.search_device(Qe.data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr(), sptr)
                else:
                    submit(Qe)

            # The same step PIPELINED: the encoder of batch i runs on a side stream, under the scan of batch i - 1 (the
            # product allows it: separate handles, the scheduler runs batches concurrently, batch_scheduler.py:286-288).
            # Three embedding buffers; a buffer is rewritten only after the search that read it has passed.
            side = torch.cuda.Stream()
            ev_enc = [torch.cuda.Event() for _ in range(3)]
            ev_srch = [torch.cuda.Event() for _ in range(3)]
            pipe_n = [0]

            def enc_pipe_step() -> None:
                n = pipe_n[0]
                pipe_n[0] += 1
                Qe = Qe2[n % 3]
                main = torch.cuda.current_stream()
                if n >= 3:
                    side.wait_event(ev_srch[n % 3])
                model.forward_device(ids_t.data_ptr(), 0, cu_t.data_ptr(), B, int(cu_np[-1]), int(lens.max()),
                                     _native.BERT_OUT_CLS, True, Qe.data_ptr(), side.cuda_stream)
                ev_enc[n % 3].record(side)
                main.wait_event(ev_enc[n % 3])
                if sharded is None:
                    index.search_device(Qe.data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr(), sptr)
                else:
                    submit(Qe)
                ev_srch[n % 3].record(main)

            def timed_loop(fn) -> float:
                for _ in range(max(2, args.warmup)):
                    fn()
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                barrier()
                el = time.perf_counter() - t0
                if dist is not None:
                    t = torch.tensor([el], dtype=torch.float64, device="cuda")
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    el = float(t.item())
                return el

            el = timed_loop(enc_step)
            # the encoder by itself (HIP events on its stream), for its own roofline object
            for _ in range(3):
                model.forward_device(ids_t.data_ptr(), 0, cu_t.data_ptr(), B, int(cu_np[-1]), int(lens.max()),
                                     _native.BERT_OUT_CLS, True, Qe2[0].data_ptr(), sptr)
            torch.cuda.synchronize()
            ee0, ee1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ee0.record()
            for _ in range(20):
                model.forward_device(ids_t.data_ptr(), 0, cu_t.data_ptr(), B, int(cu_np[-1]), int(lens.max()),
                                     _native.BERT_OUT_CLS, True, Qe2[0].data_ptr(), sptr)
            ee1.record()
            torch.cuda.synchronize()
            enc_only_ms = ee0.elapsed_time(ee1) / 20
            enc_roof = mfma_roofline(transformer_flops(ecfg.hidden, ecfg.intermediate, ecfg.n_layers, lens, head=False, first_only=True),
                                     enc_only_ms, F16_MFMA_PEAK_TF / 3.0,
                                     "fp16 MFMA, three products per fp32-accurate product: 2.5 PF/s / 3 (a 447-token batch is "
                                     "launch- and operand-latency-bound, not matrix-bound: DESIGN.md section 4)",
                                     "gemm_nt_ws_kernel<1|2,4> (64-row tiles, split-K)", "r04_encoder_mfma_util.json")
            enc_roof["encoder_alone_ms"] = enc_only_ms
            model.set_background(True)    # kernels that fit beside the scan's resident workgroups (include/rag_amd.h)
            el_p = timed_loop(enc_pipe_step)
            model.set_background(False)
            pipelined = {"value": B * args.steps / el_p, "unit": "queries/s", "ms_per_step": el_p / args.steps * 1e3,
                         "how": "encoder of batch i on a side stream under the scan of batch i - 1, its small-batch GEMMs in the "
                                "32-KiB-LDS form that fits beside the scan's resident workgroups (both on all CUs)"}
            # The same pipeline with the chip PARTITIONED: the encoder's stream owns 32 CUs (one per shader engine), the scan's
            # stream the other 224 — the scan is HBM-bound and loses 6 %, and the encoder's ~90 short dependent kernels no
            # longer queue behind the scan's resident workgroups (include/rag_amd.h rag_stream_create_masked).  Shares that
            # leave the shader engines unequal (16, 24, 40, 48 CUs) DOUBLE the scan's time: workgroups are dealt to engines by
            # count, an engine with fewer CUs than workgroups runs two persistent workgroups in turn.
            def run_partitioned(enc_cus: int, time_shared: dict) -> dict:
                from rag_inference_pipeline_amd.flat_index import create_masked_stream, destroy_stream

                total_cus = torch.cuda.get_device_properties(dev).multi_processor_count
                enc_raw = create_masked_stream(dev, 0, enc_cus)
                scan_raw = create_masked_stream(dev, enc_cus, total_cus - enc_cus)
                enc_st, scan_st = torch.cuda.ExternalStream(enc_raw), torch.cuda.ExternalStream(scan_raw)
                part_n = [0]

                def enc_part_step() -> None:
                    n = part_n[0]
                    part_n[0] += 1
                    Qe = Qe2[n % 3]
                    if n >= 3:
                        enc_st.wait_event(ev_srch[n % 3])
                    model.forward_device(ids_t.data_ptr(), 0, cu_t.data_ptr(), B, int(cu_np[-1]), int(lens.max()),
                                         _native.BERT_OUT_CLS, True, Qe.data_ptr(), enc_raw)
                    ev_enc[n % 3].record(enc_st)
                    scan_st.wait_event(ev_enc[n % 3])
                    index.search_device(Qe.data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr(), scan_raw)
                    ev_srch[n % 3].record(scan_st)

                model.set_cu_budget(enc_cus)
                index.set_cu_budget(total_cus - enc_cus)

                def alone(fn, st) -> float:   # ms per call of one side alone on its share of the chip
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for _ in range(10):
                        fn()
                    e1.record(st)
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / 10

                enc_alone = alone(lambda: model.forward_device(ids_t.data_ptr(), 0, cu_t.data_ptr(), B, int(cu_np[-1]), int(lens.max()),
                                                               _native.BERT_OUT_CLS, True, Qe2[0].data_ptr(), enc_raw), enc_st)
                scan_alone = alone(lambda: index.search_device(Qe2[0].data_ptr(), B, k, out_s.data_ptr(), out_i.data_ptr(), scan_raw), scan_st)
                el_q = timed_loop(enc_part_step)
                torch.cuda.synchronize()
                ids_part = out_i.cpu().numpy().copy()
                model.set_cu_budget(0)
                index.set_cu_budget(0)
                enc_step()
                torch.cuda.synchronize()
                same = bool(np.array_equal(ids_part, out_i.cpu().numpy()))
                del enc_st, scan_st
                destroy_stream(dev, enc_raw)
                destroy_stream(dev, scan_raw)
                return {"value": B * args.steps / el_q, "unit": "queries/s", "ms_per_step": el_q / args.steps * 1e3,
                             "how": f"encoder of batch i on a stream that owns {enc_cus} CUs ({enc_cus // 8} per XCD), scan of batch i - 1 on "
                                    f"a stream that owns the other {total_cus - enc_cus} (rag_stream_create_masked, rag_*_set_cu_budget)",
                             "encoder_cus": enc_cus, "encoder_alone_ms_on_its_cus": enc_alone, "scan_alone_ms_on_its_cus": scan_alone,
                             "identical_to_sequential": same, "time_shared_cus": time_shared}

            if sharded is None:   # (32: one CU of every shader engine of every XCD)
                pipelined = run_partitioned(int(os.environ.get("RAG_AMD_BENCH_ENCODER_CUS", "32")), pipelined)
            enc_leg = {"value": B * args.steps / el, "unit": "queries/s", "ms_per_step": el / args.steps * 1e3,
                       "tokens_per_batch": int(cu_np[-1]),
                       "roofline": enc_roof,
                       "pipelined": pipelined,
                       "encoder": "bge-base-en-v1.5 architecture (12x768), seeded random weights, CLS pooling + L2 norm, "
                                  "two-plane fp16 GEMMs (fp32 accuracy); token ids resident in HBM"}
    except Exception as exc:  # noqa: BLE001  (an extra leg must not cost the run its headline line)
        legs_failed["with_query_encoder"] = f"{type(exc).__name__}: {exc}"
        enc_leg = None

    # Extra leg (never `value`, one GPU only): the stage BASELINE configs[2] puts behind the scan — the cross-encoder
    # over 32 x 100 (query, document) pairs (reference reranker.py:237-272), ms-marco-MiniLM-L-6 architecture with
    # seeded random weights, token ids resident in HBM; in the default fp32-accurate mode and in the fp16 mode the
    # reference itself uses on a GPU (reranker.py:91-93).
    rerank_leg = None
    try:
        if not args.no_rerank_leg and world == 1:
            from rag_inference_pipeline_amd import _native
            from rag_inference_pipeline_amd.bert import BertConfig, BertModel, pack_sequences, random_weights

            rng = np.random.default_rng(99)
            plens = rng.integers(36, 76, size=B * 100)
            pseqs = [rng.integers(1000, 30000, size=int(n)).tolist() for n in plens]
            pids, ptypes, pcu = pack_sequences(pseqs, [[0] * 10 + [1] * (len(q) - 10) for q in pseqs])
            pids_t, ptypes_t, pcu_t = (torch.from_numpy(a).cuda() for a in (pids, ptypes, pcu))
            pout = torch.empty((len(pseqs), 1), dtype=torch.float32, device="cuda")
            rerank_leg = {"pairs": len(pseqs), "tokens": int(pcu[-1]),
                          "model": "cross-encoder/ms-marco-MiniLM-L-6-v2 architecture (6 x 384), seeded random weights; "
                                   "token ids resident in HBM, sigmoid scores left in HBM"}
            for mode, key in (("f32", "default_fp32_accurate"), ("f16", "fp16_mode")):
                rcfg = BertConfig.ms_marco_minilm_l6()
                rcfg.gemm_dtype = mode
                rmodel = BertModel(rcfg, random_weights(rcfg, 0), device=dev)
                st = torch.cuda.current_stream().cuda_stream

                def rpass():
                    rmodel.forward_device(pids_t.data_ptr(), ptypes_t.data_ptr(), pcu_t.data_ptr(), len(pseqs), int(pcu[-1]),
                                          int(plens.max()), _native.BERT_OUT_PROBS, False, pout.data_ptr(), st)
                for _ in range(3):
                    rpass()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    rpass()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                rflops = transformer_flops(rcfg.hidden, rcfg.intermediate, rcfg.n_layers, plens, head=True, first_only=True)
                if mode == "f32":
                    roof = mfma_roofline(rflops, ms, F16_MFMA_PEAK_TF / 3.0,
                                         "fp16 MFMA, three products per fp32-accurate product: 2.5 PF/s / 3",
                                         "gemm_nt_wl_kernel<2,1,3> (78 % of the pass)", "r04_rerank_gemm_mfma_util.json")
                else:
                    roof = mfma_roofline(rflops, ms, F16_MFMA_PEAK_TF, "fp16 MFMA dense, 2.5 PF/s",
                                         "gemm_nt_wt_kernel<1,8,4,...> (fp16 activations in fragment order)",
                                         "r04_rerank_f16_minilm_pass.json",
                                         busy_key=lambda pr: max((kk.get("mfma_busy_frac") or 0.0) for kk in pr["kernels"]
                                                                 if "gemm_nt_wt" in kk["kernel"]),
                                         floor_file="r04_rerank_f16_minilm_pass.json")
                rerank_leg[key] = {"ms_per_batch": ms, "pairs_per_s": len(pseqs) / ms * 1e3,
                                   "score_checksum": float(pout.sum().item()), "roofline": roof}
                rmodel.close()
            rerank_leg["note"] = ("default: GEMMs on the fp16 matrix cores with every fp32 operand as two fp16 planes (fp32 accuracy); "
                                  "fp16_mode (RAG_AMD_RERANKER_DTYPE=f16): fp16 activations stored in MFMA-fragment order, GEMMs and "
                                  "attention on the fp16 matrix cores, fp32 accumulation and statistics")
    except Exception as exc:  # noqa: BLE001  (an extra leg must not cost the run its headline line)
        legs_failed["rerank_stage"] = f"{type(exc).__name__}: {exc}"
        rerank_leg = None

    # Extra leg (never `value`): the IVFFlat `nprobe` mode at the shape the reference's generator writes
    # (scripts/create_test_docs.py:12,83-104: 4.5M x 768, L2, nlist 4096, nprobe 64) — opt-in RAG_AMD_IVF_MODE=nprobe.
    # Bound: HBM, against the bytes of the UNION of a batch's probed lists (each list is read once per pass of 32 queries).
    ivf_leg = None
    try:
        if world == 1 and not args.no_ivf_leg and d == 768:
            import scripts.bench_ivf as bivf

            base = dict(n=4_500_000, d=768, nlist=4096, nprobe=64, k=k, steps=20, warmup=3, exhaustive=False, unit=False)
            kept: dict = {}
            clustered = bivf.run(argparse.Namespace(**base, batches="1,32", clustered=True), keep=kept)
            ivf_cpu = None
            if not args.no_cpu_baseline:
                # the CPU path beside it: the oracle's restatement of the nprobe search (flat oracle over the centroids, then
                # over each query's gathered lists; C + OpenMP) on ONE batch of 32 of the same queries
                from oracle import flat as _oracle

                lst, q32 = kept["lists"], kept["queries"]
                avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                cores = max(1, min(16, avail, _oracle.num_threads()))   # the reference's faiss_threads policy, as the headline's baseline
                t0 = time.perf_counter()
                _oracle.ivf_search(lst.centroids, lst.quantizer_metric, lst.rows, lst.ids, lst.offsets, q32, k, 64, 1, nthreads=cores)
                dt = time.perf_counter() - t0
                ivf_cpu = {"value": len(q32) / dt, "unit": "queries/s", "cores": cores, "kind": "port",
                           "sample": f"oracle/flat.py:ivf_search, one batch of {len(q32)} queries on the clustered 4.5M x 768 lists "
                                     f"({dt * 1e3:.0f} ms)"}
            kept.clear()
            generator = bivf.run(argparse.Namespace(**base, batches="32", clustered=False))
            b32 = clustered["batches"]["32"]
            ach = b32["scan_bytes"] / b32["ms_per_batch"] / 1e6
            ivf_leg = {"clustered_balanced_lists": clustered, "generator_gaussian_rows": generator, "cpu_baseline": ivf_cpu,
                       "roofline": {"bound": "hbm", "kernel": "scan_topk_kernel<P=1, MAP> fp16 screening pass over the probed lists (whole search timed: coarse "
                                    "quantizer, plan, screening pass, resolve, self-disabling exact fallback)", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes_per_batch": b32["scan_bytes"],
                                    "bytes_basis": "rows of the union of the batch's probed lists, each once, in the fp16 copy the "
                                                   "two-stage search screens (2 B x d64 per row); the same rows in fp32: union_bytes",
                                    "traffic": (_profile_json("r04_ivf_hbm_traffic.json") or {}).get("two_stage_screening_pass", {}).get("traffic_bytes_per_launch"),
                                    "traffic_source": "profiles/r04_ivf_hbm_traffic.json: FETCH_SIZE x2 (gfx950 correction) of the screening "
                                                      "pass from a separate rocprofv3 --pmc run of this workload; not collected inside this run",
                                    "workload": "clustered, batch 32, two-stage search (the default for k <= 100)",
                                    "kernel_only_source": "profiles/r04_ivf_nprobe.json"},
                       "note": "generator_gaussian_rows: iid Gaussian rows have no cluster structure — k-means on 10 000 of "
                               "them leaves most lists empty and a few huge, every query's 64 lists hold nearly the whole corpus "
                               "and the mode costs what the exhaustive scan costs; clustered: rows from a mixture of 4096 "
                               "centres, the regime an embedding corpus is in (a query's lists are 1.6 % of the rows)"}
    except Exception as exc:  # noqa: BLE001
        legs_failed["ivf_nprobe"] = f"{type(exc).__name__}: {exc}"
        ivf_leg = None

    # Extra leg (never `value`): the same step through the two-stage exact search — fp16 screening scan
    # of a scaled copy of the corpus, canonical fp32 re-scoring of the band, per-query certificate,
    # device-side fp32 fallback (include/rag_amd.h rag_index_set_screening).  Same ids, same score bits.
    two_leg = None
    try:
        if not args.no_two_stage_leg and d <= 1024 and k <= 100:
            from rag_inference_pipeline_amd.flat_index import SCREEN_FP16

            index.set_screening(SCREEN_FP16)
            if index.screening == SCREEN_FP16:
                for _ in range(max(2, args.warmup)):
                    step()
                barrier()
                index.screen_stats(reset=True)
                index.profile_enable(True)
                index.profile(reset=True)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                barrier()
                el = time.perf_counter() - t0
                s1_ms_total, s1_launches = index.profile(reset=True)
                index.profile_enable(False)
                if dist is not None:
                    t = torch.tensor([el], dtype=torch.float64, device="cuda")
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    el = float(t.item())
                lat2 = []
                for _ in range(args.latency_steps):
                    barrier()
                    t1 = time.perf_counter()
                    step()
                    torch.cuda.synchronize()
                    lat2.append(time.perf_counter() - t1)
                st = index.screen_stats()
                step(); barrier()
                r2_s0 = (fin["s"] if world > 1 else out_s).cpu().numpy()
                r2_i0 = (fin["i"] if world > 1 else out_i).cpu().numpy()
                host_two_stage = host_leg()
                if host_two_stage is not None and "_I" in host_two_stage:
                    Dh, Ih = host_two_stage.pop("_D"), host_two_stage.pop("_I")
                    host_two_stage["identical_to_device_path"] = bool(np.array_equal(Ih, r2_i0) and
                                                                      np.array_equal(Dh.view(np.uint32), r2_s0.view(np.uint32)))
                enc2 = None
                if enc_step is not None:  # text ids -> encoder -> two-stage search
                    e2 = timed_loop(enc_step)
                    model.set_background(True)
                    e2p = timed_loop(enc_pipe_step)
                    model.set_background(False)
                    pipe2 = {"value": B * args.steps / e2p, "unit": "queries/s", "ms_per_step": e2p / args.steps * 1e3}
                    if sharded is None:   # the two-stage scan is half as long: the encoder gets 64 CUs to stay the shorter side
                        pipe2 = run_partitioned(int(os.environ.get("RAG_AMD_BENCH_ENCODER_CUS_TWO_STAGE", "64")), pipe2)
                    enc2 = {"value": B * args.steps / e2, "unit": "queries/s", "ms_per_step": e2 / args.steps * 1e3, "pipelined": pipe2}
                    step()  # leave the precomputed-embedding results in the output buffers for the comparison below
                    barrier()
                r2_s = (fin["s"] if world > 1 else out_s).cpu().numpy()
                r2_i = (fin["i"] if world > 1 else out_i).cpu().numpy()
                d64 = (d + 63) // 64 * 64
                s1_ms = s1_ms_total / max(s1_launches, 1)
                s1_bytes = 2.0 * n_local * d64
                two_leg = {
                    "value": B * args.steps / el, "unit": "queries/s", "ms_per_step": el / args.steps * 1e3,
                    "p50_latency_ms": float(np.median(lat2) * 1e3) if lat2 else None,
                    "identical_to_one_pass": bool(np.array_equal(r2_i, res_i) and
                                                  np.array_equal(r2_s.view(np.uint32), res_s.view(np.uint32))),
                    "certificate_fallbacks_rank0": st["fallbacks"], "queries_rank0": st["queries"],
                    "max_observed_error_over_bound": st["max_err_ratio"],
                    "roofline": {"bound": "hbm", "kernel": "scan_topk_kernel<P=1> (fp16 screening pass)",
                                 "achieved": s1_bytes / (s1_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": s1_bytes / (s1_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "avg_kernel_ms": s1_ms,
                                 "algorithmic_bytes_per_launch": s1_bytes},
                    "note": "fp16 copy of the corpus read once per batch (2*N*d bytes), exact fp32 second stage; "
                            "+50% index memory",
                }
                if enc2 is not None:
                    two_leg["with_query_encoder"] = enc2
                if host_two_stage is not None:
                    two_leg["host_submit_to_host_results"] = host_two_stage
    except Exception as exc:  # noqa: BLE001  (an extra leg must not cost the run its headline line)
        legs_failed["two_stage_exact"] = f"{type(exc).__name__}: {exc}"
        two_leg = None

    if model is not None:
        model.close()

    if rank == 0:
        scan_ms = scan_ms_total / max(scan_launches, 1)
        # HBM bytes per launch from the committed PMC pass (FETCH_SIZE, corrected as the microarch
        # guide prescribes); only quoted when it was collected on this exact workload
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as fh:
                tr = json.load(fh)
            w = tr["workload"]
            if (w["rows"], w["dim"], w["batch"], w["k"], w["n_gpus"]) == (N, d, B, k, world):
                traffic = tr["traffic_bytes_per_launch"]
                if two_leg is not None and "two_stage" in tr:
                    two_leg["roofline"]["traffic"] = tr["two_stage"]["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
        alg_bytes = 4.0 * n_local * d  # corpus read once per batch, per GPU
        achieved = alg_bytes / (scan_ms * 1e-3) / 1e9
        out = {
            # BASELINE.json's metric string; `value` is the queries/sec half, p50 is in p50_latency_ms
            "metric": "queries/sec + p50 retrieval latency, 10M\u00d7768 corpus, batch=32, k=10",
            "value": B * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"flat inner-product scan + exact top-{k}: {N} x {d} fp32 rows over {world} GPU(s) "
                            f"({n_local} rows on rank 0), batch {B} precomputed unit-norm query embeddings "
                            "resident in HBM (configs/retrieval_faiss_only.yaml path)"
                            + ("; per-shard top-k merged by one RCCL all-gather + device merge, two batches in flight"
                               if world > 1 else ""),
                "note": "value = the one-pass fp32 scan with queries resident in HBM (the path SURVEY 8(d) defines the "
                        "roofline on); host_submit_to_host_results = the same batches through rag_index_search "
                        "(host queries in, results on host, PCIe inclusive) with its own q/s and p50; "
                        "two_stage_exact = the same step through the exact two-stage search FAISSStore uses by "
                        "default (same ids and score bits), reported beside it",
                "rows": N, "dim": d, "batch": B, "k": k, "rows_per_gpu": n_local,
                "parallelism": f"corpus-shard x{world}",
            },
            "p50_latency_ms": p50_ms,
            "roofline": {
                "bound": "hbm",
                "kernel": "scan_topk_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": (f"profiles/{TRAFFIC_FILE}: FETCH_SIZE x2 (gfx950 correction) from a separate rocprofv3 --pmc "
                                   "pass of this workload; not collected inside this run") if traffic is not None else None,
                "avg_kernel_ms": scan_ms,
                "launches": scan_launches,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
            "result_checksum": int(np.bitwise_xor.reduce(res_i.ravel())) if res_i.size else 0,
            "result_checksum_expected": EXPECTED_ID_CHECKSUM.get((N, d, B, k, args.seed)),
            "top1_score_mean": float(res_s[:, 0].mean()),
        }
        if host_one_pass is not None:
            out["host_submit_to_host_results"] = host_one_pass
        # the step a deployment serves through FAISSStore.search (two-stage by default), next to `value`
        serving = (two_leg or {}).get("host_submit_to_host_results") or host_one_pass
        if serving is not None:
            out["serving_mode"] = dict(serving, search="two-stage exact (FAISSStore default)" if two_leg and
                                       "host_submit_to_host_results" in two_leg else "one-pass fp32 scan")
        if enc_leg is not None:
            out["with_query_encoder"] = enc_leg
        if two_leg is not None:
            out["two_stage_exact"] = two_leg
        if rerank_leg is not None:
            out["rerank_stage"] = rerank_leg
        if ivf_leg is not None:
            out["ivf_nprobe"] = ivf_leg
        if legs_failed:
            out["legs_failed"] = legs_failed
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if sharded is not None:
        sharded.close()
    index.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
