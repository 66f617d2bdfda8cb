"""IVFFlat `nprobe` mode at the shape the reference's own generator writes (scripts/create_test_docs.py:12,83-104:
4.5M x 768 fp32, L2, nlist 4096, nprobe 64; Gaussian rows, centroids taken from 10 000 training rows).

    python scripts/bench_ivf.py [--n 4500000] [--d 768] [--nlist 4096] [--nprobe 64] [--batches 1,8,32] [--steps 30]
                                [--exhaustive] [--unit | --clustered]

The corpus is built on the GPU with torch (setup only: rows drawn chunk by chunk, assigned to their nearest centroid
with one matmul per chunk, grouped by list) and handed to rag_ivf_set_lists through the host, as a file load would.
Timed: rag_ivf_search_device on one stream, queries resident in HBM, HIP events around `steps` batches of DIFFERENT
queries (eight query sets in rotation).  Prints one JSON object:
  per batch size  ms per batch, queries/s, algorithmic bytes (rows of the UNION of the batch's probed lists x d x 4:
                  what a search that reads every probed list once must move; `scan_bytes`: the same rows in the fp16
                  copy the two-stage search screens, d rounded up to 64), GB/s of the whole search against them,
                  and the bytes a per-(query, list) scan would read (rows counted once per probing query);
  with --exhaustive the flat search of the same rows beside it (what RAG_AMD_IVF_MODE=exhaustive costs).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


_CENTERS = None   # --clustered: the mixture's unit-norm centres


def draw(shape, g, unit: bool):
    x = torch.randn(shape, generator=g, device="cuda")
    if _CENTERS is not None:   # a row = unit(centre + noise of norm ~0.7): cos(row, own centre) ~ 0.82, other centres ~ 0
        j = torch.randint(0, _CENTERS.shape[0], (shape[0],), generator=g, device="cuda")
        x = _CENTERS[j] + x * (0.7 / shape[1] ** 0.5)
        return x / x.norm(dim=1, keepdim=True)
    return x / x.norm(dim=1, keepdim=True) if unit else x


def build_lists(n: int, d: int, nlist: int, nprobe: int, unit: bool, clustered: bool = False, seed: int = 1234):
    from rag_inference_pipeline_amd.index_io import IVFFlatLists
    g = torch.Generator(device="cuda").manual_seed(seed)
    if clustered:
        global _CENTERS
        c = torch.randn((nlist, d), generator=g, device="cuda")
        _CENTERS = c / c.norm(dim=1, keepdim=True)
    train = draw((max(10000, nlist), d), g, unit)          # the generator trains on 10 000 rows (create_test_docs.py:88-90)
    cent = train[torch.randperm(train.shape[0], generator=g, device="cuda")[:nlist]].contiguous()
    for _ in range(10):                                     # Lloyd iterations (faiss runs 25; empty lists keep their centroid)
        a = ((cent * cent).sum(1)[None, :] - 2.0 * (train @ cent.T)).argmin(1)
        sums = torch.zeros_like(cent).index_add_(0, a, train)
        cnt = torch.bincount(a, minlength=nlist).to(torch.float32)
        cent = torch.where(cnt[:, None] > 0, sums / cnt.clamp(min=1)[:, None], cent)
    if clustered:   # a quantizer trained on enough rows finds the mixture's centres (10 000 rows for 4096 lists do not)
        cent = _CENTERS.clone()
    cn = (cent * cent).sum(1)
    rows = torch.empty((n, d), dtype=torch.float32, device="cuda")
    assign = torch.empty((n,), dtype=torch.int32, device="cuda")
    chunk = 1 << 18
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        x = draw((r1 - r0, d), g, unit)
        rows[r0:r1] = x
        assign[r0:r1] = (cn[None, :] - 2.0 * (x @ cent.T)).argmin(1).to(torch.int32)
    order = torch.argsort(assign.to(torch.int64), stable=True)
    counts = torch.bincount(assign.to(torch.int64), minlength=nlist)
    offsets = torch.zeros(nlist + 1, dtype=torch.int64, device="cuda")
    offsets[1:] = torch.cumsum(counts, 0)
    rows_sorted = rows[order]
    del rows
    lists = IVFFlatLists(cent.cpu().numpy(), 1, rows_sorted.cpu().numpy(), order.cpu().numpy().astype(np.int64),
                         offsets.cpu().numpy(), 1, nprobe)
    del rows_sorted, order, assign
    torch.cuda.empty_cache()
    return lists, cent


def run(a, keep: dict | None = None) -> dict:
    """`a`: a namespace with n, d, nlist, nprobe, k, batches (comma list), steps, warmup, exhaustive, unit, clustered.
    `keep` (a dict): receives the lists and the first query set of the last batch size (bench.py times its CPU baseline on
    them)."""
    global _CENTERS
    _CENTERS = None
    from rag_inference_pipeline_amd.ivf_index import IVFFlatIndex

    t0 = time.time()
    lists, cent = build_lists(a.n, a.d, a.nlist, a.nprobe, a.unit, a.clustered)
    sizes = np.diff(lists.offsets)
    t1 = time.time()
    idx = IVFFlatIndex(lists, nprobe=a.nprobe)
    t2 = time.time()
    kind = ("clustered unit-norm rows (mixture of nlist centres = the quantizer)" if a.clustered else
            ("unit-norm" if a.unit else "Gaussian") + " rows, k-means on 10 000 training rows")
    out = {"workload": f"IVFFlat L2 {a.n} x {a.d} fp32, nlist {a.nlist}, nprobe {a.nprobe}, k {a.k}; {kind}, nearest-centroid "
                       "lists (reference scripts/create_test_docs.py:83-104)",
           "list_rows": {"mean": float(sizes.mean()), "min": int(sizes.min()), "max": int(sizes.max()),
                         "empty_lists": int((sizes == 0).sum())},
           "build_s": round(t1 - t0, 1), "set_lists_s": round(t2 - t1, 1), "batches": {}}
    stream = torch.cuda.Stream()
    g = torch.Generator(device="cuda").manual_seed(99)
    cn = (cent * cent).sum(1)
    for nq in [int(b) for b in str(a.batches).split(",")]:
        sets = [draw((nq, a.d), g, a.unit) for _ in range(8)]
        s = torch.empty((nq, a.k), dtype=torch.float32, device="cuda")
        i = torch.empty((nq, a.k), dtype=torch.int64, device="cuda")
        union_rows, pair_rows = [], []
        for q in sets:   # byte counts from the same nearest-list rule (ties aside)
            probe = (cn[None, :] - 2.0 * (q @ cent.T)).topk(a.nprobe, dim=1, largest=False).indices.cpu().numpy()
            union_rows.append(sum(int(sizes[np.unique(probe[b:b + 32])].sum()) for b in range(0, nq, 32)))   # per pass of <= 32 queries
            pair_rows.append(int(sizes[probe].sum()))
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for w in range(a.warmup):
                idx.search_device(sets[w % 8].data_ptr(), nq, a.k, s.data_ptr(), i.data_ptr(), stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for st in range(a.steps):
                idx.search_device(sets[st % 8].data_ptr(), nq, a.k, s.data_ptr(), i.data_ptr(), stream.cuda_stream)
            e1.record(stream)
        stream.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        ub = float(np.mean([union_rows[st % 8] for st in range(a.steps)])) * a.d * 4
        pb = float(np.mean([pair_rows[st % 8] for st in range(a.steps)])) * a.d * 4
        # what the list scan reads: the fp32 rows, or — two-stage search, k <= 100 — their fp16 copy (d rounded up to 64)
        two = bool(idx.two_stage) and a.k <= 100
        rb = ub / (a.d * 4) * ((a.d + 63) // 64 * 64) * 2 if two else ub
        out["batches"][str(nq)] = {"ms_per_batch": round(ms, 4), "queries_per_s": round(nq / ms * 1e3, 1),
                                   "union_bytes": int(ub), "pair_bytes": int(pb), "two_stage": two, "scan_bytes": int(rb),
                                   "search_gbps_vs_scan_bytes": round(rb / ms / 1e6, 1),
                                   "search_gbps_vs_union": round(ub / ms / 1e6, 1),
                                   "corpus_fraction_read": round(ub / (a.n * a.d * 4.0), 4),
                                   "id_checksum": int(i.sum().item())}
        print(f"nq={nq}: {ms:.4f} ms/batch, union {ub / 1e9:.3f} GB, scan reads {rb / 1e9:.3f} GB -> {rb / ms / 1e6:.0f} GB/s "
              f"(whole search{', two-stage' if two else ''})", file=sys.stderr)
    if a.exhaustive:
        from rag_inference_pipeline_amd.flat_index import FlatIndex
        flat = FlatIndex(a.d, 1)
        for r0 in range(0, a.n, 1 << 20):
            flat.add(lists.rows[r0:r0 + (1 << 20)])
        ex = {}
        for nq in [int(b) for b in str(a.batches).split(",")]:
            q = draw((nq, a.d), g, a.unit)
            s = torch.empty((nq, a.k), dtype=torch.float32, device="cuda")
            i = torch.empty((nq, a.k), dtype=torch.int64, device="cuda")
            with torch.cuda.stream(stream):
                for _ in range(3):
                    flat.search_device(q.data_ptr(), nq, a.k, s.data_ptr(), i.data_ptr(), stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(10):
                    flat.search_device(q.data_ptr(), nq, a.k, s.data_ptr(), i.data_ptr(), stream.cuda_stream)
                e1.record(stream)
            stream.synchronize()
            ex[str(nq)] = round(e0.elapsed_time(e1) / 10, 4)
        out["exhaustive_flat_ms_per_batch"] = ex
        # ... and through the two-stage exact search FAISSStore enables by default for a flat index
        flat.set_screening(True)
        ex2 = {}
        for nq in [int(b) for b in str(a.batches).split(",")]:
            q = draw((nq, a.d), g, a.unit)
            s = torch.empty((nq, a.k), dtype=torch.float32, device="cuda")
            i = torch.empty((nq, a.k), dtype=torch.int64, device="cuda")
            with torch.cuda.stream(stream):
                for _ in range(3):
                    flat.search_device(q.data_ptr(), nq, a.k, s.data_ptr(), i.data_ptr(), stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(10):
                    flat.search_device(q.data_ptr(), nq, a.k, s.data_ptr(), i.data_ptr(), stream.cuda_stream)
                e1.record(stream)
            stream.synchronize()
            ex2[str(nq)] = round(e0.elapsed_time(e1) / 10, 4)
        out["exhaustive_flat_two_stage_ms_per_batch"] = ex2
        flat.close()
    idx.close()
    _CENTERS = None
    if keep is not None:
        keep["lists"] = lists
        keep["queries"] = sets[0].cpu().numpy()
    del lists, cent
    torch.cuda.empty_cache()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4_500_000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=64)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batches", default="1,8,32")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--exhaustive", action="store_true")
    ap.add_argument("--unit", action="store_true", help="unit-norm rows and queries (what an embedding model emits)")
    ap.add_argument("--clustered", action="store_true", help="rows and queries from a mixture of nlist unit-norm centres, "
                    "the centres as the quantizer: balanced lists, a query's nprobe lists are 1.6 %% of the corpus")
    print(json.dumps(run(ap.parse_args())))


if __name__ == "__main__":
    main()
