"""Turn a flat index file into the IVFFlat (`IwFl`) file the reference's generator would write with faiss.

The reference builds its index with faiss (scripts/create_test_docs.py:83-104: a flat L2 quantizer, `IndexIVFFlat` with
nlist 4096, trained on 10 000 vectors, nprobe 64 saved in the file) and FAISSStore.load sets `index.nprobe`
(faiss_store.py:84-92).  A deployment that embeds its documents with this build's encoder (tools/build_index.py) gets a flat
file; this tool groups its rows into inverted lists so that `RAG_AMD_IVF_MODE=nprobe` can search it as the reference searches
its own file — without faiss:

    python -m rag_inference_pipeline_amd.tools.build_ivf --index faiss_index.f32 --out faiss_index.ivf \
        [--nlist 4096] [--nprobe 64] [--train-rows 10000] [--iters 25] [--seed 1234] [--device 0]

* training: Lloyd's k-means on a random sample of the rows (host, numpy — 10 000 x nlist x d per iteration), L2, empty
  clusters keep their centroid; like faiss, the quantizer is a flat L2 index over the centroids;
* assignment: every row goes to its nearest centroid — on the GPU, through this build's own flat search (k = 1 over the
  centroids: the coarse quantizer the nprobe mode itself uses, so a row's list is the first list its own vector would
  probe);
* the file: faiss's `IwFl` layout (index_io.write_ivfflat_file), stored id = row number of the flat file, lists in
  ascending id order — what index_io.read_ivfflat_lists / read_index_file load.
"""

from __future__ import annotations

import argparse
import time
from typing import Callable

import numpy as np

from .. import index_io


def train_centroids(sample: np.ndarray, nlist: int, iters: int = 25, seed: int = 1234) -> np.ndarray:
    """Lloyd's k-means (L2) on `sample` (n x d fp32): nlist centroids, float64 accumulation, empty clusters keep theirs."""
    sample = np.ascontiguousarray(sample, dtype=np.float32)
    n, d = sample.shape
    rng = np.random.default_rng(seed)
    if n == 0:
        return np.zeros((nlist, d), dtype=np.float32)
    pick = rng.choice(n, size=nlist, replace=n < nlist)
    cent = sample[pick].astype(np.float32).copy()
    for _ in range(max(0, iters)):
        cn = (cent.astype(np.float64) ** 2).sum(1)
        assign = np.empty(n, dtype=np.int64)
        for lo in range(0, n, 16384):
            blk = sample[lo:lo + 16384]
            assign[lo:lo + 16384] = np.argmin(cn[None, :] - 2.0 * (blk @ cent.T).astype(np.float64), axis=1)
        sums = np.zeros((nlist, d), dtype=np.float64)
        np.add.at(sums, assign, sample.astype(np.float64))
        counts = np.bincount(assign, minlength=nlist)
        moved = counts > 0
        new = cent.copy()
        new[moved] = (sums[moved] / counts[moved, None]).astype(np.float32)
        if np.array_equal(new, cent):
            break
        cent = new
    return np.ascontiguousarray(cent, dtype=np.float32)


def assign_on_gpu(rows: np.ndarray, centroids: np.ndarray, device: int = 0, chunk: int = 32768) -> np.ndarray:
    """Nearest centroid (L2, ties to the lower list number) of every row: this build's flat search, k = 1."""
    from ..flat_index import METRIC_L2, FlatIndex   # raises without librag_amd.so: no CPU fallback

    quantizer = FlatIndex(int(centroids.shape[1]), METRIC_L2, device=device)
    try:
        quantizer.add(np.ascontiguousarray(centroids, dtype=np.float32))
        out = np.empty(rows.shape[0], dtype=np.int64)
        for lo in range(0, rows.shape[0], chunk):
            _, ids = quantizer.search(np.ascontiguousarray(rows[lo:lo + chunk], dtype=np.float32), 1)
            out[lo:lo + chunk] = ids[:, 0]
        return out
    finally:
        quantizer.close()


def build_ivf(rows: np.ndarray, nlist: int, metric: int, out: str, nprobe: int = 64, train_rows: int = 10000,
              iters: int = 25, seed: int = 1234, device: int = 0,
              assign: Callable[[np.ndarray, np.ndarray], np.ndarray] | None = None) -> dict:
    """Write `out` (IwFl).  `assign(rows, centroids) -> list number per row` defaults to the GPU flat search."""
    rows = np.asarray(rows, dtype=np.float32)
    n, d = rows.shape
    if nlist <= 0:
        raise ValueError("nlist must be positive")
    t0 = time.time()
    rng = np.random.default_rng(seed)
    sample = rows[np.sort(rng.choice(n, size=min(n, max(train_rows, nlist)), replace=False))] if n else rows
    cent = train_centroids(np.ascontiguousarray(sample), nlist, iters, seed)
    t1 = time.time()
    lists = (assign or (lambda r, c: assign_on_gpu(r, c, device)))(rows, cent) if n else np.zeros(0, dtype=np.int64)
    if lists.shape != (n,) or (n and (lists.min() < 0 or lists.max() >= nlist)):
        raise RuntimeError("assignment must give one list number in [0, nlist) per row")
    t2 = time.time()
    index_io.write_ivfflat_file(out, rows, cent, lists, metric, nprobe=nprobe)
    sizes = np.bincount(lists, minlength=nlist) if n else np.zeros(nlist, dtype=np.int64)
    return {"rows": int(n), "dim": int(d), "nlist": int(nlist), "nprobe": int(nprobe), "empty_lists": int((sizes == 0).sum()),
            "largest_list": int(sizes.max()) if nlist else 0, "train_s": round(t1 - t0, 2), "assign_s": round(t2 - t1, 2),
            "write_s": round(time.time() - t2, 2)}


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--index", required=True, help="a flat index file index_io.read_index_file loads (.f32 + sidecar, .npy, FAISS flat)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=64, help="the value saved in the file (FAISS_NPROBE overrides it at load)")
    ap.add_argument("--metric", default=None, choices=["ip", "l2"], help="default: the flat file's metric")
    ap.add_argument("--train-rows", type=int, default=10000)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    rows, metric = index_io.read_index_file(a.index, index_io.metric_from_name(a.metric or "ip"), mmap=True)
    if a.metric is not None:
        metric = index_io.metric_from_name(a.metric)
    info = build_ivf(rows, a.nlist, metric, a.out, a.nprobe, a.train_rows, a.iters, a.seed, a.device)
    print(f"wrote {a.out}: {info}")


if __name__ == "__main__":
    main()
