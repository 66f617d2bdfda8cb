"""On-disk corpus formats the flat index can be loaded from.

`FAISS_INDEX_PATH` in the reference names a file for `faiss.read_index`
(reference src/pipeline/components/faiss_store.py:54-69).  A drop-in has to accept that path, so
three formats are read here, all yielding (rows float32 [n, d], metric):

  * FAISS flat index files — fourcc `IxFI` (inner product) / `IxF2`, `IxFl` (L2): header
    {d:i32, ntotal:i64, 2 x i64 unused, is_trained:u8, metric_type:i32} then a u64 element count
    and the raw fp32 payload.  This is the *published* upstream layout (faiss/impl/index_write.cpp),
    restated from its documentation: faiss is not installed here, so it is validated only through
    this module's own writer and structural checks (payload size must equal ntotal * d).
  * `.npy` — a 2-D float32 array; metric from the optional `<path>.json` sidecar or the caller.
  * raw fp32 (`.f32` / `.fvecs`-less flat binary) with a mandatory sidecar `<path>.json`
    {"d": .., "ntotal": .., "metric": "ip" | "l2"} — the format bench/corpus tools write, suited
    to memory-mapping 30 GB shards.

  * FAISS `IndexIVFFlat` files (fourcc `IwFl`, what the reference's generator writes,
    scripts/create_test_docs.py:83-104): the stored vectors are raw fp32 (IVFFlat keeps no residuals),
    so the inverted lists are unpacked back into row order (row = stored id, which must be a
    permutation of 0..ntotal-1) and searched exhaustively — the nprobe = nlist limit of that index.
    The coarse quantizer is skipped.  Same caveat as above: layout restated from upstream, validated
    structurally (list sizes must sum to ntotal, ids must be a permutation, file length must match)
    and through this module's own writer.

Other IVF/PQ variants (`IwPQ`, `IwQ4`, ...) hold compressed codes and are rejected with a message.
"""

from __future__ import annotations

import json
import os
import struct
from pathlib import Path

import numpy as np

from ._native import METRIC_INNER_PRODUCT, METRIC_L2

_FLAT_FOURCC = {b"IxFI": METRIC_INNER_PRODUCT, b"IxF2": METRIC_L2, b"IxFl": None}
_HEADER = struct.Struct("<iqqqBi")  # d, ntotal, dummy, dummy, is_trained, metric_type


def metric_from_name(name: str) -> int:
    key = str(name).strip().lower()
    if key in ("ip", "inner_product", "dot", "cosine", "0"):
        return METRIC_INNER_PRODUCT
    if key in ("l2", "euclidean", "1"):
        return METRIC_L2
    raise ValueError(f"unknown metric name: {name!r}")


def write_flat_index(path: str | os.PathLike, rows: np.ndarray, metric: int = METRIC_INNER_PRODUCT) -> None:
    """Write rows as a FAISS-layout flat index file (IxFI / IxF2)."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    if rows.ndim != 2:
        raise ValueError("rows must be 2-D")
    n, d = rows.shape
    fourcc = b"IxFI" if metric == METRIC_INNER_PRODUCT else b"IxF2"
    metric_type = 0 if metric == METRIC_INNER_PRODUCT else 1  # faiss::METRIC_INNER_PRODUCT / METRIC_L2
    with open(path, "wb") as fh:
        fh.write(fourcc)
        fh.write(_HEADER.pack(d, n, 1 << 20, 1 << 20, 1, metric_type))
        fh.write(struct.pack("<Q", n * d))
        fh.write(rows.tobytes())


class _Reader:
    def __init__(self, path: Path) -> None:
        self.buf = np.memmap(path, dtype=np.uint8, mode="r")
        self.pos = 0
        self.path = path

    def take(self, fmt: str):
        st = struct.Struct("<" + fmt)
        if self.pos + st.size > self.buf.size:
            raise ValueError(f"{self.path}: truncated file")
        vals = st.unpack(self.buf[self.pos:self.pos + st.size].tobytes())
        self.pos += st.size
        return vals if len(vals) > 1 else vals[0]

    def array(self, dtype, count: int) -> np.ndarray:
        nbytes = int(count) * np.dtype(dtype).itemsize
        if count < 0 or self.pos + nbytes > self.buf.size:
            raise ValueError(f"{self.path}: truncated file")
        out = np.frombuffer(self.buf, dtype=dtype, count=int(count), offset=self.pos)
        self.pos += nbytes
        return out

    def index_header(self) -> tuple[int, int, int]:
        d, ntotal, _, _, _, metric_type = self.take("iqqqBi")
        if metric_type > 1:
            self.take("f")  # metric_arg
        return d, ntotal, metric_type


class IVFFlatLists:
    """What an `IwFl` file holds, as the nprobe mode needs it (rag_ivf_set_lists): the coarse quantizer's centroids
    and metric, the rows in LIST order with their stored ids, the list boundaries, the metric and the nprobe value
    saved in the file."""

    def __init__(self, centroids: np.ndarray, quantizer_metric: int, rows: np.ndarray, ids: np.ndarray,
                 offsets: np.ndarray, metric: int, nprobe: int) -> None:
        self.centroids, self.quantizer_metric = centroids, quantizer_metric
        self.rows, self.ids, self.offsets = rows, ids, offsets
        self.metric, self.nprobe = metric, nprobe

    @property
    def nlist(self) -> int:
        return int(self.centroids.shape[0])

    @property
    def ntotal(self) -> int:
        return int(self.rows.shape[0])

    def shard(self, rank: int, world: int) -> "IVFFlatLists":
        """Rank `rank`'s share of a corpus-sharded deployment: the same centroids and, of EVERY list, the contiguous
        rows [len * rank / world, len * (rank + 1) / world) with their stored ids — the union of the ranks' candidates
        for a probe set is the unsharded candidate set, so per-rank top-k lists merge to the unsharded result."""
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world {world}")
        off = np.asarray(self.offsets, dtype=np.int64)
        lens = np.diff(off)
        lo = off[:-1] + lens * rank // world
        hi = off[:-1] + lens * (rank + 1) // world
        take = np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)] or [np.zeros(0, np.int64)]).astype(np.int64)
        new_off = np.zeros(len(off), dtype=np.int64)
        np.cumsum(hi - lo, out=new_off[1:])
        return IVFFlatLists(self.centroids, self.quantizer_metric, np.ascontiguousarray(self.rows[take]),
                            np.ascontiguousarray(np.asarray(self.ids)[take]), new_off, self.metric, self.nprobe)


def read_ivfflat_lists(path: str | os.PathLike) -> IVFFlatLists:
    """Parse a FAISS IndexIVFFlat file (fourcc `IwFl`) without flattening it.  Layout restated from upstream
    (faiss/impl/index_write.cpp), validated structurally and through this module's own writer (faiss is absent)."""
    path = Path(path)
    r = _Reader(path)
    if r.array(np.uint8, 4).tobytes() != b"IwFl":
        raise ValueError(f"{path}: not an IwFl file")
    d, ntotal, metric_type = r.index_header()
    if metric_type > 1 or d <= 0 or ntotal < 0:
        raise ValueError(f"{path}: unsupported IVF header (d={d}, ntotal={ntotal}, metric_type={metric_type})")
    nlist, nprobe = r.take("QQ")
    # coarse quantizer: a nested flat index holding nlist centroids
    qcc = r.array(np.uint8, 4).tobytes()
    if qcc not in _FLAT_FOURCC:
        raise ValueError(f"{path}: coarse quantizer {qcc!r} is not a flat index")
    qd, qn, q_metric_type = r.index_header()
    centroids = np.array(r.array(np.float32, r.take("Q")))
    if qd != d or qn != nlist or centroids.size != nlist * d:
        raise ValueError(f"{path}: quantizer shape ({qn}, {qd}) does not match nlist={nlist}, d={d}")
    q_metric = _FLAT_FOURCC[qcc]
    if q_metric is None:
        q_metric = METRIC_INNER_PRODUCT if q_metric_type == 0 else METRIC_L2
    # direct map: type byte, array, and (hashtable type only) a vector of (id, offset) pairs
    dm_type = r.take("B")
    r.array(np.int64, r.take("Q"))
    if dm_type == 2:
        r.array(np.int64, 2 * r.take("Q"))
    # inverted lists
    if r.array(np.uint8, 4).tobytes() != b"ilar":
        raise ValueError(f"{path}: inverted lists are not an in-memory array ('ilar')")
    il_nlist, code_size = r.take("QQ")
    if il_nlist != nlist or code_size != 4 * d:
        raise ValueError(f"{path}: list count / code size ({il_nlist}, {code_size}) do not describe raw fp32 vectors")
    list_type = r.array(np.uint8, 4).tobytes()
    raw = r.array(np.uint64, r.take("Q"))
    sizes = np.zeros(nlist, dtype=np.int64)
    if list_type == b"full":
        if raw.size != nlist:
            raise ValueError(f"{path}: 'full' size vector has {raw.size} entries, expected {nlist}")
        sizes[:] = raw
    elif list_type == b"sprs":
        if raw.size % 2:
            raise ValueError(f"{path}: 'sprs' size vector has odd length")
        pairs = raw.reshape(-1, 2)
        if pairs.size and pairs[:, 0].max() >= nlist:
            raise ValueError(f"{path}: 'sprs' list number out of range")
        sizes[pairs[:, 0].astype(np.int64)] = pairs[:, 1]
    else:
        raise ValueError(f"{path}: unknown list encoding {list_type!r}")
    if int(sizes.sum()) != ntotal:
        raise ValueError(f"{path}: inverted lists hold {int(sizes.sum())} vectors, header says {ntotal}")
    offsets = np.zeros(nlist + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    rows = np.empty((ntotal, d), dtype=np.float32)
    ids = np.empty(ntotal, dtype=np.int64)
    for l in range(nlist):
        n = int(sizes[l])
        if n == 0:
            continue
        lo = int(offsets[l])
        rows[lo:lo + n] = r.array(np.float32, n * d).reshape(n, d)
        ids[lo:lo + n] = r.array(np.int64, n)
    return IVFFlatLists(centroids.reshape(nlist, d), q_metric, rows, ids, offsets,
                        METRIC_INNER_PRODUCT if metric_type == 0 else METRIC_L2, int(nprobe))


def _read_faiss_ivfflat(path: Path) -> tuple[np.ndarray, int]:
    """The exhaustive view of an IwFl file: rows back in stored-id order (the nprobe = nlist limit)."""
    lists = read_ivfflat_lists(path)
    ntotal = lists.ntotal
    ids = lists.ids
    if ntotal and (ids.min() < 0 or ids.max() >= ntotal or np.unique(ids).size != ntotal):
        raise ValueError(f"{path}: stored ids are not a permutation of 0..ntotal-1; row order cannot be restored")
    rows = np.empty_like(lists.rows)
    rows[ids] = lists.rows
    return rows, lists.metric


def write_ivfflat_index(path: str | os.PathLike, rows: np.ndarray, nlist: int, metric: int = METRIC_L2,
                        seed: int = 0, sparse: bool = False, nprobe: int = 1, all_centroids: bool = False) -> None:
    """Write rows as a FAISS-layout IndexIVFFlat file (test helper: random centroids taken from the rows,
    nearest-centroid assignment — over the first 64 centroids, or over all of them with `all_centroids`).  Mirrors
    what scripts/create_test_docs.py produces with faiss itself."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    n, d = rows.shape
    rng = np.random.default_rng(seed)
    cent = rows[rng.choice(n, size=min(nlist, n), replace=False)] if n else np.zeros((0, d), np.float32)
    if cent.shape[0] < nlist:
        cent = np.concatenate([cent, np.zeros((nlist - cent.shape[0], d), np.float32)])
    if not n:
        assign = np.zeros(0, int)
    elif all_centroids:
        cn = (cent.astype(np.float64) ** 2).sum(1)
        assign = np.empty(n, dtype=np.int64)
        for lo in range(0, n, 8192):
            blk = rows[lo:lo + 8192].astype(np.float64)
            assign[lo:lo + 8192] = np.argmin(cn[None, :] - 2.0 * blk @ cent.astype(np.float64).T, axis=1)
    else:
        assign = np.argmin(((rows[:, None, :] - cent[None, :min(nlist, 64), :]) ** 2).sum(-1), axis=1)
    metric_type = 0 if metric == METRIC_INNER_PRODUCT else 1
    with open(path, "wb") as fh:
        fh.write(b"IwFl")
        fh.write(_HEADER.pack(d, n, 1 << 20, 1 << 20, 1, metric_type))
        fh.write(struct.pack("<QQ", nlist, nprobe))
        fh.write(b"IxF2")
        fh.write(_HEADER.pack(d, nlist, 1 << 20, 1 << 20, 1, 1))
        fh.write(struct.pack("<Q", nlist * d))
        fh.write(cent.tobytes())
        fh.write(struct.pack("<BQ", 0, 0))  # no direct map
        fh.write(b"ilar")
        fh.write(struct.pack("<QQ", nlist, 4 * d))
        order = np.argsort(assign, kind="stable")
        counts = np.bincount(assign, minlength=nlist) if n else np.zeros(nlist, dtype=np.int64)
        bounds = np.concatenate([[0], np.cumsum(counts)])
        members = [order[bounds[l]:bounds[l + 1]].astype(np.int64) for l in range(nlist)]
        if sparse:
            pairs = [(l, len(m)) for l, m in enumerate(members) if len(m)]
            fh.write(b"sprs")
            fh.write(struct.pack("<Q", 2 * len(pairs)))
            fh.write(np.array(pairs, dtype=np.uint64).tobytes())
        else:
            fh.write(b"full")
            fh.write(struct.pack("<Q", nlist))
            fh.write(np.array([len(m) for m in members], dtype=np.uint64).tobytes())
        for m in members:
            if len(m):
                fh.write(rows[m].tobytes())
                fh.write(m.tobytes())


def write_ivfflat_file(path: str | os.PathLike, rows: np.ndarray, centroids: np.ndarray, assign: np.ndarray,
                       metric: int = METRIC_L2, nprobe: int = 1, ids: np.ndarray | None = None) -> None:
    """Write a FAISS-layout IndexIVFFlat file from rows, trained centroids and one list number per row (flat L2
    quantizer, `full` list-size table, no direct map — what scripts/create_test_docs.py:83-104 produces with faiss).
    Stored ids default to the row numbers; lists hold their rows in ascending row order.  Rows are streamed list by list."""
    rows = np.asarray(rows, dtype=np.float32)
    centroids = np.ascontiguousarray(centroids, dtype=np.float32)
    n, d = rows.shape
    nlist = centroids.shape[0]
    assign = np.asarray(assign, dtype=np.int64)
    if centroids.shape[1] != d or assign.shape != (n,):
        raise ValueError("rows, centroids and assign do not fit together")
    ids = np.arange(n, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64)
    metric_type = 0 if metric == METRIC_INNER_PRODUCT else 1
    order = np.argsort(assign, kind="stable")
    counts = np.bincount(assign, minlength=nlist) if n else np.zeros(nlist, dtype=np.int64)
    bounds = np.concatenate([[0], np.cumsum(counts)])
    with open(path, "wb") as fh:
        fh.write(b"IwFl")
        fh.write(_HEADER.pack(d, n, 1 << 20, 1 << 20, 1, metric_type))
        fh.write(struct.pack("<QQ", nlist, nprobe))
        fh.write(b"IxF2")
        fh.write(_HEADER.pack(d, nlist, 1 << 20, 1 << 20, 1, 1))
        fh.write(struct.pack("<Q", nlist * d))
        fh.write(centroids.tobytes())
        fh.write(struct.pack("<BQ", 0, 0))  # no direct map
        fh.write(b"ilar")
        fh.write(struct.pack("<QQ", nlist, 4 * d))
        fh.write(b"full")
        fh.write(struct.pack("<Q", nlist))
        fh.write(counts.astype(np.uint64).tobytes())
        for l in range(nlist):
            m = order[bounds[l]:bounds[l + 1]]
            if len(m):
                fh.write(np.ascontiguousarray(rows[m], dtype=np.float32).tobytes())
                fh.write(ids[m].astype(np.int64).tobytes())


def _read_faiss_flat(path: Path, mmap: bool) -> tuple[np.ndarray, int]:
    size = path.stat().st_size
    with path.open("rb") as fh:
        fourcc = fh.read(4)
        if fourcc == b"IwFl":
            return _read_faiss_ivfflat(path)
        if fourcc in (b"IwPQ", b"IwQ4", b"IvFl", b"IwSq", b"IwSQ"):
            raise ValueError(
                f"{path}: compressed IVF index ({fourcc.decode()}); this build scans raw fp32 vectors — export "
                "them to a flat file (IxFI/IxF2, .npy, or raw fp32 + .json sidecar)")
        if fourcc not in _FLAT_FOURCC:
            raise ValueError(f"{path}: unrecognised index fourcc {fourcc!r}")
        head = fh.read(_HEADER.size)
        if len(head) != _HEADER.size:
            raise ValueError(f"{path}: truncated header")
        d, ntotal, _, _, _, metric_type = _HEADER.unpack(head)
        if metric_type > 1:
            raise ValueError(f"{path}: metric_type {metric_type} is not supported (IP and L2 only)")
        (count,) = struct.unpack("<Q", fh.read(8))
        offset = fh.tell()
    if d <= 0 or ntotal < 0 or count != ntotal * d:
        raise ValueError(f"{path}: inconsistent header (d={d}, ntotal={ntotal}, payload={count})")
    if size < offset + 4 * count:
        raise ValueError(f"{path}: payload truncated")
    metric = _FLAT_FOURCC[fourcc]
    if metric is None:
        metric = METRIC_INNER_PRODUCT if metric_type == 0 else METRIC_L2
    if mmap:
        rows = np.memmap(path, dtype=np.float32, mode="r", offset=offset, shape=(ntotal, d))
    else:
        rows = np.fromfile(path, dtype=np.float32, count=count, offset=offset).reshape(ntotal, d)
    return rows, metric


def read_index_file(path: str | os.PathLike, default_metric: int = METRIC_INNER_PRODUCT,
                    mmap: bool = False) -> tuple[np.ndarray, int]:
    """Rows and metric of an index/corpus file.  Raises FileNotFoundError / ValueError."""
    p = Path(path)
    if not p.exists():
        raise FileNotFoundError(f"FAISS index not found at {p}")
    sidecar = Path(str(p) + ".json")
    meta = json.loads(sidecar.read_text()) if sidecar.exists() else {}
    metric = metric_from_name(meta["metric"]) if "metric" in meta else default_metric
    if p.suffix == ".npy":
        rows = np.load(p, mmap_mode="r" if mmap else None)
        if rows.ndim != 2:
            raise ValueError(f"{p}: expected a 2-D array, got shape {rows.shape}")
        return rows, metric
    if "d" in meta:  # raw fp32 + sidecar
        d = int(meta["d"])
        n = int(meta.get("ntotal", p.stat().st_size // (4 * d)))
        if p.stat().st_size < 4 * n * d:
            raise ValueError(f"{p}: file smaller than ntotal * d floats")
        if mmap:
            return np.memmap(p, dtype=np.float32, mode="r", shape=(n, d)), metric
        return np.fromfile(p, dtype=np.float32, count=n * d).reshape(n, d), metric
    return _read_faiss_flat(p, mmap)
