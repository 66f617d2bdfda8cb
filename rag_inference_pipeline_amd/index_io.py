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

IVF files written by the reference's generator (scripts/create_test_docs.py:83-104, fourcc `IwFl`)
are recognised and rejected with a clear message: this build searches exhaustively.
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


def _read_faiss_flat(path: Path, mmap: bool) -> tuple[np.ndarray, int]:
    size = path.stat().st_size
    with path.open("rb") as fh:
        fourcc = fh.read(4)
        if fourcc in (b"IwFl", b"IwPQ", b"IwQ4", b"IvFl"):
            raise ValueError(
                f"{path}: IVF index ({fourcc.decode()}); this build scans exhaustively — export the "
                "vectors to a flat file (IxFI/IxF2, .npy, or raw fp32 + .json sidecar)")
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
