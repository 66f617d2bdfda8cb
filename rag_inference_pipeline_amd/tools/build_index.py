"""Build a flat index file from the document database with this build's own encoder.

The reference only ever indexes random vectors (scripts/create_test_docs.py:42-50, :92-97: `doc_id`
is the row number of an `np.random` embedding).  A deployment that wants meaningful retrieval embeds
the documents' text with the same model the queries will use; this tool does that on the GPU and
writes the raw-fp32 + JSON-sidecar format `index_io.read_index_file` loads (row i = doc_id i, so the
ids the index returns are document ids, as in the reference).

    python -m rag_inference_pipeline_amd.tools.build_index --documents-dir documents/ \
        --model synthetic:all-MiniLM-L6-v2 --out faiss_index.f32 [--metric ip] [--field content]

One process per GPU: started under torchrun (RANK / WORLD_SIZE / LOCAL_RANK in the environment), or with
--rank R --world W by hand, each process embeds the contiguous doc_id range sharded.shard_range gives it
and writes its rows at their byte offset of the one output file; rank 0 writes the sidecar once every
rank has left its `.done` marker.  No collective is involved.
"""

from __future__ import annotations

import argparse
import json
import os
import sqlite3
import time
from pathlib import Path

import numpy as np

from ..components.embedding import EmbeddingGenerator
from ..config import PipelineSettings


def build_index(documents_dir: str, model: str, out: str, metric: str = "ip", field: str = "content",
                batch_docs: int = 4096, device: int = 0, rank: int = 0, world: int = 1,
                wait_seconds: float = 3600.0) -> tuple[int, int]:
    """Embed every document (ordered by doc_id, which must be 0..n-1) and write `out` + `out`.json.
    With world > 1 this process handles rows shard_range(n, rank, world) only.  Returns (rows, dim)."""
    from ..sharded import shard_range

    if field not in ("content", "title"):
        raise ValueError("field must be 'content' or 'title'")
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    db = Path(documents_dir) / "documents.db"
    if not db.exists():
        raise FileNotFoundError(f"Document database not found at {db}")
    settings = PipelineSettings(embedding_model_name=model, DISABLE_CACHE_FOR_PROFILING="true", RAG_AMD_DEVICE=str(device))
    embedder = EmbeddingGenerator(settings)
    embedder.load()
    con = sqlite3.connect(f"file:{db}?mode=ro", uri=True)
    (n,) = con.execute("SELECT COUNT(*) FROM documents").fetchone()
    lo_hi = con.execute("SELECT MIN(doc_id), MAX(doc_id) FROM documents").fetchone()
    if n and (lo_hi[0] != 0 or lo_hi[1] != n - 1):
        raise ValueError(f"doc_id must run 0..n-1 to double as the index row number (found {lo_hi[0]}..{lo_hi[1]}, n={n})")
    dim = int(embedder._model.cfg.hidden)
    row_lo, row_hi = shard_range(n, rank, world)
    marker = lambda r: Path(f"{out}.part{r}.done")  # noqa: E731
    if rank == 0:
        for r in range(world):
            marker(r).unlink(missing_ok=True)
    t0 = time.time()
    fd = os.open(out, os.O_RDWR | os.O_CREAT, 0o644)  # every rank writes its own byte range of the one file
    try:
        cur = con.execute(f"SELECT doc_id, {field} FROM documents WHERE doc_id >= ? AND doc_id < ? ORDER BY doc_id",
                          (row_lo, row_hi))
        done = row_lo
        while True:
            rows = cur.fetchmany(batch_docs)
            if not rows:
                break
            emb = np.ascontiguousarray(embedder.encode([r[1] or "" for r in rows]), dtype=np.float32)
            if emb.shape[1] != dim:
                raise RuntimeError(f"encoder returned dimension {emb.shape[1]}, expected {dim}")
            os.pwrite(fd, emb.tobytes(), done * dim * 4)
            done += len(rows)
        if rank == 0:
            os.ftruncate(fd, n * dim * 4)  # exact size even when the last ranks are still writing inside it
        os.fsync(fd)
    finally:
        os.close(fd)
    con.close()
    embedder.unload()
    marker(rank).write_text(str(row_hi - row_lo))
    if rank == 0:
        deadline = time.time() + wait_seconds
        while not all(marker(r).exists() for r in range(world)):
            if time.time() > deadline:
                raise TimeoutError(f"ranks without a .done marker after {wait_seconds:.0f} s: "
                                   f"{[r for r in range(world) if not marker(r).exists()]}")
            time.sleep(0.05)
        Path(str(out) + ".json").write_text(json.dumps(
            {"d": dim, "ntotal": n, "metric": metric, "model": model, "field": field, "ranks": world,
             "seconds": round(time.time() - t0, 2)}))
        for r in range(world):
            marker(r).unlink(missing_ok=True)
    return n, dim


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--documents-dir", required=True)
    ap.add_argument("--model", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--metric", default="ip", choices=["ip", "l2"])
    ap.add_argument("--field", default="content", choices=["content", "title"])
    ap.add_argument("--batch-docs", type=int, default=4096)
    ap.add_argument("--device", type=int, default=None, help="default: LOCAL_RANK, else 0")
    ap.add_argument("--rank", type=int, default=int(os.environ.get("RANK", "0")))
    ap.add_argument("--world", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    a = ap.parse_args()
    device = a.device if a.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    n, d = build_index(a.documents_dir, a.model, a.out, a.metric, a.field, a.batch_docs, device, a.rank, a.world)
    print(f"rank {a.rank}/{a.world}: wrote its rows of {a.out}: {n} x {d} fp32 ({a.metric})")


if __name__ == "__main__":
    main()
