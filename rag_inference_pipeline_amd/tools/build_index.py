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

The ranks of one build share a BUILD ID (--build-id, else the launcher's TORCHELASTIC_RUN_ID when it is a real
id: torchrun's static rendezvous exports the literal "none" for every run, which is treated as absent): a
marker is `{"build_id", "rank", "rows", "n", "model", "field"}`, rank 0 accepts only markers of its own build
and job (same table size, model and field) whose row counts add up to the table, and no rank ever deletes
another rank's marker — so neither a rank that finished before rank 0
started nor the leftovers of a crashed earlier build can be mistaken for each other.  The sidecar of an
earlier build is removed first and the new one appears atomically, last.
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


def _read_marker(path: Path) -> dict | None:
    try:
        m = json.loads(path.read_text())
        return m if isinstance(m, dict) else None
    except (OSError, ValueError):
        return None  # absent, or another rank is writing it right now (markers appear by rename, so: absent)


def build_index(documents_dir: str, model: str, out: str, metric: str = "ip", field: str = "content",
                batch_docs: int = 4096, device: int = 0, rank: int = 0, world: int = 1,
                wait_seconds: float = 3600.0, build_id: str | None = None) -> tuple[int, int]:
    """Embed every document (ordered by doc_id, which must be 0..n-1) and write `out` + `out`.json.
    With world > 1 this process handles rows shard_range(n, rank, world) only and `build_id` (the same
    string on every rank of this build) is required.  Returns (rows, dim)."""
    from ..sharded import shard_range

    if field not in ("content", "title"):
        raise ValueError("field must be 'content' or 'title'")
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    run_id = (os.environ.get("TORCHELASTIC_RUN_ID") or "").strip()
    if run_id.lower() == "none":  # torchrun's default --rdzv-id under the static rendezvous: the same for every run
        run_id = ""
    build_id = build_id or run_id or ("single" if world == 1 else None)
    if not build_id:
        raise ValueError("a multi-rank build needs one build id shared by its ranks: pass --build-id (torchrun's "
                         "TORCHELASTIC_RUN_ID is used only when it is a real id, not the default \"none\")")
    marker = lambda r: Path(f"{out}.part{r}.done")  # noqa: E731
    marker(rank).unlink(missing_ok=True)            # only ever this rank's own marker
    if rank == 0:
        Path(str(out) + ".json").unlink(missing_ok=True)  # the rows are about to change under an old sidecar
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
    tmp = Path(f"{out}.part{rank}.done.tmp")
    job = {"n": int(n), "model": model, "field": field}
    tmp.write_text(json.dumps({"build_id": build_id, "rank": rank, "rows": done - row_lo, **job}))
    os.replace(tmp, marker(rank))                   # a marker is either absent or complete
    if rank == 0:
        deadline = time.time() + wait_seconds
        while True:
            marks = [_read_marker(marker(r)) for r in range(world)]
            mine = [m if m and m.get("build_id") == build_id and m.get("rank") == r
                    and all(m.get(key) == val for key, val in job.items()) else None for r, m in enumerate(marks)]
            if all(mine):
                break
            if time.time() > deadline:
                raise TimeoutError(f"ranks without a .done marker of build {build_id!r} after {wait_seconds:.0f} s: "
                                   f"{[r for r in range(world) if not mine[r]]}")
            time.sleep(0.05)
        written = sum(int(m["rows"]) for m in mine)
        if written != n:
            raise RuntimeError(f"the ranks of build {build_id!r} wrote {written} rows, the table has {n}")
        side_tmp = Path(str(out) + ".json.tmp")
        side_tmp.write_text(json.dumps(
            {"d": dim, "ntotal": n, "metric": metric, "model": model, "field": field, "ranks": world,
             "build_id": build_id, "seconds": round(time.time() - t0, 2)}))
        os.replace(side_tmp, str(out) + ".json")
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
    ap.add_argument("--build-id", default=None, help="one string shared by the ranks of this build "
                                                     "(default: TORCHELASTIC_RUN_ID unless it is torchrun's default \"none\"; "
                                                     "required when --world > 1 without one)")
    a = ap.parse_args()
    device = a.device if a.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    n, d = build_index(a.documents_dir, a.model, a.out, a.metric, a.field, a.batch_docs, device, a.rank, a.world,
                       build_id=a.build_id)
    print(f"rank {a.rank}/{a.world}: wrote its rows of {a.out}: {n} x {d} fp32 ({a.metric})")


if __name__ == "__main__":
    main()
