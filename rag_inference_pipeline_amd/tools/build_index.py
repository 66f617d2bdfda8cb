"""Build a flat index file from the document database with this build's own encoder.

The reference only ever indexes random vectors (scripts/create_test_docs.py:42-50, :92-97: `doc_id`
is the row number of an `np.random` embedding).  A deployment that wants meaningful retrieval embeds
the documents' text with the same model the queries will use; this tool does that on the GPU and
writes the raw-fp32 + JSON-sidecar format `index_io.read_index_file` loads (row i = doc_id i, so the
ids the index returns are document ids, as in the reference).

    python -m rag_inference_pipeline_amd.tools.build_index --documents-dir documents/ \
        --model synthetic:all-MiniLM-L6-v2 --out faiss_index.f32 [--metric ip] [--field content]
"""

from __future__ import annotations

import argparse
import json
import sqlite3
import time
from pathlib import Path

import numpy as np

from ..components.embedding import EmbeddingGenerator
from ..config import PipelineSettings


def build_index(documents_dir: str, model: str, out: str, metric: str = "ip", field: str = "content",
                batch_docs: int = 4096, device: int = 0) -> tuple[int, int]:
    """Embed every document (ordered by doc_id, which must be 0..n-1) and write `out` + `out`.json.
    Returns (rows, dim)."""
    if field not in ("content", "title"):
        raise ValueError("field must be 'content' or 'title'")
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
    dim = 0
    t0 = time.time()
    with open(out, "wb") as fh:
        cur = con.execute(f"SELECT doc_id, {field} FROM documents ORDER BY doc_id")
        done = 0
        while True:
            rows = cur.fetchmany(batch_docs)
            if not rows:
                break
            emb = embedder.encode([r[1] or "" for r in rows])
            dim = emb.shape[1]
            np.ascontiguousarray(emb, dtype=np.float32).tofile(fh)
            done += len(rows)
    con.close()
    embedder.unload()
    Path(str(out) + ".json").write_text(json.dumps(
        {"d": dim, "ntotal": n, "metric": metric, "model": model, "field": field, "seconds": round(time.time() - t0, 2)}))
    return n, dim


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--documents-dir", required=True)
    ap.add_argument("--model", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--metric", default="ip", choices=["ip", "l2"])
    ap.add_argument("--field", default="content", choices=["content", "title"])
    ap.add_argument("--batch-docs", type=int, default=4096)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    n, d = build_index(a.documents_dir, a.model, a.out, a.metric, a.field, a.batch_docs, a.device)
    print(f"wrote {a.out}: {n} x {d} fp32 ({a.metric})")


if __name__ == "__main__":
    main()
