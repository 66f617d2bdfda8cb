"""DocumentStore — SQLite `documents(doc_id, title, content, category)` reader.

The step right after the hot path (reference
src/pipeline/components/document_store.py:87-302).  Host-side I/O, kept deliberately small:
thread-local connections, results ordered by the requested ids, unknown ids silently dropped
(reference :276 — the caller's zip() then misaligns scores; mirrored, not fixed, so behaviour
matches), optional truncation of title/content to `truncate_length` characters (:59-84).
"""

from __future__ import annotations

import logging
import sqlite3
import threading
from pathlib import Path

from ..config import PipelineSettings

logger = logging.getLogger(__name__)


class Document:
    __slots__ = ("doc_id", "title", "content", "category")

    def __init__(self, doc_id: int, title: str, content: str, category: str | None = None) -> None:
        self.doc_id = doc_id
        self.title = title
        self.content = content
        self.category = category

    def to_dict(self) -> dict[str, str | int]:
        out: dict[str, str | int] = {"doc_id": self.doc_id, "title": self.title, "content": self.content}
        if self.category is not None:
            out["category"] = self.category
        return out

    def truncate(self, max_length: int) -> "Document":
        return Document(self.doc_id, (self.title or "")[:max_length], (self.content or "")[:max_length],
                        self.category)


class DocumentStore:
    def __init__(self, settings: PipelineSettings) -> None:
        self.settings = settings
        self.db_path = Path(settings.documents_dir) / "documents.db"
        if not self.db_path.exists():
            raise FileNotFoundError(f"Document database not found at {self.db_path}")
        self._local = threading.local()
        self._conns: list[sqlite3.Connection] = []
        self._lock = threading.Lock()

    def _connection(self) -> sqlite3.Connection:
        conn = getattr(self._local, "conn", None)
        if conn is None:
            conn = sqlite3.connect(f"file:{self.db_path}?mode=ro", uri=True, check_same_thread=False)
            conn.row_factory = sqlite3.Row
            self._local.conn = conn
            with self._lock:
                self._conns.append(conn)
        return conn

    _MAX_VARS = 900  # stay under SQLite's bound-variable limit (999 in old builds)

    def _fetch_map(self, wanted: list[int], truncate_length: int | None) -> dict[int, Document]:
        """{doc_id: Document} of the ids that exist, already truncated; one IN (...) query per 900 ids."""
        found: dict[int, Document] = {}
        cur = self._connection().cursor()
        cur.row_factory = None  # plain tuples: no sqlite3.Row per document
        cut = truncate_length
        try:
            for lo in range(0, len(wanted), self._MAX_VARS):
                part = wanted[lo:lo + self._MAX_VARS]
                marks = ",".join("?" * len(part))
                for doc_id, title, content, category in cur.execute(
                        f"SELECT doc_id, title, content, category FROM documents WHERE doc_id IN ({marks})", part):
                    title, content = title or "", content or ""
                    if cut is not None:
                        title, content = title[:cut], content[:cut]
                    found[doc_id] = Document(doc_id, title, content, category or None)
        except sqlite3.Error as exc:
            logger.exception("SQLite error fetching documents")
            raise RuntimeError(f"Failed to fetch documents: {exc}") from exc
        finally:
            cur.close()
        return found

    def fetch_documents(self, doc_ids: list[int]) -> list[Document]:
        if not doc_ids:
            return []
        wanted = [int(i) for i in doc_ids]
        found = self._fetch_map(wanted, None)
        return [found[i] for i in wanted if i in found]

    def fetch_documents_batch(self, doc_ids_batch: list[list[int]],
                              truncate_length: int | None = None) -> list[list[Document]]:
        """Documents of a whole batch, per query in the requested order, unknown ids dropped (reference
        :278-302 fans a thread pool of per-query fetches out; here the batch's distinct ids go through ONE
        pass over the table — 3200 ids at top-100 — and the per-query lists are cut from the result)."""
        lists = [[int(i) for i in ids] for ids in doc_ids_batch]
        distinct = list(dict.fromkeys(i for ids in lists for i in ids))
        found = self._fetch_map(distinct, truncate_length) if distinct else {}
        return [[found[i] for i in ids if i in found] for ids in lists]

    def close_all(self) -> None:
        with self._lock:
            for conn in self._conns:
                try:
                    conn.close()
                except sqlite3.Error:
                    pass
            self._conns.clear()
        self._local = threading.local()
