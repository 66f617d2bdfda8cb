"""Minimal stand-ins for the telemetry hooks the hot path touches.

The reference wires Prometheus collectors and a psutil stage profiler through the batch scheduler
and the retrieval executor (reference src/pipeline/telemetry/metrics.py:43-193,
telemetry/profiling.py:177-227).  Observability is out of scope for this build;
what is kept is the *stage vocabulary* — the three stage names and the flush reasons — recorded in
plain in-process counters so a host application can export them however it likes.
"""

from __future__ import annotations

import contextlib
import threading
import time
from collections import defaultdict
from collections.abc import Iterator

STAGE_EMBEDDING = "retrieval.embedding"
STAGE_FAISS_SEARCH = "retrieval.faiss_search"
STAGE_DOCUMENT_FETCH = "retrieval.document_fetch"


class StageTimers:
    """Accumulated wall time and call count per stage label (thread-safe)."""

    def __init__(self) -> None:
        self._lock = threading.Lock()
        self._total: dict[str, float] = defaultdict(float)
        self._count: dict[str, int] = defaultdict(int)

    @contextlib.contextmanager
    def track(self, stage: str) -> Iterator[None]:
        t0 = time.perf_counter()
        try:
            yield
        finally:
            dt = time.perf_counter() - t0
            with self._lock:
                self._total[stage] += dt
                self._count[stage] += 1

    def snapshot(self) -> dict[str, dict[str, float]]:
        with self._lock:
            return {s: {"seconds": self._total[s], "calls": self._count[s]} for s in self._total}

    def reset(self) -> None:
        with self._lock:
            self._total.clear()
            self._count.clear()


class Counters:
    def __init__(self) -> None:
        self._lock = threading.Lock()
        self._values: dict[tuple, float] = defaultdict(float)

    def inc(self, name: str, amount: float = 1.0, **labels: str) -> None:
        with self._lock:
            self._values[(name, tuple(sorted(labels.items())))] += amount

    def set(self, name: str, value: float, **labels: str) -> None:
        with self._lock:
            self._values[(name, tuple(sorted(labels.items())))] = value

    def get(self, name: str, **labels: str) -> float:
        with self._lock:
            return self._values.get((name, tuple(sorted(labels.items()))), 0.0)


stage_timers = StageTimers()
counters = Counters()
