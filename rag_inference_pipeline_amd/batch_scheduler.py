"""Opportunistic request batching (reference src/pipeline/services/gateway/batch_scheduler.py).

Behaviour carried over, because the retrieval executor and its callers rely on it:
  * a batch is flushed when it reaches `batch_size` ("full"), when the delay timer started by its
    first request fires ("timeout"), or on stop() ("shutdown")                       (:167-288)
  * every flush starts its own task, so batches may be in flight concurrently        (:286-288)
  * an exception in process_batch_fn becomes RuntimeError("Batch processing failed") on every
    future of the batch; a result-count mismatch becomes a ValueError on every future (:299-316)
  * enqueue() before start() raises RuntimeError("BatchScheduler is not running")    (:177-179)
  * adaptive mode keeps batch_size fixed and moves the delay between min(0.01, max) and max with
    an EWMA (0.7 / 0.3) over the mean of the last 10 queue depths                    (:28-76, :120-131)
"""

from __future__ import annotations

import asyncio
import logging
import time
from collections import deque
from collections.abc import Awaitable, Callable
from dataclasses import dataclass, field
from typing import Any, Generic, TypeVar

from .telemetry import counters

logger = logging.getLogger(__name__)
T = TypeVar("T")


class AdaptiveBatchPolicy:
    """Delay policy: more queued requests -> wait longer so batches fill (reference :28-76)."""

    def __init__(self, max_batch_size: int = 16, min_delay_sec: float = 0.01,
                 max_delay_sec: float = 0.2) -> None:
        self.max_batch_size = max_batch_size
        self.min_delay_sec = min_delay_sec
        self.max_delay_sec = max_delay_sec
        self.current_delay = min_delay_sec
        self._max_history = 10
        self._recent_depths: deque[int] = deque(maxlen=self._max_history)

    def update(self, queue_depth: int) -> float:
        self._recent_depths.append(queue_depth)
        mean_depth = sum(self._recent_depths) / len(self._recent_depths)
        span = self.max_delay_sec - self.min_delay_sec
        fill = min(mean_depth / self.max_batch_size, 1.0)
        target = self.max_delay_sec if mean_depth >= self.max_batch_size else self.min_delay_sec + fill * span
        self.current_delay = 0.7 * self.current_delay + 0.3 * target
        return max(self.min_delay_sec, min(self.current_delay, self.max_delay_sec))


@dataclass
class Batch(Generic[T]):
    batch_id: int
    requests: list[Any]
    futures: list["asyncio.Future[T]"] = field(default_factory=list)
    created_at: float = field(default_factory=time.time)

    def __len__(self) -> int:
        return len(self.requests)


class BatchScheduler(Generic[T]):
    def __init__(self, batch_size: int, max_batch_delay_ms: int,
                 process_batch_fn: Callable[[Batch[T]], Awaitable[list[T]]],
                 service_name: str = "gateway", enable_adaptive: bool = False) -> None:
        self.batch_size = batch_size
        self.max_batch_delay_sec = max_batch_delay_ms / 1000.0
        self.process_batch_fn = process_batch_fn
        self.service_name = service_name
        self.enable_adaptive = enable_adaptive
        self._configured_batch_size = batch_size
        self.policy: AdaptiveBatchPolicy | None = None
        if enable_adaptive:
            self.policy = AdaptiveBatchPolicy(
                max_batch_size=batch_size,
                min_delay_sec=min(0.01, self.max_batch_delay_sec),
                max_delay_sec=self.max_batch_delay_sec,
            )
        self._pending_requests: list[Any] = []
        self._pending_futures: list[asyncio.Future[T]] = []
        self._batch_counter = 0
        self._lock = asyncio.Lock()
        self._timer_task: asyncio.Task[None] | None = None
        self._background_tasks: set[asyncio.Task[None]] = set()
        self._running = False

    async def start(self) -> None:
        self._running = True
        logger.info("BatchScheduler started: batch_size=%d, max_delay=%.3fs", self.batch_size,
                    self.max_batch_delay_sec)

    async def stop(self) -> None:
        self._running = False
        if self._timer_task is not None and not self._timer_task.done():
            self._timer_task.cancel()
            await asyncio.wait([self._timer_task])
        async with self._lock:
            if self._pending_requests:
                await self._flush_batch(reason="shutdown")
        logger.info("BatchScheduler stopped")

    async def enqueue(self, request: Any) -> T:
        if not self._running:
            raise RuntimeError("BatchScheduler is not running")
        future: asyncio.Future[T] = asyncio.get_running_loop().create_future()
        async with self._lock:
            self._pending_requests.append(request)
            self._pending_futures.append(future)
            depth = len(self._pending_requests)
            if self.enable_adaptive and self.policy is not None:
                self.max_batch_delay_sec = self.policy.update(depth)
            counters.set("pipeline_queue_depth", depth, service=self.service_name)
            if depth >= self.batch_size:
                await self._flush_batch(reason="full")
            elif depth == 1:
                self._timer_task = asyncio.create_task(self._batch_timer())
        return await future

    async def _batch_timer(self) -> None:
        try:
            await asyncio.sleep(self.max_batch_delay_sec)
            async with self._lock:
                if self._pending_requests:
                    await self._flush_batch(reason="timeout")
        except asyncio.CancelledError:
            pass  # the batch filled up first

    async def _flush_batch(self, reason: str = "unknown") -> None:
        """Caller holds self._lock."""
        if not self._pending_requests:
            return
        timer = self._timer_task
        if timer is not None and not timer.done() and timer is not asyncio.current_task():
            timer.cancel()
        counters.inc("pipeline_batch_flush_total", service=self.service_name, reason=reason)
        self._batch_counter += 1
        batch: Batch[T] = Batch(batch_id=self._batch_counter, requests=list(self._pending_requests),
                                futures=list(self._pending_futures))
        self._pending_requests.clear()
        self._pending_futures.clear()
        counters.set("pipeline_queue_depth", 0, service=self.service_name)
        logger.info("Batch %d <- %d requests (reason: %s)", batch.batch_id, len(batch), reason)
        task = asyncio.create_task(self._process_batch(batch))
        self._background_tasks.add(task)
        task.add_done_callback(self._background_tasks.discard)

    async def _process_batch(self, batch: Batch[T]) -> None:
        try:
            results = await self.process_batch_fn(batch)
        except Exception:
            logger.exception("Error processing batch %d", batch.batch_id)
            for future in batch.futures:
                if not future.done():
                    future.set_exception(RuntimeError("Batch processing failed"))
            return
        if len(results) != len(batch.futures):
            error = ValueError(f"Result count mismatch: expected {len(batch.futures)}, got {len(results)}")
            logger.error("%s", error)
            for future in batch.futures:
                if not future.done():
                    future.set_exception(error)
            return
        for future, result in zip(batch.futures, results):
            if not future.done():
                future.set_result(result)
