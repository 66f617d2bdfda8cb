"""Small TTL + LRU cache (the role of the reference's utils/cache.py:20-118 LRUCache).

Only what the embedder's optional text cache needs: get / put / clear, capacity eviction in LRU
order, entries older than `ttl` seconds treated as missing.  Not thread-safe by itself — callers
hold their own lock, as the reference's components do (embedding.py:63, :140, :163)."""

from __future__ import annotations

import time
from collections import OrderedDict
from typing import Generic, Hashable, TypeVar

K = TypeVar("K", bound=Hashable)
V = TypeVar("V")


class LRUCache(Generic[K, V]):
    def __init__(self, capacity: int = 1000, ttl: float | None = None, name: str = "cache") -> None:
        self.capacity = max(1, int(capacity))
        self.ttl = ttl
        self.name = name
        self._data: OrderedDict[K, tuple[float, V]] = OrderedDict()
        self.hits = 0
        self.misses = 0

    def get(self, key: K) -> V | None:
        item = self._data.get(key)
        if item is None:
            self.misses += 1
            return None
        stamp, value = item
        if self.ttl is not None and time.monotonic() - stamp > self.ttl:
            del self._data[key]
            self.misses += 1
            return None
        self._data.move_to_end(key)
        self.hits += 1
        return value

    def put(self, key: K, value: V) -> None:
        self._data[key] = (time.monotonic(), value)
        self._data.move_to_end(key)
        while len(self._data) > self.capacity:
            self._data.popitem(last=False)

    def clear(self) -> None:
        self._data.clear()

    def __len__(self) -> int:
        return len(self._data)
