"""Corpus-sharded flat index: one process per GPU, one all-gather per batch.

The reference has no multi-device code (one process, one index: faiss_store.py:37).  The path
shards naturally — the top-k of a union is the top-k of the per-part top-k lists — so rank r of a
G-rank group owns the contiguous rows [N*r/G, N*(r+1)/G), reports GLOBAL ids (local row + base),
and a search is:

    local scan + top-k   ->   ONE all_gather of [ids | scores] (12 * nq * k bytes per rank)
                         ->   merge of G sorted lists on every rank (one device kernel that
                              reads the gather's receive buffer in place)

Over xGMI the gather is latency-bound (3.84 KB per rank at nq=32, k=10), so the packed buffer goes
out as a single collective rather than one per tensor.  `torch.distributed` supplies the group
(backend "nccl" = RCCL on ROCm); with the "gloo" backend the packed buffer is staged through host
memory, which is how the multi-rank path is exercised on CPU-only and single-GPU machines.
"""

from __future__ import annotations

import threading
from typing import Any, Callable, Protocol

import numpy as np


OP_SHUTDOWN, OP_SEARCH, OP_RERANK = 0, 1, 2  # first word of the leader's request head


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Rows [lo, hi) owned by `rank`: contiguous, sizes differ by at most one row."""
    return n_total * rank // world, n_total * (rank + 1) // world


def pack_layout(nq: int, k: int) -> tuple[int, int]:
    """(byte offset of the score block, bytes per rank) of one rank's packed result buffer:
    nq*k int64 ids, then nq*k fp32 scores, padded to a multiple of 8 bytes so that every rank's id
    block stays 8-byte aligned in the gathered buffer."""
    n = nq * k
    return 8 * n, (12 * n + 7) // 8 * 8


class _LocalIndex(Protocol):
    def search_device(self, q_ptr: int, nq: int, k: int, out_scores_ptr: int, out_ids_ptr: int,
                      stream: int = 0) -> None: ...


class ShardedFlatIndex:
    """Collective wrapper around one local index per rank.

    `local` is this rank's FlatIndex (already holding its shard, id offset set to the shard base).
    `merge` merges (world, nq, k) score/id tensors into (nq, k) outputs; the default is the device
    kernel behind rag_merge_topk_device.
    """

    def __init__(self, local: _LocalIndex, metric: int = 0, device: int | str | None = None, group: Any = None,
                 merge: Callable[..., None] | None = None) -> None:
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("ShardedFlatIndex needs an initialised torch.distributed process group")
        self._torch, self._dist = torch, dist
        self.local = local
        self.metric = int(metric)
        self._handlers: dict[int, Callable[[], None]] = {}
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        if device == "cpu":  # host-only rehearsal of the collective path (tests)
            self.device = torch.device("cpu")
        else:
            self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self._merge = merge
        self._bufs: dict[tuple[int, int], dict[str, Any]] = {}
        # One request at a time on the serving channel.  The reference's scheduler starts every flushed
        # batch as its own task on a pool thread (gateway/batch_scheduler.py:286-288), so rank 0 can be
        # inside search()/rerank_batch() on several threads at once; a request is a SEQUENCE of
        # collectives (head, payload, gather, merge, read-back of the per-shape buffers) that the
        # followers replay in order, so two of them must never interleave.  Re-entrant: leader_search
        # holds it around search().
        self._lock = threading.RLock()

    # -- buffers ---------------------------------------------------------------------------------
    def _buffers(self, nq: int, k: int) -> dict[str, Any]:
        key = (nq, k)
        b = self._bufs.get(key)
        if b is None:
            torch = self._torch
            s_off, nbytes = pack_layout(nq, k)
            pack = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            b = {
                "pack": pack,
                "pack_i": pack[:s_off].view(torch.int64).view(nq, k),
                "pack_s": pack[s_off:s_off + 4 * nq * k].view(torch.float32).view(nq, k),
                "gathered": torch.empty(self.world * nbytes, dtype=torch.uint8, device=self.device),
                "out_s": torch.empty((nq, k), dtype=torch.float32, device=self.device),
                "out_i": torch.empty((nq, k), dtype=torch.int64, device=self.device),
            }
            if self.backend != "nccl":
                b["pack_host"] = torch.empty(nbytes, dtype=torch.uint8).pin_memory() \
                    if self.device.type == "cuda" else torch.empty(nbytes, dtype=torch.uint8)
                b["gathered_host"] = torch.empty(self.world * nbytes, dtype=torch.uint8)
            self._bufs[key] = b
        return b

    # -- the collective step -----------------------------------------------------------------------
    def _all_gather(self, b: dict[str, Any]) -> None:
        dist = self._dist
        if self.backend == "nccl":
            dist.all_gather_into_tensor(b["gathered"], b["pack"], group=self.group)
            return
        # host-staged gather (gloo): same bytes, same layout
        b["pack_host"].copy_(b["pack"])
        dist.all_gather_into_tensor(b["gathered_host"], b["pack_host"], group=self.group)
        b["gathered"].copy_(b["gathered_host"])

    def search_tensors(self, queries: Any, k: int) -> tuple[Any, Any]:
        """Collective: every rank passes the same (nq, d) float32 device tensor; every rank gets the
        merged (scores, ids) device tensors (views into per-shape buffers, valid until the next call)."""
        torch = self._torch
        nq = int(queries.shape[0])
        b = self._buffers(nq, k)
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        self.local.search_device(queries.data_ptr(), nq, k, b["pack_s"].data_ptr(), b["pack_i"].data_ptr(), stream)
        self._all_gather(b)
        s_off, nbytes = pack_layout(nq, k)
        if self._merge is not None:  # test double: unpack on the host side of the tensor API
            g = b["gathered"].view(self.world, nbytes)
            all_i = g[:, :s_off].contiguous().view(torch.int64).view(self.world, nq, k)
            all_s = g[:, s_off:s_off + 4 * nq * k].contiguous().view(torch.float32).view(self.world, nq, k)
            self._merge(self.metric, all_s, all_i, b["out_s"], b["out_i"])
        else:
            from .flat_index import merge_topk_packed_device

            merge_topk_packed_device(self.device.index or 0, self.metric, self.world, nq, k, b["gathered"].data_ptr(),
                                     nbytes, s_off, b["out_s"].data_ptr(), b["out_i"].data_ptr(), stream)
        return b["out_s"], b["out_i"]

    def search(self, queries: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
        """Host convenience (collective): numpy in, numpy out, same contract as FlatIndex.search."""
        torch = self._torch
        with self._lock:  # the per-shape buffers are shared: keep them until the result is on the host
            q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(self.device)
            s, i = self.search_tensors(q, k)
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            return s.cpu().numpy().copy(), i.cpu().numpy().copy()

    # -- serving: rank 0 answers requests, the other ranks follow -------------------------------------
    # The reference's retrieval node is ONE process (uvicorn) calling index.search(); with the corpus
    # split over G ranks, rank 0 keeps that role.  Every request it serves starts with a 4-word head
    # [op, nq, k, d] broadcast to the followers, which sit in follower_loop(): OP_SEARCH ships the batch
    # and runs the collective search; other ops run a handler a component registered (the reranker's
    # query-sharded pass, components/reranker.py); OP_SHUTDOWN ends the loop.
    def _ctl_device(self) -> Any:
        return self.device if self.backend == "nccl" else self._torch.device("cpu")

    def _send_head(self, op: int, nq: int = 0, k: int = 0, d: int = 0) -> None:
        head = self._torch.tensor([op, nq, k, d], dtype=self._torch.int64, device=self._ctl_device())
        self._dist.broadcast(head, src=0, group=self.group)

    def register_handler(self, op: int, fn: Callable[[], None]) -> None:
        """Followers: run `fn()` when the leader announces `op` (fn performs the matching collectives)."""
        if op in (OP_SHUTDOWN, OP_SEARCH):
            raise ValueError(f"op {op} is reserved")
        self._handlers[int(op)] = fn

    def exclusive(self) -> Any:
        """`with link.exclusive():` — hold the serving channel for one whole request: the head, every
        collective that follows it and the read-back of the result.  leader_search() takes it itself;
        a component that drives its own exchange through leader_call() (the reranker) wraps the
        exchange in it."""
        return self._lock

    def leader_call(self, op: int) -> None:
        """Rank 0: put every follower into its handler for `op`; the caller then runs the same
        collectives the handler runs — inside `with link.exclusive():`."""
        if self.rank != 0:
            raise RuntimeError("leader_call() is for rank 0")
        if self.world > 1:
            self._send_head(int(op))

    def leader_search(self, queries: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
        """Rank 0: broadcast the batch, run the collective search, return the merged result."""
        if self.rank != 0:
            raise RuntimeError("leader_search() is for rank 0; other ranks run follower_loop()")
        torch, dist = self._torch, self._dist
        q = np.ascontiguousarray(queries, dtype=np.float32)
        with self._lock:
            self._send_head(OP_SEARCH, q.shape[0], int(k), q.shape[1])
            qt = torch.from_numpy(q).to(self._ctl_device())
            dist.broadcast(qt, src=0, group=self.group)
            return self.search(q, k)

    def follower_loop(self) -> int:
        """Ranks > 0: serve the leader's requests until it sends shutdown(); returns the number served."""
        if self.rank == 0:
            raise RuntimeError("follower_loop() is for ranks other than 0")
        torch, dist = self._torch, self._dist
        served = 0
        while True:
            head = torch.zeros(4, dtype=torch.int64, device=self._ctl_device())
            dist.broadcast(head, src=0, group=self.group)
            op, nq, k, d = (int(v) for v in head.tolist())
            if op == OP_SHUTDOWN:
                return served
            if op == OP_SEARCH:
                qt = torch.empty((nq, d), dtype=torch.float32, device=self._ctl_device())
                dist.broadcast(qt, src=0, group=self.group)
                self.search(qt.cpu().numpy(), k)
            elif op in self._handlers:
                self._handlers[op]()
            else:
                raise RuntimeError(f"follower received op {op} with no handler registered")
            served += 1

    def shutdown(self) -> None:
        """Rank 0: release the followers from follower_loop()."""
        if self.rank == 0 and self.world > 1:
            with self._lock:
                self._send_head(OP_SHUTDOWN)
