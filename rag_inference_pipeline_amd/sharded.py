"""Corpus-sharded flat index: one process per GPU, one all-gather per batch.

The reference has no multi-device code (one process, one index: faiss_store.py:37).  The path
shards naturally — the top-k of a union is the top-k of the per-part top-k lists — so rank r of a
G-rank group owns the contiguous rows [N*r/G, N*(r+1)/G), reports GLOBAL ids (local row + base),
and a search is:

    local scan + top-k   ->   ONE all_gather of [ids | scores | flag] (12 * nq * k + 8 bytes per rank)
                         ->   merge of G sorted lists on every rank (one device kernel that
                              reads the gather's receive buffer in place and ORs the flags)

Over xGMI the gather is latency-bound (3.85 KB per rank at nq=32, k=10), so the packed buffer goes
out as a single collective rather than one per tensor.  `torch.distributed` supplies the group
(backend "nccl" = RCCL on ROCm); with the "gloo" backend the packed buffer is staged through host
memory, which is how the multi-rank path is exercised on CPU-only and single-GPU machines.

The flag.  With the two-stage search (rag_index_set_screening) a rank's local list is exact unless one
of its per-query certificates failed.  Instead of enqueueing the fp32 fallback behind every search
(two self-disabling launches and two kernel boundaries per batch), a rank runs its local search with
RAG_SEARCH_DEFER_FALLBACK: its "not final" word travels in its block of the all-gather, the merge
kernel ORs the G words into one, and only when that word is set — on every rank alike, since every
rank merges the same gathered buffer — do the ranks repeat the batch through the one-pass fp32 scan
and gather again.  The caller reads one word back with the results it was going to read anyway.

The collective (C1).  Over RCCL the step is ONE call into the C ABI — rag_index_search_gather_device: local search ->
ncclAllGather on an own communicator -> flagged merge, all on the caller's stream (include/rag_amd.h, csrc/rag_comm.hip).
torch.distributed only bootstraps it (the 128-byte unique id travels from rank 0 through the group once) and remains the
transport of the "gloo" rehearsal path; RAG_AMD_OWN_RCCL=0 keeps torch's collectives on the data path (A/B checks).
With `overlap_collective` the all-gather and the merge of batch i run on a second stream beside the local search of
batch i+1, which takes the collective's xGMI latency out of the step when two batches are in flight.

Serving (the product path: FAISSStore.search on rank 0 -> leader_search).  A request is ONE fixed-size message
`[4 x int64 head | max_batch x d fp32]` that rank 0 fills in pinned memory, uploads once and broadcasts once; followers
learn its head from a posted store (own communicator) or a 32-byte read-back and hand the query *device pointer* inside
the message buffer straight to the local search.  Round 4: the leader decides everything — a batch whose flag came
back set is re-sent as an OP_SEARCH_EXACT request — so followers never wait for a result: they take the next request
while the previous batch is still on the GPU, and rank 0 keeps up to `depth` requests in flight from as many threads
(the reference's scheduler runs batches concurrently, batch_scheduler.py:286-288).  With `overlap_collective` requests
travel on a communicator and stream of their own, so neither the broadcast nor the all-gather sits between two scans.
"""

from __future__ import annotations

import ctypes as C
import logging
import os
import threading
from typing import Any, Callable, Protocol

import numpy as np

from ._native import SEARCH_DEFER_FALLBACK, SEARCH_EXACT_ONE_PASS

logger = logging.getLogger(__name__)

OP_SHUTDOWN, OP_SEARCH, OP_RERANK, OP_SEARCH_EXACT = 0, 1, 2, 3  # first word of the leader's request head
HEAD_BYTES = 32                               # [op, nq, k, d] as int64


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Rows [lo, hi) owned by `rank`: contiguous, sizes differ by at most one row."""
    return n_total * rank // world, n_total * (rank + 1) // world


def pack_layout(nq: int, k: int) -> tuple[int, int, int]:
    """(byte offset of the score block, byte offset of the flag word, bytes per rank) of one rank's
    packed result buffer: nq*k int64 ids, nq*k fp32 scores, one uint32 "not final" word, padded to a
    multiple of 8 bytes so that every rank's id block stays 8-byte aligned in the gathered buffer."""
    n = nq * k
    return 8 * n, 12 * n, (12 * n + 4 + 7) // 8 * 8


class _LocalIndex(Protocol):
    def search_device(self, q_ptr: int, nq: int, k: int, out_scores_ptr: int, out_ids_ptr: int,
                      stream: int = 0) -> None: ...


class _Slot:
    """One set of per-(nq, k) buffers: a search in flight owns it from submit() to collect()."""

    def __init__(self, owner: "ShardedFlatIndex", nq: int, k: int) -> None:
        torch = owner._torch
        dev, world = owner.device, owner.world
        self.nq, self.k = nq, k
        s_off, f_off, nbytes = pack_layout(nq, k)
        self.s_off, self.f_off, self.nbytes = s_off, f_off, nbytes

        def views(buf: Any) -> tuple[Any, Any, Any]:
            return (buf[:s_off].view(torch.int64).view(nq, k), buf[s_off:f_off].view(torch.float32).view(nq, k),
                    buf[f_off:f_off + 4].view(torch.int32))

        self.pack = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.pack_i, self.pack_s, self.pack_f = views(self.pack)
        self.gathered = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
        # merged result in the same layout: ids | scores | OR of the ranks' flags — ONE read-back per batch
        self.res = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.out_i, self.out_s, self.out_any = views(self.res)
        on_gpu = dev.type == "cuda"
        self.res_host = torch.empty(nbytes, dtype=torch.uint8).pin_memory() if on_gpu else self.res
        self.host_i, self.host_s, self.host_any = views(self.res_host)
        self.event = torch.cuda.Event() if on_gpu else None
        if owner.backend != "nccl":
            self.pack_host = torch.empty(nbytes, dtype=torch.uint8).pin_memory() if on_gpu \
                else torch.empty(nbytes, dtype=torch.uint8)
            self.gathered_host = torch.empty(world * nbytes, dtype=torch.uint8)
        # the slot's buffers as the C ABI takes them, wrapped once (rag_index_search_gather_device is called per batch)
        self.c_args = tuple(C.c_void_p(t.data_ptr()) for t in (self.pack, self.gathered, self.out_s, self.out_i,
                                                               self.out_any, self.res_host))
        self.pending: Any = None   # the query tensor of the search in flight (kept alive for a repeat)
        self.repeats = 0           # fp32 repeats this slot has run (tests, stats)


class _Message:
    """One serving message: pinned host block (rank 0 fills it), device block (every rank receives into it)."""

    def __init__(self, owner: "ShardedFlatIndex", on_gpu: bool) -> None:
        torch = owner._torch
        self.host = torch.zeros(owner._msg_bytes, dtype=torch.uint8)
        self.event = None
        if on_gpu:
            self.host = self.host.pin_memory()
            self.dev = torch.zeros(owner._msg_bytes, dtype=torch.uint8, device=owner.device)
            self.event = torch.cuda.Event()
        else:
            self.dev = self.host
        self.head_np = self.host[:HEAD_BYTES].numpy().view(np.int64)
        self.q_np = self.host[HEAD_BYTES:].numpy().view(np.float32)
        self.q_dev = self.dev[HEAD_BYTES:].view(torch.float32)


class ShardedFlatIndex:
    """Collective wrapper around one local index per rank.

    `local` is this rank's FlatIndex (already holding its shard, id offset set to the shard base).
    `merge` merges (world, nq, k) score/id tensors into (nq, k) outputs; the default is the device
    kernel behind rag_merge_topk_packed_flagged_device.  `dim` is the index dimension (default
    `local.d`) and `max_batch` the largest batch one serving message carries (the reference's
    retrieval_batch_size, config/__init__.py:278; larger batches go out as several requests).
    """

    def __init__(self, local: _LocalIndex, metric: int = 0, device: int | str | None = None, group: Any = None,
                 merge: Callable[..., None] | None = None, dim: int | None = None, max_batch: int = 32,
                 depth: int = 2, overlap_collective: bool | None = None) -> None:
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
        self.dim = int(dim if dim is not None else getattr(local, "d"))
        self.max_batch = int(max_batch)
        self.depth = max(1, int(depth))
        self._slots: dict[tuple[int, int], list[_Slot]] = {}
        self._next: dict[tuple[int, int], int] = {}
        self.repeats = 0  # batches repeated through the fp32 scan because some rank's certificate failed
        # One request at a time on the serving channel.  The reference's scheduler starts every flushed
        # batch as its own task on a pool thread (gateway/batch_scheduler.py:286-288), so rank 0 can be
        # inside search()/rerank_batch() on several threads at once; a request is a SEQUENCE of
        # collectives (message, gather, merge, possibly a repeat, read-back of the per-shape buffers) that
        # the followers replay in order, so two of them must never interleave.  Re-entrant: leader_search
        # holds it around the search.
        self._lock = threading.RLock()
        # the serving messages: [4 x int64 head | max_batch x dim fp32], filled by rank 0, ONE broadcast per request;
        # request number s uses message s % depth on every rank (at most `depth` requests are in flight and they finish
        # in stream order, so by the time a message is reused the search that read it is done)
        # (every request has this size, whatever its op: a follower posts its receive before it knows what is coming —
        # 98 KB at d = 768 is 1-2 us of xGMI time, the price of one collective per request instead of two)
        self._msg_bytes = (HEAD_BYTES + 4 * self.max_batch * self.dim + 7) // 8 * 8
        on_gpu = self.device.type == "cuda"
        self._msgs = [_Message(self, on_gpu) for _ in range(self.depth)]
        self._req_seq = 0             # requests seen on the serving channel (every rank counts the same ones)
        self._inflight = 0            # rank 0: search requests enqueued and not yet finished
        self._cv = threading.Condition(self._lock)
        # followers over RCCL: the message arrives in device memory and only its head comes back to the host
        self._head_pin = torch.zeros(HEAD_BYTES, dtype=torch.uint8).pin_memory() \
            if on_gpu and self.backend == "nccl" else None
        # C1: the collectives of the data path on an own RCCL communicator, enqueued by the C ABI on the search's stream
        self._comm: C.c_void_p | None = None
        self._comm_stream = None      # second stream for all-gather + merge (overlap_collective)
        self._req_comm: C.c_void_p | None = None   # overlap_collective: requests on a communicator and stream of their own
        self._req_stream = None
        self._head_mirror = None      # followers: pinned [head x 4 | seq] the request's last kernel posts
        if on_gpu and self.backend == "nccl" and merge is None and hasattr(local, "search_gather_device") \
                and os.environ.get("RAG_AMD_OWN_RCCL", "1") != "0":
            self._comm = self._create_comm()
            if self._comm is not None:
                self._head_mirror = torch.zeros(8 * 5, dtype=torch.uint8).pin_memory()
        if overlap_collective is None:   # the default wherever there is a collective to overlap; RAG_AMD_COMM_OVERLAP=0 turns it off
            overlap_collective = os.environ.get("RAG_AMD_COMM_OVERLAP", "1") != "0" and self.world > 1
        if overlap_collective and self._comm is not None:
            self._comm_stream = torch.cuda.Stream(device=self.device)
            self._req_comm = self._create_comm()
            if self._req_comm is not None:
                self._req_stream = torch.cuda.Stream(device=self.device)

    def _create_comm(self) -> "C.c_void_p | None":
        """Collective.  Rank 0 draws the unique id, the group's own broadcast carries it (bootstrap only), every rank
        joins; the ranks then agree (one all-reduce) that all of them hold a communicator — otherwise all fall back to
        torch's collectives together, so that no rank is left alone inside an RCCL call."""
        from . import _native
        torch, dist = self._torch, self._dist
        lib = _native.lib()
        ok = 1
        ident = torch.zeros(128, dtype=torch.uint8)
        path = _native.rccl_library_path()
        if lib.rag_comm_runtime(path.encode() if path else None, None) != 0:
            ok = 0
            logger.warning("own RCCL communicator unavailable: %s", lib.rag_last_error().decode("utf-8", "replace"))
        if ok and self.rank == 0:
            buf = (C.c_uint8 * 128)()
            if lib.rag_comm_unique_id(buf) != 0:
                ok = 0
                logger.warning("rag_comm_unique_id failed: %s", lib.rag_last_error().decode("utf-8", "replace"))
            else:
                ident = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
        # every rank must be able to join BEFORE anyone enters ncclCommInitRank (which waits for all of them)
        ready = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(ready, op=dist.ReduceOp.MIN, group=self.group)
        if int(ready.item()) != 1:
            logger.warning("sharded index on rank %d: RCCL cannot be bound on every rank; torch.distributed collectives "
                           "stay on the data path", self.rank)
            return None
        payload = ident.to(self.device)
        dist.broadcast(payload, src=0, group=self.group)
        raw = (C.c_uint8 * 128).from_buffer_copy(bytes(payload.cpu().tolist()))
        handle = C.c_void_p()
        if lib.rag_comm_create(raw, self.rank, self.world, self.device.index or 0, C.byref(handle)) != 0:
            ok = 0
            logger.warning("rag_comm_create failed: %s", lib.rag_last_error().decode("utf-8", "replace"))
        agreed = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=self.group)
        if int(agreed.item()) == 1:
            return handle
        if handle:
            lib.rag_comm_destroy(handle)
        logger.warning("sharded index on rank %d: own RCCL communicator not established on every rank; "
                       "torch.distributed collectives stay on the data path", self.rank)
        return None

    def close(self) -> None:
        """Release the communicator (every rank, once no search is in flight)."""
        if self._comm is not None or self._req_comm is not None:
            from . import _native
            if self.device.type == "cuda":
                self._torch.cuda.synchronize(self.device)
            for comm in (self._req_comm, self._comm):
                if comm is not None:
                    _native.lib().rag_comm_destroy(comm)
            self._comm = self._req_comm = None

    @property
    def own_rccl(self) -> bool:
        """True when the all-gather and the request broadcast run on this object's own RCCL communicator."""
        return self._comm is not None

    # -- buffers ---------------------------------------------------------------------------------
    def _slot(self, nq: int, k: int) -> _Slot:
        key = (nq, k)
        ring = self._slots.get(key)
        if ring is None:
            ring = self._slots[key] = [_Slot(self, nq, k) for _ in range(self.depth)]
            self._next[key] = 0
        i = self._next[key]
        slot = ring[i]
        if slot.pending is not None:   # (checked BEFORE the cursor moves: a refused submit leaves the ring as it was)
            raise RuntimeError(f"more than {self.depth} searches of shape ({nq}, {k}) in flight: collect() the oldest "
                               "before the next submit()")
        self._next[key] = (i + 1) % len(ring)
        return slot

    def _stream(self) -> int:
        return self._torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    # -- the collective step -----------------------------------------------------------------------
    def _all_gather(self, s: _Slot) -> None:
        dist = self._dist
        if self.backend == "nccl":
            dist.all_gather_into_tensor(s.gathered, s.pack, group=self.group)
            return
        # host-staged gather (gloo): same bytes, same layout
        s.pack_host.copy_(s.pack)
        dist.all_gather_into_tensor(s.gathered_host, s.pack_host, group=self.group)
        s.gathered.copy_(s.gathered_host)

    def _local_search(self, s: _Slot, queries: Any, mode: int) -> None:
        ex = getattr(self.local, "search_device_ex", None)
        if ex is not None:
            ex(queries.data_ptr(), s.nq, s.k, s.pack_s.data_ptr(), s.pack_i.data_ptr(), mode, s.pack_f.data_ptr(),
               self._stream())  # the flag word is written in every mode (0: final)
        else:  # a local index without the two-stage choice (test doubles): its result is final
            self.local.search_device(queries.data_ptr(), s.nq, s.k, s.pack_s.data_ptr(), s.pack_i.data_ptr(),
                                     self._stream())
            s.pack_f.zero_()

    def _gather_and_merge(self, s: _Slot) -> None:
        torch = self._torch
        self._all_gather(s)
        if self._merge is not None:  # test double: unpack on the host side of the tensor API
            g = s.gathered.view(self.world, s.nbytes)
            all_i = g[:, :s.s_off].contiguous().view(torch.int64).view(self.world, s.nq, s.k)
            all_s = g[:, s.s_off:s.f_off].contiguous().view(torch.float32).view(self.world, s.nq, s.k)
            self._merge(self.metric, all_s, all_i, s.out_s, s.out_i)
            s.out_any.copy_(g[:, s.f_off:s.f_off + 4].contiguous().view(torch.int32).view(self.world).max().reshape(1))
        else:
            from .flat_index import merge_topk_packed_flagged_device

            # the merge kernel writes ids, scores and the OR-ed flag to the pinned host block as well: no read-back copy
            merge_topk_packed_flagged_device(self.device.index or 0, self.metric, self.world, s.nq, s.k,
                                             s.gathered.data_ptr(), s.nbytes, s.s_off, s.f_off, s.out_s.data_ptr(),
                                             s.out_i.data_ptr(), s.out_any.data_ptr(),
                                             s.res_host.data_ptr() if s.event is not None else 0, self._stream())
            if s.event is not None:
                s.event.record()
            return
        if s.event is not None:  # test-double merge on a GPU: ids, scores and the flag come back in one copy
            s.res_host.copy_(s.res, non_blocking=True)
            s.event.record()

    def submit(self, queries: Any, k: int) -> _Slot:
        """Collective, asynchronous: every rank passes the same (nq, d) float32 device tensor (kept alive and
        unchanged until collect()).  Enqueues local search -> all-gather -> merge -> read-back on the current
        stream and returns at once; up to `depth` searches of one shape may be in flight."""
        nq = int(queries.shape[0])
        s = self._slot(nq, int(k))
        s.pending = queries
        self._step(s, queries, SEARCH_DEFER_FALLBACK)
        return s

    def _step(self, s: _Slot, queries: Any, mode: int) -> None:
        """local search -> all-gather -> merge for one slot: ONE C-ABI call on the own communicator, else the three
        pieces with torch's collective in the middle."""
        if self._comm is None:
            self._local_search(s, queries, mode)
            self._gather_and_merge(s)
            return
        cs = self._comm_stream
        self.local.search_gather_device(self._comm, queries.data_ptr(), s.nq, s.k, int(mode), s.c_args, self._stream(),
                                        cs.cuda_stream if cs is not None else None)
        if cs is not None:
            s.event.record(cs)
        else:
            s.event.record()

    def collect(self, s: _Slot) -> tuple[Any, Any]:
        """Collective: wait for submit()'s search; if some rank's two-stage certificate failed (the same word
        on every rank) repeat the batch through the one-pass fp32 scan and gather again.  Returns the merged
        (scores, ids) device tensors — views into the slot, valid until it is reused; `slot.host_s` /
        `slot.host_i` hold the same values in pinned host memory."""
        if s.pending is None:
            raise RuntimeError("collect() without a submit() in flight on this slot")
        if s.event is not None:
            s.event.synchronize()
        if int(s.host_any[0]) != 0:
            self._step(s, s.pending, SEARCH_EXACT_ONE_PASS)
            if s.event is not None:
                s.event.synchronize()
            s.repeats += 1
            self.repeats += 1
        s.pending = None
        return s.out_s, s.out_i

    def search_tensors(self, queries: Any, k: int) -> tuple[Any, Any]:
        """Collective: every rank passes the same (nq, d) float32 device tensor; every rank gets the
        merged (scores, ids) device tensors (views into per-shape buffers, valid until `depth` further
        searches of this shape).  Waits for the result (the flag decides whether it is final); callers that
        want several batches in flight use submit() / collect()."""
        return self.collect(self.submit(queries, k))

    def search(self, queries: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
        """Host convenience (collective): numpy in, numpy out, same contract as FlatIndex.search."""
        torch = self._torch
        with self._lock:  # the per-shape buffers are shared: keep them until the result is on the host
            q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(self.device)
            s = self.submit(q, k)
            self.collect(s)
            return s.host_s.numpy().copy(), s.host_i.numpy().copy()

    # -- serving: rank 0 answers requests, the other ranks follow -------------------------------------
    # The reference's retrieval node is ONE process (uvicorn) calling index.search(); with the corpus
    # split over G ranks, rank 0 keeps that role.  Every request it serves is one fixed-size message
    # [op, nq, k, d | queries] broadcast to the followers, which sit in follower_loop(): OP_SEARCH /
    # OP_SEARCH_EXACT enqueue the collective search on the queries inside the message (two-stage with a
    # deferred fallback / one-pass fp32); other ops run a handler a component registered (the reranker's
    # query-sharded pass, components/reranker.py); OP_SHUTDOWN ends the loop.
    def _request(self, op: int, nq: int = 0, k: int = 0, d: int = 0, queries: np.ndarray | None = None) -> _Message:
        """One request on the channel (the caller holds the lock).  Rank 0 passes the head (and the batch); the other
        ranks pass nothing and receive.  Afterwards every rank holds the message in `msg.dev`; on followers the head
        is in `self._last_head`.  The search that follows is ordered behind the message on the device."""
        dist = self._dist
        self._req_seq += 1
        msg = self._msgs[self._req_seq % len(self._msgs)]
        leader = self.rank == 0
        if leader:
            if msg.event is not None and self.backend == "nccl":
                msg.event.synchronize()   # (long done: see the ring's rule above; this is the cheap proof)
            msg.head_np[:] = (op, nq, k, d)
            if queries is not None:
                msg.q_np[:nq * self.dim].reshape(nq, self.dim)[...] = queries
        if self._comm is not None:
            # upload (rank 0) + ncclBroadcast + a last kernel that posts head and sequence number to pinned host
            # memory on the followers: one C-ABI call; on a communicator and stream of its own when there is one
            from . import _native
            from .flat_index import stream_wait
            lib = _native.lib()
            comm = self._req_comm if self._req_comm is not None else self._comm
            rs = self._req_stream
            st = rs.cuda_stream if rs is not None else self._stream()
            _native.check(lib.rag_comm_request_device(
                comm, C.c_void_p(msg.host.data_ptr()) if leader else None, C.c_void_p(msg.dev.data_ptr()),
                self._msg_bytes, 0, None if leader else C.c_void_p(self._head_mirror.data_ptr()), self._req_seq,
                C.c_void_p(st)))
            if leader:
                if rs is not None:
                    msg.event.record(rs)
                else:
                    msg.event.record()
            else:   # the request's last kernel posts the head: poll one host word, no stream synchronisation
                head_c = (C.c_int64 * 4)()
                _native.check(lib.rag_comm_wait_head(C.c_void_p(self._head_mirror.data_ptr()), self._req_seq, -1, head_c))
                self._last_head = tuple(int(v) for v in head_c)
            if rs is not None:   # the search stream continues behind the message
                stream_wait(self.device.index or 0, self._stream(), st)
            return msg
        if self.backend == "nccl":
            if leader:
                msg.dev.copy_(msg.host, non_blocking=True)  # the batch is uploaded once
                msg.event.record()
            dist.broadcast(msg.dev, src=0, group=self.group)
            if not leader:   # only the 32-byte head comes back to the host
                self._head_pin.copy_(msg.dev[:HEAD_BYTES], non_blocking=True)
                self._torch.cuda.current_stream(self.device).synchronize()
                self._last_head = tuple(int(v) for v in self._head_pin.numpy().view(np.int64))
        else:
            dist.broadcast(msg.host, src=0, group=self.group)
            if msg.dev is not msg.host:
                msg.dev.copy_(msg.host)
            if not leader:
                self._last_head = tuple(int(v) for v in msg.head_np)
        return msg

    def register_handler(self, op: int, fn: Callable[[], None]) -> None:
        """Followers: run `fn()` when the leader announces `op` (fn performs the matching collectives)."""
        if op in (OP_SHUTDOWN, OP_SEARCH, OP_SEARCH_EXACT):
            raise ValueError(f"op {op} is reserved")
        self._handlers[int(op)] = fn

    def exclusive(self) -> Any:
        """`with link.exclusive():` — hold the serving channel for one whole exchange: the message and every
        collective that follows it.  leader_search() takes it itself (around each request it enqueues, not around
        the wait for the result); a component that drives its own exchange through leader_call() (the reranker)
        wraps the exchange in it."""
        return self._lock

    def leader_call(self, op: int) -> None:
        """Rank 0: put every follower into its handler for `op`; the caller then runs the same
        collectives the handler runs — inside `with link.exclusive():`."""
        if self.rank != 0:
            raise RuntimeError("leader_call() is for rank 0")
        if self.world > 1:
            with self._cv:
                # searches in flight still read their queries out of the message ring (on their own stream when requests
                # travel on a separate one): a request that is not counted in `_inflight` waits until they have drained
                while self._inflight:
                    self._cv.wait()
                self._request(int(op))

    def _leader_enqueue(self, q: np.ndarray, k: int, exact: bool, slot: _Slot | None) -> _Slot:
        """Rank 0, under the lock: one search request and its collective step, enqueued; returns its slot."""
        nq = int(q.shape[0])
        with self._cv:
            if slot is None:
                key = (nq, k)
                while True:   # a free slot of this shape and room on the channel
                    ring = self._slots.get(key)
                    free = ring is None or ring[self._next[key]].pending is None
                    if free and self._inflight < self.depth:
                        break
                    self._cv.wait()
                slot = self._slot(nq, k)
                slot.pending = q          # (taken: the device view of the queries replaces it below)
                self._inflight += 1
                fresh = True
            else:
                fresh = False
            try:
                msg = self._request(OP_SEARCH_EXACT if exact else OP_SEARCH, nq, k, self.dim, q)
                slot.pending = msg.q_dev[:nq * self.dim].view(nq, self.dim)
                self._step(slot, slot.pending, SEARCH_EXACT_ONE_PASS if exact else SEARCH_DEFER_FALLBACK)
            except BaseException:
                if fresh:   # (a repeat's slot is released by leader_search's own clean-up)
                    slot.pending = None
                    self._inflight -= 1
                    self._cv.notify_all()
                raise
        return slot

    def leader_search(self, queries: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
        """Rank 0: ship the batch in one broadcast, run the collective search, return the merged result.  Thread-safe:
        up to `depth` requests are in flight at once (the lock covers a request's enqueue, not the wait for its
        result), so two pool threads keep the GPUs busy back to back."""
        if self.rank != 0:
            raise RuntimeError("leader_search() is for rank 0; other ranks run follower_loop()")
        q = np.asarray(queries)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"queries must have shape (nq, {self.dim}), got {q.shape}")
        nq_total, k = int(q.shape[0]), int(k)
        D = np.empty((nq_total, k), dtype=np.float32)
        I = np.empty((nq_total, k), dtype=np.int64)
        for lo in range(0, nq_total, self.max_batch):
            nq = min(self.max_batch, nq_total - lo)
            s = self._leader_enqueue(q[lo:lo + nq], k, False, None)
            try:
                if s.event is not None:
                    s.event.synchronize()
                if int(s.host_any[0]) != 0:
                    # some rank's certificate failed: the same batch again as an OP_SEARCH_EXACT request (the followers
                    # do what they are told; they never look at a flag)
                    self._leader_enqueue(q[lo:lo + nq], k, True, s)
                    if s.event is not None:
                        s.event.synchronize()
                    s.repeats += 1
                    self.repeats += 1
                D[lo:lo + nq] = s.host_s.numpy()
                I[lo:lo + nq] = s.host_i.numpy()
            finally:
                with self._cv:
                    s.pending = None
                    self._inflight -= 1
                    self._cv.notify_all()
        return D, I

    def follower_loop(self) -> int:
        """Ranks > 0: serve the leader's requests until it sends shutdown(); returns the number served.  A search
        request is enqueued and left to run — the next request is taken while it is still on the GPU."""
        if self.rank == 0:
            raise RuntimeError("follower_loop() is for ranks other than 0")
        served = 0
        while True:
            with self._lock:
                msg = self._request(OP_SHUTDOWN)   # (followers pass nothing: they receive)
                op, nq, k, d = self._last_head
                if op == OP_SHUTDOWN:
                    if self.device.type == "cuda":
                        self._torch.cuda.synchronize(self.device)
                    return served
                if op in (OP_SEARCH, OP_SEARCH_EXACT):
                    if d != self.dim or not 0 < nq <= self.max_batch:
                        raise RuntimeError(f"follower received a search of {nq} x {d}; this rank serves up to "
                                           f"{self.max_batch} x {self.dim}")
                    s = self._slot_for_follower(nq, k)
                    self._step(s, msg.q_dev[:nq * d].view(nq, d),
                               SEARCH_EXACT_ONE_PASS if op == OP_SEARCH_EXACT else SEARCH_DEFER_FALLBACK)
                elif op in self._handlers:
                    self._handlers[op]()
                else:
                    raise RuntimeError(f"follower received op {op} with no handler registered")
            served += 1

    def _slot_for_follower(self, nq: int, k: int) -> _Slot:
        """The next slot of the shape's ring, without the in-flight check: the leader bounds what is in flight, and a
        slot's buffers are reused in stream order."""
        key = (nq, k)
        ring = self._slots.get(key)
        if ring is None:
            ring = self._slots[key] = [_Slot(self, nq, k) for _ in range(self.depth)]
            self._next[key] = 0
        i = self._next[key]
        self._next[key] = (i + 1) % len(ring)
        return ring[i]

    def shutdown(self) -> None:
        """Rank 0: release the followers from follower_loop() (after the requests in flight have finished)."""
        if self.rank == 0 and self.world > 1:
            with self._cv:
                while self._inflight:
                    self._cv.wait()
                self._request(OP_SHUTDOWN)
