"""DeviceEmbeddings — a batch of query embeddings that has not left HBM.

The reference hands a host numpy array from its embedder to its index
(services/retrieval/api.py:351-390: `embedder.encode(...)` -> `index.search(embeddings, k)`).  With both
stages on the GPU that round trip (D2H of the embeddings, a host copy, H2D of the same bytes, two
synchronisations) costs more than the index search itself, so `EmbeddingGenerator.encode_device` returns
this handle instead and `FAISSStore.search` accepts it: the search is enqueued on the stream the encoder ran
on, and the only read-back of the batch is its ids and scores.

Duck-types the two things `FAISSStore.search` validates on an array (`ndim`, `shape`); `numpy()` is the escape
hatch for any consumer that wants the values on the host.
"""

from __future__ import annotations

from typing import Any

import numpy as np


class DeviceEmbeddings:
    ndim = 2

    def __init__(self, tensor: Any, stream: int, device: int, producer_alive: Any = None, range_word: Any = None,
                 recompute: Any = None) -> None:
        self._tensor = tensor          # (n, d) float32, C-contiguous, on `device`; filled by work enqueued on `stream`
        self.stream = int(stream)      # hipStream_t of the producer: consumers enqueue behind it or synchronise it
        self.device = int(device)
        self.shape = (int(tensor.shape[0]), int(tensor.shape[1]))
        # The block must not go back to the allocator while the producer's stream may still write to it.  The
        # stream belongs to the producer (it dies with the model handle, whose destruction waits for the device),
        # so it is not registered with torch's allocator; instead the handle remembers whether somebody has
        # waited for the stream since (`settled`) and waits itself, once, if it is dropped unconsumed.
        self._producer_alive = producer_alive or (lambda: True)
        self._settled = False
        # The encoder's GEMMs write fp32 operands as two fp16 numbers; an activation outside fp16's range voids the
        # pass and the kernels say so in this pinned word (include/rag_amd.h rag_bert_forward_device).  `recompute`
        # (set by the embedder) then yields the same embeddings through the host path, which has fp32's range.
        self._range_word = range_word
        self.recompute = recompute

    @property
    def data_ptr(self) -> int:
        return int(self._tensor.data_ptr())

    def __len__(self) -> int:
        return self.shape[0]

    def settled(self) -> None:
        """The consumer has synchronised `stream` behind this batch (rag_index_search_device_host_out does)."""
        self._settled = True

    def valid(self) -> bool:
        """After the producer's stream has been waited for: False if the pass left fp16's range (use recompute())."""
        return self._range_word is None or int(self._range_word[0]) == 0

    def _wait(self) -> None:
        if not self._settled and self._producer_alive():
            import torch

            torch.cuda.ExternalStream(self.stream, device=self.device).synchronize()
        self._settled = True

    def __del__(self) -> None:
        try:
            self._wait()
        except Exception:
            pass

    def numpy(self) -> np.ndarray:
        """The embeddings on the host (waits for the producer's stream)."""
        self._wait()
        if not self.valid() and self.recompute is not None:
            return self.recompute()
        return self._tensor.cpu().numpy()

    def astype(self, dtype: Any) -> np.ndarray:   # a consumer written for the reference's arrays still works
        return self.numpy().astype(dtype)

    def __array__(self, dtype: Any = None, copy: Any = None) -> np.ndarray:
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)
