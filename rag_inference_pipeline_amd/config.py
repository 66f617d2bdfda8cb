"""Settings consumed by the retrieval hot path.

Mirrors the field names, defaults and environment behaviour of the reference's PipelineSettings
(reference src/pipeline/config/__init__.py:54-59, fields :106-325) for the knobs this path reads.
As in the reference, a field declared with an alias is set by its UPPER_CASE environment variable,
a field without one only by its exact lower-case name ("Gotcha").
pydantic-settings is not a dependency: the environment is read by hand.
"""

from __future__ import annotations

import os
from typing import Any

from pydantic import BaseModel, ConfigDict, Field


def _default_cpu_threads() -> int:
    # reference config/__init__.py:32-46: min(16, cores) on a dedicated node
    return min(16, os.cpu_count() or 8)


class PipelineSettings(BaseModel):
    model_config = ConfigDict(populate_by_name=True, extra="ignore")

    # -- node / profile (reference :69-79, :247-257)
    node_number: int = Field(default=1, alias="NODE_NUMBER")
    profiling_run_id: str = Field(default="default", alias="PROFILING_RUN_ID")
    pipeline_role_profile: str = Field(default="", alias="PIPELINE_ROLE_PROFILE")
    role_profile_override_path: str = Field(default="", alias="ROLE_PROFILE_OVERRIDE_PATH")

    # -- data (reference :106-116)
    faiss_index_path: str = Field(default="faiss_index.bin", alias="FAISS_INDEX_PATH")
    documents_dir: str = Field(default="documents/", alias="DOCUMENTS_DIR")

    # -- device / threads (reference :118-141)
    only_cpu: bool = Field(default=True, alias="ONLY_CPU")
    faiss_threads: int = Field(default_factory=_default_cpu_threads, alias="FAISS_THREADS")
    cpu_worker_threads: int = Field(default_factory=_default_cpu_threads, alias="CPU_WORKER_THREADS")

    # -- batching / caches (reference :143-196)
    enable_adaptive_batching: bool = Field(default=True, alias="ENABLE_ADAPTIVE_BATCHING")
    cache_max_ttl: float = Field(default=60.0, alias="CACHE_MAX_TTL")
    disable_cache_for_profiling: bool = Field(default=False, alias="DISABLE_CACHE_FOR_PROFILING")
    retrieval_cache_capacity: int = Field(default=1000, alias="RETRIEVAL_CACHE_CAPACITY")
    document_cache_capacity: int = Field(default=1000, alias="DOCUMENT_CACHE_CAPACITY")
    faiss_use_mmap: bool = Field(default=False, alias="FAISS_USE_MMAP")
    faiss_nprobe: int = Field(default=64, alias="FAISS_NPROBE")
    documents_payload_mode: str = Field(default="id_only", alias="DOCUMENTS_PAYLOAD_MODE")

    # -- un-aliased knobs (reference :226-325): settable by their lower-case name only
    faiss_dim: int = 768
    retrieval_k: int = 10
    truncate_length: int = 512
    rerank_top_n: int = 10
    retrieval_batch_size: int = 32
    retrieval_max_batch_delay_ms: int = 50
    embedding_model_name: str = "BAAI/bge-base-en-v1.5"
    reranker_model_name: str = "BAAI/bge-reranker-base"

    # -- telemetry switches the executor looks at (reference :368-392)
    enable_profiling: bool = Field(default=False, alias="ENABLE_PROFILING")
    profiling_sample_rate: float = Field(default=1.0, alias="PROFILING_SAMPLE_RATE")
    enable_torch_compile: bool = Field(default=False, alias="ENABLE_TORCH_COMPILE")

    # -- knobs that exist only in this build
    gpu_device: int = Field(default=0, alias="RAG_AMD_DEVICE")
    faiss_metric: str = Field(default="ip", alias="RAG_AMD_METRIC")  # for files that carry no metric
    # "f16" = fp16-input GEMMs for the cross-encoder, the precision the reference uses on a GPU
    # (reranker.py:91-93); "f32" (default) matches the CPU path the parity tests are held to
    reranker_dtype: str = Field(default="f32", alias="RAG_AMD_RERANKER_DTYPE")
    # two-stage exact search (include/rag_amd.h rag_index_set_screening): same results as the one-pass
    # fp32 scan at about half the HBM traffic, for +50 % index memory; applies to d <= 2048, k <= 100
    faiss_two_stage: bool = Field(default=True, alias="RAG_AMD_TWO_STAGE")
    # an IndexIVFFlat file (what the reference's generator writes): "exhaustive" scans every list — more exact than the
    # reference; "nprobe" restates the reference's search (the FAISS_NPROBE lists nearest to each query: faiss_store.py:84-92)
    faiss_ivf_mode: str = Field(default="exhaustive", alias="RAG_AMD_IVF_MODE")
    # After the first batch (every component loaded and warm) move what is alive into the collector's permanent
    # generation (gc.freeze): a full collection over a process that has imported torch / transformers / pydantic walks
    # ~10^6 objects and takes 40-60 ms — one batch in ten of a top-100 rerank profile took 65 ms instead of 20.
    gc_freeze_after_warmup: bool = Field(default=True, alias="RAG_AMD_GC_FREEZE")
    # A share of the GPU for the query encoder (0 = off; multiples of 32: one CU of every shader engine of every XCD):
    # its stream owns that many compute units and the index's search stream the rest, so that the encoder of one
    # batch runs BESIDE the corpus scan of the batch before it when the scheduler has two batches in flight
    # (include/rag_amd.h rag_stream_create_masked).  Single-GPU deployments; worth it when the scan takes longer than
    # the encoder (10M-row corpora: 32 with the one-pass scan, 64 with the two-stage search).
    encoder_cus: int = Field(default=0, alias="RAG_AMD_ENCODER_CUS")

    @classmethod
    def from_env(cls, env: dict[str, str] | None = None, **overrides: Any) -> "PipelineSettings":
        env = dict(os.environ if env is None else env)
        values: dict[str, Any] = {}
        for name, field in cls.model_fields.items():
            key = field.alias or name  # case-sensitive, like the reference
            if key in env:
                values[key] = env[key]
        values.update(overrides)
        return cls(**values)


def resolve_gpu_device(settings: Any = None) -> int:
    """The GPU this process's components live on — one rule for index, encoder and cross-encoder.

    One process, one GPU (the reference's deployment, and any group of one rank): `settings.gpu_device`
    (RAG_AMD_DEVICE, default 0).  One process per GPU under an initialised torch.distributed group of
    more than one rank: the launcher's binding — LOCAL_RANK when the launcher exports it (torchrun
    does), else torch's current device.  Without this rule every rank's cross-encoder would sit on GPU 0
    beside shard 0 while only the index followed the rank."""
    base = int(getattr(settings, "gpu_device", 0) or 0)
    try:
        import torch.distributed as dist
    except ImportError:
        return base
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return base
    import torch

    n = torch.cuda.device_count()
    local = os.environ.get("LOCAL_RANK")
    if local is not None and n > 0:
        return int(local) % n
    return torch.cuda.current_device() if n > 0 else base


_settings: PipelineSettings | None = None


def get_settings() -> PipelineSettings:
    """Process-wide settings singleton (reference config/__init__.py:487-498)."""
    global _settings
    if _settings is None:
        _settings = PipelineSettings.from_env()
    return _settings


def reset_settings() -> None:
    global _settings
    _settings = None
