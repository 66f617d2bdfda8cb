"""Request / response carriers of the retrieval path.

Field-for-field the reference's PendingRequest (src/pipeline/services/gateway/schemas.py:93-108)
and RetrievalDocument / RetrievalResponseItem (src/pipeline/services/retrieval/schemas.py:13-57).
"""

from __future__ import annotations

import base64

from pydantic import BaseModel, Field, field_validator


def _maybe_b64(v: str | bytes | None) -> bytes | None:
    return base64.b64decode(v) if isinstance(v, str) else v


class PendingRequest(BaseModel):
    request_id: str
    query: str
    embedding: list[float] | None = None
    # optional binary form of `embedding`: little-endian fp32 bytes (base64 text in JSON).  Not in the
    # reference: 32 x 768 floats cost ~6 ms to parse as JSON numbers and 0.3 ms this way.
    embedding_f32: bytes | None = None
    docs: list[dict[str, str | int | float]] | None = None
    compressed_docs: bytes | None = None
    timestamp: float = Field(..., description="Time request was received")

    @field_validator("compressed_docs", "embedding_f32", mode="before")
    @classmethod
    def _b64(cls, v: str | bytes | None) -> bytes | None:
        return _maybe_b64(v)


class RetrievalRequestItem(BaseModel):
    request_id: str
    query: str
    embedding: list[float] | None = None
    embedding_f32: bytes | None = None  # see PendingRequest

    @field_validator("embedding_f32", mode="before")
    @classmethod
    def _b64(cls, v: str | bytes | None) -> bytes | None:
        return _maybe_b64(v)


class RetrievalDocument(BaseModel):
    doc_id: int
    title: str
    content: str
    category: str = ""
    score: float


class RetrievalResponseItem(BaseModel):
    request_id: str
    docs: list[RetrievalDocument]
    compressed_docs: bytes | None = None

    @field_validator("compressed_docs", mode="before")
    @classmethod
    def _b64(cls, v: str | bytes | None) -> bytes | None:
        return base64.b64decode(v) if isinstance(v, str) else v
