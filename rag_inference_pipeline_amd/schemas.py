"""Request / response carriers of the retrieval path.

Field-for-field the reference's PendingRequest (src/pipeline/services/gateway/schemas.py:93-108)
and RetrievalDocument / RetrievalResponseItem (src/pipeline/services/retrieval/schemas.py:13-57).
"""

from __future__ import annotations

import base64

from pydantic import BaseModel, Field, field_validator


class PendingRequest(BaseModel):
    request_id: str
    query: str
    embedding: list[float] | None = None
    docs: list[dict[str, str | int | float]] | None = None
    compressed_docs: bytes | None = None
    timestamp: float = Field(..., description="Time request was received")

    @field_validator("compressed_docs", mode="before")
    @classmethod
    def _b64(cls, v: str | bytes | None) -> bytes | None:
        return base64.b64decode(v) if isinstance(v, str) else v


class RetrievalRequestItem(BaseModel):
    request_id: str
    query: str
    embedding: list[float] | None = None


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
