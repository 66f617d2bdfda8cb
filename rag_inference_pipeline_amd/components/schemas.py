"""Document carriers of the rerank interface (reference src/pipeline/components/schemas.py:9-28)."""

from pydantic import BaseModel, Field


class Document(BaseModel):
    doc_id: int = Field(..., description="Document ID")
    title: str = Field(..., description="Document title")
    content: str = Field(..., description="Document content")
    category: str = Field(default="", description="Document category")


class RerankedDocument(BaseModel):
    doc_id: int = Field(..., description="Document ID")
    title: str = Field(..., description="Document title")
    content: str = Field(..., description="Document content")
    category: str = Field(default="", description="Document category")
    score: float = Field(..., description="Reranking score")
