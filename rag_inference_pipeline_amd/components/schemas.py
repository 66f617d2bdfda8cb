"""Document carriers of the rerank interface (reference src/pipeline/components/schemas.py:9-28)."""

from typing import Any, Callable

from pydantic import BaseModel, Field


def fast_constructor(cls: type[BaseModel]) -> Callable[..., Any]:
    """`make(**fields)` for a pydantic model whose field values are already validated and ALL given (defaults
    included): what `cls.model_construct` does — instance, __dict__, fields-set — without its per-call walk over the
    field definitions.  A rerank batch at top-100 builds 3200 result objects; model_construct costs 2.5 us
    each (16 ms of a 60 ms batch for the two wrappings the reference does), this about 0.5 us."""
    names = frozenset(cls.model_fields)
    new, setattr_ = cls.__new__, object.__setattr__

    def make(**values: Any) -> Any:
        obj = new(cls)
        setattr_(obj, "__dict__", values)
        setattr_(obj, "__pydantic_fields_set__", set(names))
        setattr_(obj, "__pydantic_extra__", None)
        setattr_(obj, "__pydantic_private__", None)
        return obj

    return make


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
