"""Type strings of the plugin surface (reference src/pipeline/enums.py:31-41)."""

from enum import Enum


class ComponentType(str, Enum):
    """`components[].type` values a YAML role profile may carry."""

    EMBEDDING = "embedding"
    FAISS = "faiss"
    DOCUMENT_STORE = "document_store"
    RERANKER = "reranker"
    LLM = "llm"
    SENTIMENT = "sentiment"
    TOXICITY = "toxicity"
    GATEWAY = "gateway"


# Types this package implements (the retrieval hot path and the store right behind it).
ACCELERATED_TYPES = (ComponentType.EMBEDDING, ComponentType.FAISS, ComponentType.RERANKER,
                     ComponentType.DOCUMENT_STORE)
