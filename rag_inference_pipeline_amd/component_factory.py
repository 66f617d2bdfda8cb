"""Type-string -> factory map of the plugin surface (reference src/pipeline/component_factory.py).

`create_component(type, settings, config)` resolves exactly as the reference does (:81-104): the
key as given, then the ComponentType whose value equals the string, else
ValueError("Unknown component type: ...").  The factories for the retrieval path build this
package's HIP-backed components; like the reference's they ignore `config` (:20-35).

Types outside the accelerated path (llm, sentiment, toxicity, gateway) have no factory here: a host
application that wants to instantiate a whole reference profile registers its own with
`register_factory` — the LLM stage runs on stock PyTorch-ROCm and is not this package's business.
"""

from __future__ import annotations

from collections.abc import Callable
from typing import Any

from .config import PipelineSettings
from .enums import ComponentType

ComponentFactory = Callable[[PipelineSettings, dict[str, Any]], Any]


def create_embedding_generator(settings: PipelineSettings, _config: dict[str, Any]) -> Any:
    from .components.embedding import EmbeddingGenerator

    return EmbeddingGenerator(settings)


def create_faiss_store(settings: PipelineSettings, _config: dict[str, Any]) -> Any:
    from .components.faiss_store import FAISSStore

    return FAISSStore(settings)


def create_document_store(settings: PipelineSettings, _config: dict[str, Any]) -> Any:
    from .components.document_store import DocumentStore

    return DocumentStore(settings)


def create_reranker(settings: PipelineSettings, _config: dict[str, Any]) -> Any:
    from .components.reranker import Reranker

    return Reranker(settings)


COMPONENT_FACTORIES: dict[ComponentType | str, ComponentFactory] = {
    ComponentType.EMBEDDING: create_embedding_generator,
    "embedding_generator": create_embedding_generator,
    ComponentType.FAISS: create_faiss_store,
    "faiss_store": create_faiss_store,
    ComponentType.DOCUMENT_STORE: create_document_store,
    ComponentType.RERANKER: create_reranker,
}


def register_factory(component_type: ComponentType | str, factory: ComponentFactory) -> None:
    """Plug in a factory for a type this package does not implement (or override one)."""
    COMPONENT_FACTORIES[component_type] = factory


def create_component(component_type: ComponentType | str, settings: PipelineSettings,
                     config: dict[str, Any] | None = None) -> object:
    config = {} if config is None else config
    factory = COMPONENT_FACTORIES.get(component_type)
    if factory is None and isinstance(component_type, str):
        try:
            factory = COMPONENT_FACTORIES.get(ComponentType(component_type))
        except ValueError:
            factory = None
    if factory is None:
        raise ValueError(f"Unknown component type: {component_type}")
    return factory(settings, config)
