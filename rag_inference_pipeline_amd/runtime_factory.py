"""YAML role profile -> populated ComponentRegistry.

The component-assembly half of the reference's runtime_factory (src/pipeline/runtime_factory.py):
profile resolution (:70-126), per-component creation + lifecycle registration (:191-237) and the
default-alias rule (:49-67, :164-188).  The FastAPI application, routers and middleware the
reference builds around the registry (:128-163, :240-364) are serving infrastructure and are not
part of this package; `build_registry_from_profile` returns what a host app needs to mount them.
"""

from __future__ import annotations

import contextlib
import logging
from pathlib import Path

from .component_factory import create_component
from .component_registry import ComponentRegistry
from .config import PipelineSettings
from .enums import ComponentType
from .profile_schema import ComponentConfig, ProfileFile, load_profile_file

logger = logging.getLogger(__name__)

_DEFAULT_ALIAS: dict[ComponentType | str, str] = {
    ComponentType.EMBEDDING: "embedding_generator",
    "embedding_generator": "embedding_generator",
    ComponentType.FAISS: "faiss_store",
    "faiss_store": "faiss_store",
    ComponentType.DOCUMENT_STORE: "document_store",
    ComponentType.RERANKER: "reranker",
    ComponentType.LLM: "llm_generator",
    "llm_generator": "llm_generator",
    ComponentType.SENTIMENT: "sentiment_analyzer",
    "sentiment_analyzer": "sentiment_analyzer",
    ComponentType.TOXICITY: "toxicity_filter",
    "toxicity_filter": "toxicity_filter",
    ComponentType.GATEWAY: "orchestrator",
    "orchestrator": "orchestrator",
}

_NODE_DEFAULT_PROFILE = {0: "baseline_gateway", 1: "retriever_with_rerank", 2: "generation_no_rerank"}


def default_alias_for_type(ctype: ComponentType | str) -> str | None:
    return _DEFAULT_ALIAS.get(ctype)


def resolve_profile_path(settings: PipelineSettings, configs_dir: str | Path | None = None) -> Path | None:
    """Override path, else configs/<PIPELINE_ROLE_PROFILE>.yaml|.yml, else the node-number default."""
    if settings.role_profile_override_path:
        return Path(settings.role_profile_override_path)
    base = Path(configs_dir) if configs_dir is not None else Path.cwd() / "configs"
    name = settings.pipeline_role_profile or _NODE_DEFAULT_PROFILE.get(settings.node_number, "")
    if not name:
        return None
    path = base / f"{name}.yaml"
    return path if path.exists() else base / f"{name}.yml"


def load_role_profile(settings: PipelineSettings, configs_dir: str | Path | None = None) -> ProfileFile:
    path = resolve_profile_path(settings, configs_dir)
    if path is None or not path.exists():
        raise ValueError(
            "No valid role profile found. Please set PIPELINE_ROLE_PROFILE to a valid profile name "
            f"(e.g. 'gateway') or ROLE_PROFILE_OVERRIDE_PATH to a YAML file. Checked path: {path}")
    logger.info("Loading role profile from file: %s", path)
    return load_profile_file(path)


def initialize_component(registry: ComponentRegistry, settings: PipelineSettings, profile: ProfileFile,
                         cfg: ComponentConfig, aliases: dict[str, str]) -> object:
    """Create one component, register it with its lifecycle hooks (load() runs now), add aliases."""
    if cfg.type == ComponentType.GATEWAY.value:
        if profile.batch_size is not None:
            cfg.config["batch_size"] = profile.batch_size
        if profile.batch_timeout is not None:
            cfg.config["batch_timeout"] = profile.batch_timeout
    ctype: ComponentType | str = cfg.type
    with contextlib.suppress(ValueError):
        ctype = ComponentType(cfg.type)
    component = create_component(ctype, settings, cfg.config)
    stop_hook = getattr(component, "stop", None) or getattr(component, "close_all", None)
    registry.register(name=cfg.name, component=component, load_hook=getattr(component, "load", None),
                      start_hook=getattr(component, "start", None), stop_hook=stop_hook,
                      unload_hook=getattr(component, "unload", None))
    for alias in cfg.aliases:
        if alias in aliases:
            raise ValueError(f"Duplicate alias '{alias}' defined in component '{cfg.name}'")
        aliases[alias] = cfg.name
        registry.register_alias(alias, cfg.name)
    default_alias = default_alias_for_type(ctype)
    if default_alias and default_alias not in aliases and default_alias != cfg.name:
        aliases[default_alias] = cfg.name
        registry.register_alias(default_alias, cfg.name)
    return component


def build_registry_from_profile(settings: PipelineSettings, profile: ProfileFile | None = None,
                                configs_dir: str | Path | None = None
                                ) -> tuple[ComponentRegistry, ProfileFile, dict[str, str]]:
    """Instantiate every component of the profile, in file order; returns (registry, profile, aliases)."""
    profile = profile or load_role_profile(settings, configs_dir)
    registry = ComponentRegistry()
    aliases: dict[str, str] = {}
    logger.info("Initializing components for profile: %s", profile.name)
    for cfg in profile.components:
        initialize_component(registry, settings, profile, cfg, aliases)
    for route in profile.routes:
        for alias, target in route.component_aliases.items():
            if alias in aliases and aliases[alias] != target:
                raise ValueError(f"Duplicate alias '{alias}' defined in route '{route.prefix}'")
            aliases[alias] = target
            registry.register_alias(alias, target)
    return registry, profile, aliases


def run_follower(settings: PipelineSettings, profile: ProfileFile | None = None,
                 configs_dir: str | Path | None = None) -> int:
    """Ranks other than 0 of a one-process-per-GPU deployment (no counterpart in the reference, which is
    a single process): build the same profile, load the index shard (and the reranker, if the profile
    has one) on this rank's GPU, then serve rank 0's searches and query-sharded rerank passes until it
    unloads.  `torch.distributed` must already be initialised.  Returns the number of requests served."""
    registry, _, _ = build_registry_from_profile(settings, profile, configs_dir)
    store, reranker = registry.get("faiss_store"), registry.get("reranker")
    if store is None:
        raise RuntimeError("profile has no faiss store: nothing for a follower rank to serve")
    store.load()
    if reranker is not None:
        reranker.load()
    try:
        return store.serve_forever(reranker=reranker)
    finally:
        if reranker is not None:
            reranker.unload()
        store.unload()
