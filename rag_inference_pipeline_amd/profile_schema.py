"""YAML role-profile schema (reference src/pipeline/config/profile_schema.py:6-44).

The reference's configs/*.yaml load through this unchanged: same keys, same defaults, same two
validation rules (unique route prefixes; route aliases must name a declared component).
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, Literal

import yaml
from pydantic import BaseModel, Field, field_validator, model_validator


class ComponentConfig(BaseModel):
    name: str
    type: str
    config: dict[str, Any] = Field(default_factory=dict)
    aliases: list[str] = Field(default_factory=list)


class RouteConfig(BaseModel):
    target: Literal["gateway", "retrieval", "generation"]
    prefix: str = "/"
    component_aliases: dict[str, str] = Field(default_factory=dict)


class ProfileFile(BaseModel):
    name: str
    description: str = ""
    batch_size: int | None = None
    batch_timeout: float | None = None
    components: list[ComponentConfig] = Field(default_factory=list)
    routes: list[RouteConfig] = Field(default_factory=list)

    @field_validator("routes")
    @classmethod
    def _unique_prefixes(cls, routes: list[RouteConfig]) -> list[RouteConfig]:
        seen: set[str] = set()
        for route in routes:
            if route.prefix in seen:
                raise ValueError("Duplicate prefixes found in routes")
            seen.add(route.prefix)
        return routes

    @model_validator(mode="after")
    def _aliases_resolve(self) -> "ProfileFile":
        declared = {c.name for c in self.components}
        for route in self.routes:
            for alias, target in route.component_aliases.items():
                if target not in declared:
                    raise ValueError(f"Route alias '{alias}' points to unknown component '{target}'")
        return self


def load_profile_file(path: str | Path) -> ProfileFile:
    """Parse one YAML profile; any failure surfaces as ValueError("Invalid role profile file: ...")
    as in the reference (runtime_factory.py:117-126)."""
    try:
        with Path(path).open() as fh:
            data = yaml.safe_load(fh)
        return ProfileFile(**data)
    except Exception as exc:
        raise ValueError(f"Invalid role profile file: {exc}") from exc
