"""YAML role-profile schema (reference src/pipeline/config/profile_schema.py:6-44).

The reference's configs/*.yaml load through this unchanged: same keys, same defaults, same two
validation rules (unique route prefixes; route aliases must name a declared component), same messages.
"""

from __future__ import annotations

from collections import Counter
from pathlib import Path
from typing import Any, Literal

import yaml
from pydantic import BaseModel, Field, model_validator

ServiceTarget = Literal["gateway", "retrieval", "generation"]


class RouteConfig(BaseModel):
    """One mounted service: which router, under which URL prefix, with which alias -> component map."""

    component_aliases: dict[str, str] = Field(default_factory=dict)
    prefix: str = "/"
    target: ServiceTarget


class ComponentConfig(BaseModel):
    """One component instance: registry name, factory type string, factory options, extra aliases."""

    aliases: list[str] = Field(default_factory=list)
    config: dict[str, Any] = Field(default_factory=dict)
    type: str
    name: str


class ProfileFile(BaseModel):
    routes: list[RouteConfig] = Field(default_factory=list)
    components: list[ComponentConfig] = Field(default_factory=list)
    batch_timeout: float | None = None   # None: the settings' gateway batch timeout applies
    batch_size: int | None = None        # None: the settings' gateway batch size applies
    description: str = ""
    name: str

    @model_validator(mode="after")
    def _cross_checks(self) -> "ProfileFile":
        repeated = [p for p, n in Counter(r.prefix for r in self.routes).items() if n > 1]
        if repeated:
            raise ValueError("Duplicate prefixes found in routes")
        known = frozenset(c.name for c in self.components)
        dangling = [(alias, comp) for r in self.routes for alias, comp in r.component_aliases.items()
                    if comp not in known]
        if dangling:
            alias, comp = dangling[0]
            raise ValueError(f"Route alias '{alias}' points to unknown component '{comp}'")
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
