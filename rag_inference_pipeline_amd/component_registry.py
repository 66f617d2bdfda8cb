"""Name/alias registry with lifecycle hooks (reference src/pipeline/component_registry.py:9-126).

Behaviour kept from the reference because callers depend on it:
  * register() runs the load hook at once; if it raises, the component is removed again and the
    exception propagates (:57-66);
  * duplicate names and conflicting aliases are ValueErrors (:44-46, :22-29);
  * get() resolves aliases and returns None for unknown names (:80-83);
  * start_all() runs in registration order and awaits coroutine hooks, stop_all()/unload_all() run
    in reverse and swallow errors (:85-126).
"""

from __future__ import annotations

import inspect
import logging
from collections.abc import Callable
from typing import Any

logger = logging.getLogger(__name__)

_HOOKS = ("load", "start", "stop", "unload")


class ComponentRegistry:
    def __init__(self) -> None:
        self._components: dict[str, object] = {}
        self._aliases: dict[str, str] = {}
        self._lifecycle_hooks: dict[str, dict[str, Callable | None]] = {}
        self._startup_order: list[str] = []

    @property
    def components(self) -> dict[str, object]:
        return self._components

    def register_alias(self, alias: str, name: str) -> None:
        if alias in self._components:
            raise ValueError(f"Alias '{alias}' conflicts with existing component name")
        current = self._aliases.get(alias)
        if current is not None and current != name:
            raise ValueError(f"Alias '{alias}' already registered to '{current}'")
        self._aliases[alias] = name

    def register(self, name: str, component: object, load_hook: Callable | None = None,
                 start_hook: Callable | None = None, stop_hook: Callable | None = None,
                 unload_hook: Callable | None = None) -> None:
        if name in self._components:
            raise ValueError(f"Component {name} already registered")
        self._components[name] = component
        self._lifecycle_hooks[name] = dict(zip(_HOOKS, (load_hook, start_hook, stop_hook, unload_hook)))
        self._startup_order.append(name)
        if load_hook is not None:
            try:
                logger.info("Loading component: %s", name)
                load_hook()
            except Exception:
                logger.exception("Failed to load component %s", name)
                self.unregister(name)
                raise

    def unregister(self, name: str) -> None:
        if name not in self._components:
            return
        del self._components[name]
        del self._lifecycle_hooks[name]
        if name in self._startup_order:
            self._startup_order.remove(name)
        for alias in [a for a, target in self._aliases.items() if target == name]:
            del self._aliases[alias]

    def get(self, name: str) -> Any:
        return self._components.get(self._aliases.get(name, name))

    async def _run(self, names: list[str], hook_name: str, swallow: bool) -> None:
        for name in names:
            hook = self._lifecycle_hooks[name].get(hook_name)
            if hook is None:
                continue
            try:
                logger.info("%s component: %s", hook_name, name)
                result = hook()
                if inspect.isawaitable(result):
                    await result
            except Exception:
                logger.exception("Failed to %s component %s", hook_name, name)
                if not swallow:
                    raise

    async def start_all(self) -> None:
        await self._run(list(self._startup_order), "start", swallow=False)

    async def stop_all(self) -> None:
        await self._run(list(reversed(self._startup_order)), "stop", swallow=True)

    def unload_all(self) -> None:
        for name in reversed(list(self._startup_order)):
            hook = self._lifecycle_hooks[name].get("unload")
            if hook is None:
                continue
            try:
                logger.info("Unloading component: %s", name)
                hook()
            except Exception:
                logger.exception("Failed to unload component %s", name)
