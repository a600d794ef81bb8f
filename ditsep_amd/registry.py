"""Name -> class registries with the reference's semantics
(reference src/utils/registry.py:5-36: ValueError on an unknown name, a warning on re-registration)."""
import warnings
from typing import Callable


class Registry:
    def __init__(self, managed_thing: str):
        self.managed_thing = managed_thing
        self._registry = {}

    def register(self, name: str) -> Callable:
        def inner(cls):
            if name in self._registry:
                warnings.warn(f"{self.managed_thing} with name '{name}' doubly registered, old class will be replaced.")
            self._registry[name] = cls
            return cls

        return inner

    def get_by_name(self, name: str):
        if name in self._registry:
            return self._registry[name]
        raise ValueError(f"{self.managed_thing} with name '{name}' unknown.")

    def get_all_names(self):
        return list(self._registry.keys())
