"""String-keyed class tables for predictors / correctors / SDEs.

Interface contract taken from the reference (src/utils/registry.py:5-36): `register(name)` is a class decorator,
`get_by_name` raises ValueError for a name nobody registered, registering a name twice replaces the class and warns,
`get_all_names` lists the names.  The message texts are part of that contract (callers and tests match on them).
"""
import warnings


class Registry:
    """One table; `kind` is the noun used in its messages ("Predictor", "Corrector", "SDE")."""

    __slots__ = ("managed_thing", "_table")

    def __init__(self, kind: str):
        self.managed_thing = kind
        self._table = {}

    def _add(self, name, cls):
        if self._table.get(name) is not None:
            warnings.warn(f"{self.managed_thing} with name '{name}' doubly registered, old class will be replaced.")
        self._table[name] = cls
        return cls

    def register(self, name: str):
        return lambda cls: self._add(name, cls)

    def get_by_name(self, name: str):
        try:
            return self._table[name]
        except KeyError:
            raise ValueError(f"{self.managed_thing} with name '{name}' unknown.") from None

    def get_all_names(self):
        return [*self._table]
