"""Name -> class registries (the plugin surface of Foundation).

Mirrors the behaviour of the reference's `Registry` (F/base/registrar.py:8-103):
case-insensitive lookup, `add` usable as a decorator, optional base-class check,
`entries` sorted.  Re-implemented here; nothing is imported from the reference.
"""


class Registry:
    def __init__(self, base_class=None):
        self.base_class = base_class
        self._names = []
        self._by_lower = {}

    def add(self, cls):
        name = getattr(cls, "name", "")
        if not name or "." in name:
            raise AssertionError("registered classes need a dot-free `name`")
        if self.base_class is not None and not issubclass(cls, self.base_class):
            raise AssertionError(
                "{} is not a subclass of {}".format(cls.__name__, self.base_class.__name__)
            )
        self._by_lower[name.lower()] = cls
        if name not in self._names:
            self._names.append(name)
        return cls

    def get(self, cls_name):
        key = cls_name.lower()
        if key not in self._by_lower:
            raise KeyError('"{}" is not a name of a registered class'.format(cls_name))
        return self._by_lower[key]

    def has(self, cls_name):
        return cls_name.lower() in self._by_lower

    @property
    def entries(self):
        return sorted(self._names)
