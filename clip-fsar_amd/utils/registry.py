"""Name -> class registry with the reference's plugin semantics (reference utils/registry.py:6-66):
``@REG.register()`` keys a class by ``__name__`` and refuses duplicates; ``REG.get(name)`` returns the class or
``None``."""


class Registry(object):
    def __init__(self, table_name=""):
        self.table_name = table_name
        self._entry_map = {}

    def _register(self, name, entry):
        if type(name) is not str:
            raise AssertionError("registry keys are strings")
        if name in self._entry_map:
            raise AssertionError("{} {} already registered.".format(self.table_name, name))
        self._entry_map[name] = entry

    def register(self):
        def deco(obj):
            self._register(obj.__name__, obj)
            return obj
        return deco

    def get(self, name):
        return self._entry_map.get(name, None)

    def get_all_registered(self):
        return self._entry_map.keys()
