"""Plugin registry with the reference's semantics (utils/repository.py:1-13):
a dict keyed by `obj.__name__`, registration asserts uniqueness and returns the
object so it can be used as a decorator."""


class Repository(dict):
    def register(self, obj):
        name = obj.__name__
        assert name not in self, f'{name} is already registered'
        self[name] = obj
        return obj
