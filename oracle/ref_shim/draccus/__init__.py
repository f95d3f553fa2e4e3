"""Stand-in for draccus.ChoiceRegistry (config plumbing the reference imports at module import time)."""


class ChoiceRegistry:
    _registry = {}

    @classmethod
    def register_subclass(cls, name, subclass=None):
        def deco(sc):
            cls._registry.setdefault(cls.__name__, {})[name] = sc
            return sc
        return deco(subclass) if subclass is not None else deco

    @classmethod
    def get_choice_class(cls, name):
        return cls._registry.get(cls.__name__, {}).get(name)


def wrap(*a, **k):
    return lambda f: f
