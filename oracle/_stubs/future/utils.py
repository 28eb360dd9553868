PY2 = False
PY3 = True


def with_metaclass(meta, *bases):
    """Create a temporary base class with the given metaclass (py3 only)."""
    class _Tmp(meta):
        def __new__(cls, name, this_bases, d):
            return meta(name, bases, d)
    return type.__new__(_Tmp, 'temporary_class', (), {})
