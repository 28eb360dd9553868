"""Validated nested option dictionaries.

Same contract as the reference's ``sporco.cdict.ConstrainedDict``
(sporco/cdict.py:55-306): a class-level ``defaults`` tree fixes the allowed
keys; unknown keys raise :class:`UnknownKeyError`, replacing a sub-tree by a
non-dict raises :class:`InvalidValueError`; a tuple key addresses a nested
node (``opt['AutoRho', 'Period']``); assigning a dict to an existing node
merges instead of replacing.
"""

import pprint


def _keytext(arg):
    return ".".join(str(a) for a in arg) if isinstance(arg, (list, tuple)) else str(arg)


class UnknownKeyError(KeyError):
    """Key absent from the defaults tree."""

    def __str__(self):
        return 'Unknown dictionary key: ' + _keytext(self.args[0])

    __repr__ = __str__


class InvalidValueError(ValueError):
    """A sub-tree was assigned a non-dict value."""

    def __str__(self):
        return 'Invalid dictionary value for key: ' + _keytext(self.args[0])

    __repr__ = __str__


def _walk(tree, path, full):
    node = tree
    for key in path:
        if not isinstance(node, dict):
            raise InvalidValueError(node)
        if key not in node:
            raise UnknownKeyError(full)
        node = dict.__getitem__(node, key) if isinstance(node, ConstrainedDict) \
            else node[key]
    return node


class ConstrainedDict(dict):
    """dict whose key set (recursively) is fixed by the ``defaults`` attribute."""

    defaults = {}

    def __init__(self, d=None, pth=(), dflt=None):
        super(ConstrainedDict, self).__init__()
        self.pth = tuple(pth)
        # sub-nodes share the root's defaults tree
        self.dflt = type(self).defaults if dflt is None else dflt
        self.update(_walk(self.dflt, self.pth, self.pth))
        if d is not None:
            self.update(d)

    # -- tree helpers kept for API compatibility ---------------------------
    @staticmethod
    def getparent(d, pth):
        return _walk(d, pth[:-1], pth)

    @staticmethod
    def getnode(d, pth):
        return _walk(d, pth, pth)

    # -- mapping protocol -----------------------------------------------------
    def update(self, d):
        for key in list(d.keys()):
            self[key] = d[key]

    def _resolve(self, key):
        if isinstance(key, tuple):
            return _walk(self, key[:-1], key), key[-1]
        return self, key

    def __getitem__(self, key):
        node, last = self._resolve(key)
        if last not in node:
            raise UnknownKeyError(key)
        return dict.__getitem__(node, last)

    def __setitem__(self, key, value):
        node, last = self._resolve(key)
        plain_dict = isinstance(value, dict) and not isinstance(value, ConstrainedDict)
        if plain_dict and last in node:
            # merge into the existing sub-tree
            dict.__getitem__(node, last).update(value)
            return
        if plain_dict:
            value = ConstrainedDict(value, node.pth + (last,), self.dflt)
        node.check(last, value)
        dict.__setitem__(node, last, value)

    def check(self, key, value):
        # unpickling re-inserts items before the attributes exist
        if not hasattr(self, 'dflt'):
            return
        ref = _walk(self.dflt, self.pth, self.pth)
        if key not in ref:
            raise UnknownKeyError(self.pth + (key,))
        if isinstance(ref[key], dict) and not isinstance(value, dict):
            raise InvalidValueError(self.pth + (key,))

    def __str__(self):
        return pprint.pformat(self)


def keycmp(a, b, pth=()):
    """Raise if dict tree ``b`` has keys (recursively) that ``a`` lacks."""
    for key in b:
        if key not in a:
            raise UnknownKeyError(pth + (key,))
        if isinstance(a[key], dict):
            if not isinstance(b[key], dict):
                raise InvalidValueError(pth + (key,))
            keycmp(a[key], b[key], pth + (key,))
