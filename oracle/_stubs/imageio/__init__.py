"""Stub of `imageio`; the oracle scripts never read image files through it."""


def imread(*args, **kwargs):
    raise RuntimeError("imageio stub: image IO is not available in this environment")
