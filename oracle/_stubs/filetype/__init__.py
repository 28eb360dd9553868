"""Stub of `filetype` (only is_image is referenced by the reference's util.py)."""


def is_image(path):
    return str(path).lower().endswith(('.png', '.jpg', '.jpeg', '.tif', '.tiff', '.bmp'))
