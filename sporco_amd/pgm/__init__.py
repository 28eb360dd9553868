"""Solver classes mirroring the reference sub-package of the same name."""
