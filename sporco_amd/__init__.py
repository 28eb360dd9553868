"""sporco_amd -- MI355X (gfx950) backend for SPORCO's convolutional sparse coding path.

Drop-in for the reference's ``sporco.admm.cbpdn.ConvBPDN`` / ``ConvBPDNJoint``,
``sporco.pgm.cbpdn.ConvBPDN`` and ``sporco.dictlrn.cbpdndl.ConvBPDNDictLearn``
(same class API and Options), with the per-iteration arithmetic executed by
hand-written HIP kernels reached through the C ABI of ``libsporco_amd.so``
(include/sporco_amd.h).  Host code keeps the iteration loop, adaptive-rho,
statistics and timers, exactly as the reference's Python does.

There is no CPU fallback: without the built library and an AMD GPU the
solvers raise :class:`sporco_amd.BackendError`.
"""

from ._lib import BackendError, device_count, device_info, load as load_library  # noqa: F401

__version__ = '0.1.0'
