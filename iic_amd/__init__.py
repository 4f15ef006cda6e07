"""iic_amd -- MI355X-native (gfx950) hot path of Invariant Information Clustering.

Host-side mirror of the reference's operator interface for the training hot path
(SURVEY.md §8b): ``IID_loss``, ``sobel_process`` and the ``ClusterNet5g*`` architectures,
implemented on hand-written HIP kernels behind the C ABI of ``libiic_hip.so``
(include/iic_hip.h).  ``iic_amd.install.install()`` registers them under the reference's
module names so its training scripts import them unchanged.

Submodules: losses, transforms, archs, optim, dist, install, geom, ops, _lib.
"""
import importlib

_LAZY = {
  "IID_loss": ("losses", "IID_loss"),
  "IID_loss_heads": ("losses", "IID_loss_heads"),
  "sobel_process": ("transforms", "sobel_process"),
  "Adam": ("optim", "Adam"),
}


def __getattr__(name):
  # lazy so that `import iic_amd` works on a box without a GPU (CPU test tier)
  if name in _LAZY:
    mod, attr = _LAZY[name]
    return getattr(importlib.import_module("." + mod, __name__), attr)
  raise AttributeError("module %r has no attribute %r" % (__name__, name))
