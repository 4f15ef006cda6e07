"""Drop-in registration: make the reference's import sites resolve to the HIP path.

The reference's training scripts bind their hot-path operators by name at import time
(SURVEY.md §8b):
    from code.utils.cluster.IID_losses import IID_loss          (cluster_sobel.py:21)
    from code.utils.cluster.transforms import sobel_process     (cluster_sobel.py:19)
    import code.archs as archs ; archs.__dict__[config.arch](config)   (:17,140)
    from code.utils.cluster.general import get_opt ; get_opt("Adam")   (:18,149)
``install()`` makes the (Python-2) reference tree importable under Python 3
(``iic_amd.py2compat.enable``), imports those reference modules and rebinds exactly these
names to the MI355X implementations; everything else in the reference keeps running as is.
It is STRICT by default: a name that cannot be rebound raises, so a run can never silently
train on the reference's own PyTorch modules.
``python -m iic_amd.run <reference script module> [args...]`` = install() + run the unchanged
script.
"""
import importlib
import sys

PATCHES = [
  # (reference module, attribute, our module, our attribute)
  ("code.utils.cluster.IID_losses", "IID_loss", "iic_amd.losses", "IID_loss"),
  ("code.utils.cluster.transforms", "sobel_process", "iic_amd.transforms", "sobel_process"),
  ("code.archs", "ClusterNet5g", "iic_amd.archs", "ClusterNet5g"),
  ("code.archs", "ClusterNet5gTwoHead", "iic_amd.archs", "ClusterNet5gTwoHead"),
  ("code.archs.cluster", "ClusterNet5g", "iic_amd.archs", "ClusterNet5g"),
  ("code.archs.cluster", "ClusterNet5gTwoHead", "iic_amd.archs", "ClusterNet5gTwoHead"),
  ("code.archs", "ClusterNet6c", "iic_amd.archs", "ClusterNet6c"),
  ("code.archs", "ClusterNet6cTwoHead", "iic_amd.archs", "ClusterNet6cTwoHead"),
  ("code.archs.cluster", "ClusterNet6c", "iic_amd.archs", "ClusterNet6c"),
  ("code.archs.cluster", "ClusterNet6cTwoHead", "iic_amd.archs", "ClusterNet6cTwoHead"),
  ("code.archs", "SegmentationNet10a", "iic_amd.archs", "SegmentationNet10a"),
  ("code.archs", "SegmentationNet10aTwoHead", "iic_amd.archs", "SegmentationNet10aTwoHead"),
  ("code.archs.segmentation", "SegmentationNet10a", "iic_amd.archs", "SegmentationNet10a"),
  ("code.archs.segmentation", "SegmentationNet10aTwoHead", "iic_amd.archs", "SegmentationNet10aTwoHead"),
  # evaluation matching (cluster_eval.py:11 binds these by name from .eval_metrics)
  ("code.utils.cluster.eval_metrics", "_original_match", "iic_amd.eval_metrics", "_original_match"),
  ("code.utils.cluster.eval_metrics", "_hungarian_match", "iic_amd.eval_metrics", "_hungarian_match"),
  ("code.utils.cluster.eval_metrics", "_acc", "iic_amd.eval_metrics", "_acc"),
  ("code.utils.cluster.cluster_eval", "_original_match", "iic_amd.eval_metrics", "_original_match"),
  ("code.utils.cluster.cluster_eval", "_hungarian_match", "iic_amd.eval_metrics", "_hungarian_match"),
  ("code.utils.cluster.cluster_eval", "_acc", "iic_amd.eval_metrics", "_acc"),
  # cluster_eval.py:10,12 and segmentation_eval.py:9 bind the loss / Sobel by name as well
  ("code.utils.cluster.cluster_eval", "IID_loss", "iic_amd.losses", "IID_loss"),
  ("code.utils.cluster.cluster_eval", "sobel_process", "iic_amd.transforms", "sobel_process"),
  ("code.utils.segmentation.segmentation_eval", "sobel_process", "iic_amd.transforms", "sobel_process"),
  ("code.utils.segmentation.IID_losses", "IID_segmentation_loss", "iic_amd.seg_losses", "IID_segmentation_loss"),
  ("code.utils.segmentation.IID_losses", "IID_segmentation_loss_uncollapsed", "iic_amd.seg_losses",
   "IID_segmentation_loss_uncollapsed"),
  # optimiser: get_opt("Adam") (general.py:5-9) -> the fused multi-tensor HIP Adam
  ("code.utils.cluster.general", "Adam", "iic_amd.optim", "Adam"),
]


def py2_shims():
  """Names the Python-2 reference uses that Python 3 dropped (SURVEY.md §8b last row)."""
  from . import py2compat
  py2compat.py2_builtins()


def install(strict=True, reference_root=None, py2=True):
  """Rebind the reference's hot-path names. Returns the list of (module, attr) patched.

  py2=True first installs the Python-2 import hook for the reference tree (found through
  `reference_root`, $IIC_REFERENCE or sys.path).  strict=False only reports names that could not
  be rebound (stderr) instead of raising -- for partial trees in tests."""
  if py2:
    from . import py2compat
    try:
      py2compat.enable(reference_root)
    except ImportError:
      if strict:
        raise
      py2compat.py2_builtins()
  done = []
  for ref_mod, attr, our_mod, our_attr in PATCHES:
    try:
      m = importlib.import_module(ref_mod)
      if not hasattr(m, attr):
        raise AttributeError("reference module %s has no attribute %s" % (ref_mod, attr))
    except Exception as e:   # reference module (or one of its deps) not importable here
      if strict:
        raise ImportError("iic_amd.install: cannot rebind %s.%s (%s: %s)" %
                          (ref_mod, attr, type(e).__name__, e)) from e
      sys.stderr.write("[iic_amd.install] skip %s.%s (%s)\n" % (ref_mod, attr, e))
      continue
    ours = getattr(importlib.import_module(our_mod), our_attr)
    setattr(m, attr, ours)
    done.append((ref_mod, attr))
  # get_opt() looks the class up in a module-level dict (general.py:5-9)
  gen = sys.modules.get("code.utils.cluster.general")
  if gen is not None and ("code.utils.cluster.general", "Adam") in done:
    table = getattr(gen, "_opt_dict", None)
    if isinstance(table, dict):
      table["Adam"] = gen.Adam
      done.append(("code.utils.cluster.general", "_opt_dict['Adam']"))
    elif strict:
      raise ImportError("iic_amd.install: code.utils.cluster.general._opt_dict not found")
  return done
