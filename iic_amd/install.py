"""Drop-in registration: make the reference's import sites resolve to the HIP path.

The reference's training scripts bind their hot-path operators by name at import time
(SURVEY.md §8b):
    from code.utils.cluster.IID_losses import IID_loss          (cluster_sobel.py:21)
    from code.utils.cluster.transforms import sobel_process     (cluster_sobel.py:19)
    import code.archs as archs ; archs.__dict__[config.arch](config)   (:17,140)
``install()`` imports those reference modules (they must be importable, i.e. the reference
tree is on sys.path and its own dependencies are present) and rebinds exactly these names to
the MI355X implementations; everything else in the reference keeps running as is.
``python -m iic_amd.run <reference script module> [args...]`` = py2->py3 shims + install()
+ run the unchanged script.
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
  ("code.utils.segmentation.IID_losses", "IID_segmentation_loss", "iic_amd.seg_losses", "IID_segmentation_loss"),
  ("code.utils.segmentation.IID_losses", "IID_segmentation_loss_uncollapsed", "iic_amd.seg_losses",
   "IID_segmentation_loss_uncollapsed"),
]


def py2_shims():
  """Names the Python-2 reference uses that Python 3 dropped (SURVEY.md §8b last row)."""
  import builtins
  import itertools
  if not hasattr(builtins, "xrange"):
    builtins.xrange = range
  if not hasattr(itertools, "izip"):
    itertools.izip = zip


def install(strict=False):
  """Rebind the reference's hot-path names. Returns the list of (module, attr) patched."""
  done = []
  for ref_mod, attr, our_mod, our_attr in PATCHES:
    try:
      m = importlib.import_module(ref_mod)
    except Exception as e:   # reference module (or one of its deps) not importable here
      if strict:
        raise
      sys.stderr.write("[iic_amd.install] skip %s.%s (%s)\n" % (ref_mod, attr, e))
      continue
    ours = getattr(importlib.import_module(our_mod), our_attr)
    setattr(m, attr, ours)
    done.append((ref_mod, attr))
  return done
