"""Import shims for the *read-only* Python-2 reference at /root/reference.

Used ONLY by oracle/gen_golden.py (and optional local cross-checks) inside the
build container; /root/reference does not exist on the GPU box, so nothing in
tests marked ``gpu``, smoke() or bench.py may import this module.
Recipe: SURVEY.md Appendix A.
"""
import builtins
import importlib
import importlib.util
import os
import sys
import types

REF = os.environ.get("IIC_REFERENCE", "/root/reference")


def available():
  return os.path.isdir(os.path.join(REF, "code"))


def _shim():
  if not hasattr(builtins, "xrange"):
    builtins.xrange = range
  if "cv2" not in sys.modules:
    sys.modules["cv2"] = types.ModuleType("cv2")


def ref_cluster_losses():
  """code/utils/cluster/IID_losses.py loaded by path (pure torch)."""
  _shim()
  path = os.path.join(REF, "code/utils/cluster/IID_losses.py")
  spec = importlib.util.spec_from_file_location("ref_iid_cluster", path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def ref_seg_losses():
  """code/utils/segmentation/IID_losses.py via a package shim (relative imports)."""
  _shim()
  if "refseg" not in sys.modules:
    pkg = types.ModuleType("refseg")
    pkg.__path__ = [os.path.join(REF, "code/utils/segmentation")]
    sys.modules["refseg"] = pkg
  return importlib.import_module("refseg.IID_losses")


def ref_cluster_archs():
  """net5g / net6c / two-head modules (implicit-relative imports need sys.path)."""
  _shim()
  d = os.path.join(REF, "code/archs/cluster")
  if d not in sys.path:
    sys.path.insert(0, d)
  mods = {}
  for n in ("net5g", "net6c", "net5g_two_head", "net6c_two_head"):
    mods[n] = importlib.import_module(n)
  return mods


def ref_seg_archs():
  _shim()
  for name, sub in (("refarchs", "code/archs"), ("refarchs.cluster", "code/archs/cluster"),
                    ("refarchs.segmentation", "code/archs/segmentation")):
    if name not in sys.modules:
      pkg = types.ModuleType(name)
      pkg.__path__ = [os.path.join(REF, sub)]
      sys.modules[name] = pkg
  d = os.path.join(REF, "code/archs/segmentation")
  if d not in sys.path:
    sys.path.insert(0, d)
  return {"net10a": importlib.import_module("refarchs.segmentation.net10a")}


def ref_sobel_process():
  """sobel_process needs .cuda(); we return its source-equivalent CPU call by
  monkeypatching Tensor.cuda to identity for the duration of the call."""
  _shim()
  import torch
  if "torchvision" not in sys.modules:
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms
    sys.modules["torchvision.transforms.functional"] = tv.transforms.functional
  if "PIL" not in sys.modules:
    try:
      import PIL  # noqa
    except Exception:
      pil = types.ModuleType("PIL"); pil.Image = types.ModuleType("PIL.Image")
      sys.modules["PIL"] = pil; sys.modules["PIL.Image"] = pil.Image
  path = os.path.join(REF, "code/utils/cluster/transforms.py")
  spec = importlib.util.spec_from_file_location("ref_cluster_transforms", path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)

  def sobel_cpu(imgs, include_rgb, using_IR=False):
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
      return mod.sobel_process(imgs, include_rgb, using_IR=using_IR)
    finally:
      torch.Tensor.cuda = orig
  return sobel_cpu


def ref_eval_metrics():
  """code/utils/cluster/eval_metrics.py, executed from its source with asserts stripped
  (``assert flat_preds.is_cuda`` cannot hold in this CPU-only container; nothing else changes) and
  with a recording stand-in for ``sklearn.utils.linear_assignment_`` (removed from scikit-learn
  >= 0.23; the reference pins 0.19.1, whose ``linear_assignment`` is the Hungarian/Munkres
  algorithm).  The stand-in solves the same problem with scipy and keeps the cost matrix it was
  given, so the golden fixture pins the reference's own num_correct counts."""
  _shim()
  import numpy as np
  from scipy.optimize import linear_sum_assignment
  rec = {}

  def linear_assignment(cost):
    rec["cost"] = np.array(cost)
    r, c = linear_sum_assignment(cost)
    return np.stack([r, c], axis=1)
  fake = types.ModuleType("sklearn.utils.linear_assignment_")
  fake.linear_assignment = linear_assignment
  sys.modules["sklearn.utils.linear_assignment_"] = fake
  path = os.path.join(REF, "code/utils/cluster/eval_metrics.py")
  src = open(path).read()
  mod = types.ModuleType("ref_eval_metrics")
  mod.__file__ = path
  if not hasattr(dict, "iteritems"):
    # dict.iteritems (py2) is used on a plain dict: give the module a dict subclass-free helper
    src_exec = src.replace(".iteritems()", ".items()")   # in-memory only; the file is untouched
  else:
    src_exec = src
  exec(compile(src_exec, path, "exec", optimize=1), mod.__dict__)
  mod._recorded = rec
  return mod
