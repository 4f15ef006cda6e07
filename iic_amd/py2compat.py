"""Run the reference's UNCHANGED Python-2 sources under Python 3 (last row).

xu-ji/IIC is Python 2.7 / torch 0.4.1 code; this image has Python 3.10 only.  Its training
scripts are the drop-in boundary of this repo ("code/scripts/cluster and code/scripts/segmentation
drive it unchanged"), so they must import and run *as they are on disk*.  ``enable(root)``
installs a ``sys.meta_path`` finder for the reference's top-level package ``code`` (which also
shadows the stdlib module of that name) whose loader translates every module IN MEMORY at import
time -- nothing is written next to the sources, no byte-code cache is produced:

  1. lib2to3 fixers (stdlib): ``print`` statements, implicit relative imports
     (code/archs/__init__.py:1-3, net5g.py:3 ...), ``dict.iteritems`` (general.py:59,
     eval_metrics.py:26), ``xrange``, ``itertools.izip``, ``has_key``, ``basestring`` ...
  2. an AST pass that restores Python-2 ``/`` semantics (int / int floors) wherever the module
     does not import ``division`` from ``__future__`` -- cluster_sobel_twohead.py:122 computes the
     dataloader batch size with it.  ``x /= n`` stays an in-place operation for tensors.

Third-party modules the reference imports but this image lacks are provided only when the real
import fails: ``sklearn.utils.linear_assignment_`` (removed in scikit-learn 0.23; the same optimal
assignment solved by scipy), and import-only stand-ins for ``cv2`` / ``torchvision`` whose
attributes raise on use (they belong to the data layer, which is outside this repo's scope).
"""
import ast
import builtins
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import itertools
import operator
import numbers
import os
import sys
import types
import warnings

_PKG = "code"

_FIXERS = ["lib2to3.fixes.fix_" + n for n in (
  "print", "import", "dict", "xrange", "itertools", "itertools_imports", "has_key", "basestring",
  "unicode", "long", "raise", "except", "exec", "ne", "repr", "numliterals", "next", "reduce",
  "raw_input", "zip", "map", "filter", "funcattrs", "methodattrs", "idioms_safe")]
_tool = None


def _refactor(src, path):
  global _tool
  if _tool is None:
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      from lib2to3 import refactor
      avail = set(refactor.get_fixers_from_package("lib2to3.fixes"))
      _tool = refactor.RefactoringTool([f for f in _FIXERS if f in avail])
  if not src.endswith("\n"):
    src += "\n"
  return str(_tool.refactor_string(src, path))


def _int_kind(v):
  """0: not an integer operand; 1: python / numpy integer scalar or integer ndarray; 2: integer torch tensor."""
  if isinstance(v, bool):
    return 1
  if isinstance(v, numbers.Integral):
    return 1
  mod = type(v).__module__
  if mod == "numpy" or mod.startswith("numpy."):
    dt = getattr(v, "dtype", None)
    return 1 if dt is not None and dt.kind in "iub" else 0
  if mod == "torch" or mod.startswith("torch."):
    isf = getattr(v, "is_floating_point", None)
    if isf is None:
      return 0
    return 0 if (isf() or v.is_complex()) else 2
  return 0


def py2div(a, b):
  """Python-2 ``a / b`` without ``from __future__ import division`` (the reference's modules have none):
  floor division for two ints -- python ints, numpy integer scalars and integer ndarrays (numpy's classic
  division) --, C-style truncating division for integer torch tensors (torch 0.4.1's integer `/`), true
  division otherwise."""
  ka, kb = _int_kind(a), _int_kind(b)
  if ka and kb:
    if ka == 2 or kb == 2:
      import torch
      return torch.div(a, b, rounding_mode="trunc")
    return a // b
  return a / b


def py2idiv(a, b):
  """Python-2 ``a /= b`` (in place where the type supports it, e.g. torch tensors)."""
  ka, kb = _int_kind(a), _int_kind(b)
  if ka and kb:
    if ka == 2:
      return a.div_(b, rounding_mode="trunc")
    if kb == 2:
      import torch
      return torch.div(a, b, rounding_mode="trunc")
    if hasattr(a, "__ifloordiv__"):
      return operator.ifloordiv(a, b)
    return a // b
  return operator.itruediv(a, b)


class _Py2Division(ast.NodeTransformer):
  def visit_BinOp(self, node):
    self.generic_visit(node)
    if isinstance(node.op, ast.Div):
      return ast.copy_location(ast.Call(
        func=ast.Name(id="__iic_py2div__", ctx=ast.Load()), args=[node.left, node.right],
        keywords=[]), node)
    return node

  def visit_AugAssign(self, node):
    self.generic_visit(node)
    if isinstance(node.op, ast.Div):
      load = _as_load(node.target)
      return ast.copy_location(ast.Assign(
        targets=[node.target],
        value=ast.Call(func=ast.Name(id="__iic_py2idiv__", ctx=ast.Load()),
                       args=[load, node.value], keywords=[])), node)
    return node


def _as_load(target):
  t = ast.parse(ast.unparse(target), mode="eval").body
  return t


def _has_future_division(tree):
  for n in tree.body:
    if isinstance(n, ast.ImportFrom) and n.module == "__future__":
      if any(a.name == "division" for a in n.names):
        return True
  return False


def translate(src, path):
  """Python-2 module source -> Python-3 code object (in memory)."""
  src3 = _refactor(src, path)
  tree = ast.parse(src3, path)
  if not _has_future_division(tree):
    tree = _Py2Division().visit(tree)
    ast.fix_missing_locations(tree)
  return compile(tree, path, "exec", dont_inherit=True)


class _Py2Loader(importlib.abc.Loader):
  def __init__(self, fullname, path, is_pkg):
    self.fullname, self.path, self.is_pkg = fullname, path, is_pkg

  def create_module(self, spec):
    return None

  def get_filename(self, fullname=None):
    return self.path

  def get_source(self, fullname=None):
    with open(self.path, "r") as f:
      return f.read()

  def exec_module(self, module):
    code = translate(self.get_source(), self.path)
    module.__dict__["__iic_py2div__"] = py2div
    module.__dict__["__iic_py2idiv__"] = py2idiv
    exec(code, module.__dict__)


class Py2ReferenceFinder(importlib.abc.MetaPathFinder):
  """Resolves ``code`` and ``code.*`` from <root>/code, translating on load."""

  def __init__(self, root):
    self.root = os.path.abspath(root)

  def find_spec(self, fullname, path=None, target=None):
    if fullname != _PKG and not fullname.startswith(_PKG + "."):
      return None
    base = os.path.join(self.root, *fullname.split("."))
    if os.path.isdir(base) and os.path.exists(os.path.join(base, "__init__.py")):
      p = os.path.join(base, "__init__.py")
      return importlib.util.spec_from_file_location(
        fullname, p, loader=_Py2Loader(fullname, p, True), submodule_search_locations=[base])
    if os.path.exists(base + ".py"):
      return importlib.util.spec_from_file_location(
        fullname, base + ".py", loader=_Py2Loader(fullname, base + ".py", False))
    return None


def run_script(modname, root=None, run_name="__main__"):
  """``python -m <modname>`` for a reference script module (e.g.
  ``code.scripts.cluster.cluster_sobel``): translated like every other reference module and executed
  as ``__main__``."""
  spec = importlib.util.find_spec(modname)
  if spec is None or not isinstance(spec.loader, _Py2Loader):
    raise ImportError("%s is not a module of the reference tree (enable(root) first)" % modname)
  code = translate(spec.loader.get_source(), spec.origin)
  g = {"__name__": run_name, "__file__": spec.origin, "__package__": modname.rpartition(".")[0],
       "__builtins__": builtins, "__iic_py2div__": py2div, "__iic_py2idiv__": py2idiv}
  exec(code, g)
  return g


# ------------------------------------------------------------------------------------------
# third-party modules the reference imports and this image lacks
# ------------------------------------------------------------------------------------------
def _linear_assignment(cost):
  """sklearn.utils.linear_assignment_.linear_assignment (scikit-learn 0.19.1,
  package_versions.txt): minimum-cost assignment, returned as an [n, 2] index array.  Solved
  with scipy's Hungarian implementation (any optimal assignment has the same total cost)."""
  import numpy as np
  from scipy.optimize import linear_sum_assignment
  r, c = linear_sum_assignment(np.asarray(cost))
  return np.stack([r, c], axis=1)


class _ImportOnlyModule(types.ModuleType):
  """Stand-in for a data-layer dependency that is absent here: importing it (and its submodules)
  works, *using* anything from it raises with an explanation."""

  def __init__(self, name):
    super(_ImportOnlyModule, self).__init__(name)
    self.__path__ = []
    self.__iic_stub__ = True

  def __getattr__(self, attr):
    if attr.startswith("__"):
      raise AttributeError(attr)
    full = self.__name__ + "." + attr
    if full in sys.modules:
      return sys.modules[full]

    class _MissingMeta(type):
      # `torchvision.datasets.MNIST` at module level of a script (cluster_greyscale_twohead.py:133) must
      # evaluate; only USING the result raises
      def __getattr__(cls, name):
        if name.startswith("__"):
          raise AttributeError(name)
        return _MissingMeta(name, (cls,), {})

    class _Missing(object, metaclass=_MissingMeta):
      def __init__(s, *a, **k):
        raise ImportError("%s is not installed in this image (iic_amd.py2compat provides an "
                          "import-only stand-in; the reference's data layer is out of scope)" % full)
    _Missing.__name__ = attr
    return _Missing


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
  """Last on sys.meta_path: only consulted when the real module does not exist."""
  ROOTS = ("cv2", "torchvision")

  def find_spec(self, fullname, path=None, target=None):
    if fullname == "sklearn.utils.linear_assignment_" or fullname.split(".")[0] in self.ROOTS:
      return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
    return None

  def create_module(self, spec):
    if spec.name == "sklearn.utils.linear_assignment_":
      m = types.ModuleType(spec.name)
      m.linear_assignment = _linear_assignment
      return m
    return _ImportOnlyModule(spec.name)

  def exec_module(self, module):
    pass


_STATE = {"finder": None, "stubs": None}


def torch04_shims():
  """The reference was written against PyTorch 0.4.1 (package_versions.txt), where byte tensors were the
  mask type.  Its evaluation code selects with uint8 masks (code/utils/segmentation/segmentation_eval.py:
  58-60,126-128: `flat_preds.masked_select(mask=mask_all)` with a torch.uint8 mask), which current
  PyTorch rejects ("expected BoolTensor for mask").  masked_select is re-bound to accept byte masks
  with their 0.4.1 meaning (non-zero = selected); bool masks pass through untouched.  Idempotent."""
  import torch
  if getattr(torch.Tensor.masked_select, "__iic_torch04__", False):
    return
  orig_method, orig_fn = torch.Tensor.masked_select, torch.masked_select

  def _as_bool(mask):
    return mask != 0 if torch.is_tensor(mask) and mask.dtype == torch.uint8 else mask

  def masked_select(self, mask):
    return orig_method(self, _as_bool(mask))

  def masked_select_fn(input, mask, **kw):
    return orig_fn(input, _as_bool(mask), **kw)
  masked_select.__iic_torch04__ = True
  torch.Tensor.masked_select = masked_select
  torch.masked_select = masked_select_fn


def py2_builtins():
  """Names the Python-2 reference uses that survive lib2to3 untouched when they are reached
  dynamically (``itertools.izip`` as an attribute, ``xrange`` in eval'd strings)."""
  if not hasattr(builtins, "xrange"):
    builtins.xrange = range
  if not hasattr(itertools, "izip"):
    itertools.izip = zip


def enable(root=None):
  """Install the finder for <root>/code (default: $IIC_REFERENCE, else the first sys.path entry
  that contains ``code/__init__.py``).  Idempotent; returns the root in use."""
  if root is None:
    root = os.environ.get("IIC_REFERENCE")
  if root is None:
    for p in sys.path:
      if p and os.path.exists(os.path.join(p, _PKG, "__init__.py")):
        root = p
        break
  if root is None or not os.path.exists(os.path.join(root, _PKG, "__init__.py")):
    raise ImportError("reference tree not found: pass root=, set IIC_REFERENCE or put the "
                      "directory that contains code/ on PYTHONPATH")
  root = os.path.abspath(root)
  py2_builtins()
  torch04_shims()
  f = _STATE["finder"]
  if f is not None and f.root != root:
    disable()
    f = None
  if f is None:
    # the stdlib module `code` (or a half-imported reference package) may already be loaded
    for k in [k for k in sys.modules if k == _PKG or k.startswith(_PKG + ".")]:
      del sys.modules[k]
    f = Py2ReferenceFinder(root)
    sys.meta_path.insert(0, f)
    _STATE["finder"] = f
  if _STATE["stubs"] is None:
    s = _StubFinder()
    sys.meta_path.append(s)
    _STATE["stubs"] = s
  return root


def disable():
  for key in ("finder", "stubs"):
    f = _STATE[key]
    if f is not None and f in sys.meta_path:
      sys.meta_path.remove(f)
    _STATE[key] = None
  for k in [k for k in sys.modules if k == _PKG or k.startswith(_PKG + ".")]:
    del sys.modules[k]
  for k in [k for k, m in list(sys.modules.items()) if getattr(m, "__iic_stub__", False)]:
    del sys.modules[k]
