"""The C-ABI library loads and exports every symbol include/iic_hip.h declares (no compute
calls -- runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  src = open(os.path.join(ROOT, "include", "iic_hip.h")).read()
  src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
  return sorted(set(re.findall(r"\b(iic_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
  from iic_amd import _lib
  assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
  names = _declared()
  assert len(names) >= 30
  h = ctypes.CDLL(_lib.LIB_PATH)
  for n in names:
    assert hasattr(h, n), "libiic_hip.so does not export %s" % n
  # the Python binding covers exactly the declared API
  assert sorted(_lib.EXPORTED_SYMBOLS) == names
  L = _lib.lib()
  assert L.iic_version() >= 1
  assert L.iic_iid_nsplit(660) >= 1
  assert L.iic_iid_workspace_bytes(5, 70) >= 5 * 70 * 70 * 8      # [H][k][k] float64 + the multi-block stage's scratch


def test_geom_struct_matches_header_size():
  from iic_amd import _lib
  assert ctypes.sizeof(_lib.ConvGeom) == 4 * (3 + 3 + 4 + 3 + 4 + 1 + 32 + 32 + 4)


def test_product_path_has_no_cpu_fallback():
  import pytest
  import torch
  from iic_amd.losses import IID_loss
  z = torch.full((4, 3), 1.0 / 3)
  with pytest.raises(AssertionError):
    IID_loss(z, z)


def test_product_code_never_imports_oracle():
  pkg = os.path.join(ROOT, "iic_amd")
  for dp, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(".py"):
        src = open(os.path.join(dp, f)).read()
        assert "import oracle" not in src and "from oracle" not in src, f
