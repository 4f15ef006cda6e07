import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
  config.addinivalue_line("markers", "hooks: toggles an iic_debug_* switch -- only the instrumented library has them "
                                     "(IIC_HIP_LIB=dbg; tests/test_gpu_kernels.py runs these in a sub-process)")


def pytest_collection_modifyitems(config, items):
  """The product library has no measurement switches: tests that need one are deselected unless this process loaded the
  instrumented flavour (iic_amd/_lib.py).  test_switch_dependent_tests_pass_in_the_instrumented_library runs them."""
  from iic_amd import _lib
  if _lib.HAS_HOOKS:
    return
  keep, drop = [], []
  for it in items:
    (drop if it.get_closest_marker("hooks") else keep).append(it)
  if drop:
    config.hook.pytest_deselected(items=drop)
    items[:] = keep


def hook(name, *args):
  """Call switch `name` of the instrumented library; in the product library only a call that restores a DEFAULT may
  come here (a no-op there): every test that sets anything else carries the `hooks` marker."""
  import ctypes
  from iic_amd import _lib
  if _lib.HAS_HOOKS:
    return getattr(ctypes.CDLL(_lib.LIB_PATH), name)(*args)
  return None


@pytest.fixture(scope="session")
def golden_dir():
  return GOLDEN
