"""The Python-2 -> 3 layer that lets the UNCHANGED reference scripts run (SURVEY.md §8b last
row): unit tests of the in-memory translation, the strict installer against
the REAL reference tree, and the real cluster_sobel.py run for two batches on CPU stand-ins."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("IIC_REFERENCE", "/root/reference")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "code")),
                               reason="reference tree not present (GPU box)")


def _run(code, *args, timeout=600):
  env = dict(os.environ, PYTHONPATH=ROOT, IIC_REFERENCE=REF, MPLBACKEND="Agg")
  return subprocess.run([sys.executable, "-W", "ignore", "-c", code] + list(args), env=env,
                        capture_output=True, text=True, timeout=timeout)


def _exec(src, path="/tmp/x.py", extra=None):
  from iic_amd import py2compat
  g = {"__name__": "m", "__iic_py2div__": py2compat.py2div, "__iic_py2idiv__": py2compat.py2idiv}
  g.update(extra or {})
  exec(py2compat.translate(textwrap.dedent(src), path), g)
  return g


def test_translate_py2_idioms():
  g = _exec("""
    import itertools
    d = {"a": 1, "b": 2}
    items = sorted(d.iteritems())
    print "statement form", len(items)
    n = 7 / 2
    f = 7 / 2.0
    total = 0
    for i, j in itertools.izip(xrange(3), xrange(3)):
      total += i * j
    has = d.has_key("a")
    k = d.keys()
    k.sort()
  """)
  assert g["items"] == [("a", 1), ("b", 2)] and g["n"] == 3 and g["f"] == 3.5
  assert g["total"] == 5 and g["has"] is True and g["k"] == ["a", "b"]


def test_translate_division_semantics():
  import torch
  g = _exec("""
    import torch
    batch_sz, num_dataloaders = 700, 3
    per = batch_sz / num_dataloaders          # cluster_sobel_twohead.py:122
    t = torch.ones(4)
    alias = t
    t /= 4                                    # cluster_sobel.py:252: in place on a tensor
    acc = 3 / float(4)
    i = 9
    i /= 2
  """)
  assert g["per"] == 233 and isinstance(g["per"], int)
  assert g["alias"] is g["t"] and torch.equal(g["t"], torch.full((4,), 0.25))
  assert g["acc"] == 0.75 and g["i"] == 4
  g = _exec("""
    from __future__ import division
    x = 7 / 2
  """)
  assert g["x"] == 3.5


def test_division_of_integer_arrays_and_tensors():
  """Python 2 / numpy classic division and torch 0.4.1's integer `/` (the reference has no
  `from __future__ import division`): integer ndarrays floor, integer tensors truncate, anything with a
  float operand divides truly; the in-place form keeps the object."""
  import numpy as np
  import torch
  from iic_amd.py2compat import py2div, py2idiv
  a = np.array([7, -7, 9])
  assert py2div(a, 2).tolist() == [3, -4, 4] and py2div(a, 2).dtype.kind == "i"
  assert py2div(a, np.array([2, 2, 4])).tolist() == [3, -4, 2]
  assert py2div(a, 2.0).dtype.kind == "f" and py2div(np.int64(7), 2) == 3 and py2div(np.float32(7), 2) == 3.5
  t = torch.tensor([7, -7, 9])
  assert py2div(t, 2).tolist() == [3, -3, 4] and not py2div(t, 2).is_floating_point()
  assert py2div(t.float(), 2).tolist() == [3.5, -3.5, 4.5] and py2div(t, 2.0).is_floating_point()
  b = np.array([8, 9])
  assert py2idiv(b, 4) is b and b.tolist() == [2, 2]
  u = torch.tensor([8, 9])
  assert py2idiv(u, 4) is u and u.tolist() == [2, 2]
  f = torch.tensor([1.0, 2.0])
  assert py2idiv(f, 4) is f and f.tolist() == [0.25, 0.5]


def test_implicit_relative_imports_and_stubs(tmp_path):
  root = tmp_path / "tree"
  (root / "code" / "pkg").mkdir(parents=True)
  (root / "code" / "__init__.py").write_text("")
  (root / "code" / "pkg" / "__init__.py").write_text("from sibling import *\nfrom other import helper\n")
  (root / "code" / "pkg" / "sibling.py").write_text(
    "import cv2\nimport torchvision.transforms.functional as tf\n"
    "from sklearn.utils.linear_assignment_ import linear_assignment\n"
    "__all__ = ['VALUE', 'solve']\nVALUE = 10 / 4\n"
    "def solve(c):\n  return linear_assignment(c)\n")
  (root / "code" / "pkg" / "other.py").write_text("def helper():\n  print 'py2 print'\n  return 'ok'\n")
  r = _run(textwrap.dedent("""
    import sys
    from iic_amd import py2compat
    py2compat.enable(sys.argv[1])
    import code.pkg as p
    assert p.VALUE == 2 and p.helper() == 'ok', (p.VALUE,)
    m = p.solve([[4, 1], [2, 5]])
    assert sorted(map(tuple, m.tolist())) == [(0, 1), (1, 0)], m
    import cv2
    try:
      cv2.imread('x')
    except ImportError as e:
      assert 'not installed' in str(e)
    else:
      raise SystemExit('stub cv2 must raise on use')
    import os
    assert not any('__pycache__' in d for d, _, _ in os.walk(sys.argv[1])), 'hook must not write caches'
    py2compat.disable()
    import code
    assert hasattr(code, 'InteractiveConsole'), 'stdlib code module must be back after disable()'
    print('OK')
  """), str(root))
  assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@needs_ref
def test_install_strict_patches_every_name_of_the_real_reference():
  r = _run(textwrap.dedent("""
    from iic_amd import install
    done = install.install(strict=True)
    assert len(done) == len(install.PATCHES) + 1, (len(done), len(install.PATCHES))
    import code.archs as archs
    import iic_amd.archs, iic_amd.optim, iic_amd.losses, iic_amd.seg_losses
    for n in ("ClusterNet5g", "ClusterNet5gTwoHead", "ClusterNet6c", "ClusterNet6cTwoHead",
              "SegmentationNet10a", "SegmentationNet10aTwoHead"):
      assert archs.__dict__[n] is getattr(iic_amd.archs, n), n
    from code.utils.cluster.general import get_opt
    assert get_opt("Adam") is iic_amd.optim.Adam
    # scripts bind by name at import: what they would get
    from code.utils.cluster.IID_losses import IID_loss
    from code.utils.segmentation.IID_losses import IID_segmentation_loss_uncollapsed as segl
    assert IID_loss is iic_amd.losses.IID_loss and segl is iic_amd.seg_losses.IID_segmentation_loss_uncollapsed
    # every script module of the two in-scope families translates (compiles) under the hook
    import os, sys
    from iic_amd import py2compat
    root = os.environ["IIC_REFERENCE"]
    n = 0
    for fam in ("cluster", "segmentation"):
      d = os.path.join(root, "code", "scripts", fam)
      for f in sorted(os.listdir(d)):
        if f.endswith(".py") and f != "__init__.py":
          py2compat.translate(open(os.path.join(d, f)).read(), os.path.join(d, f)); n += 1
    print("OK", len(done), n)
  """))
  assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@needs_ref
def test_install_strict_raises_when_a_name_cannot_be_rebound(tmp_path):
  root = tmp_path / "partial"
  (root / "code").mkdir(parents=True)
  (root / "code" / "__init__.py").write_text("")
  r = _run(textwrap.dedent("""
    import sys
    from iic_amd import install
    try:
      install.install(strict=True, reference_root=sys.argv[1])
    except ImportError as e:
      print("RAISED", e)
  """), str(root))
  assert "RAISED" in r.stdout and "cannot rebind" in r.stdout, r.stdout + r.stderr


def _drive(tmp_path, script):
  env = dict(os.environ, PYTHONPATH=ROOT, IIC_REFERENCE=REF, MPLBACKEND="Agg", OMP_NUM_THREADS="8")
  r = subprocess.run([sys.executable, "-W", "ignore", os.path.join(ROOT, "tests", "ref_script_driver.py"),
                      str(tmp_path), script], env=env, capture_output=True, text=True, timeout=1200)
  line = [l for l in r.stdout.splitlines() if l.startswith("IIC_DRIVER_RESULT ")]
  assert line, r.stdout[-3000:] + r.stderr[-3000:]
  res = json.loads(line[0][len("IIC_DRIVER_RESULT "):])
  assert res["error"] is None, (res["error"], r.stderr[-3000:])
  b = res["bound"]
  assert b["arch"] and b["loss"] and b["seg_loss"] and b["opt"] and b["eval"] and b["sobel_eval"]
  assert b["n_patched"] == b["n_patches"] + 1
  assert res["exit"] == 0                               # the scripts leave through exit(0) under --test_code
  assert "plots.png" in res["files"] and "config.pickle" in res["files"]
  return res, r.stdout


@needs_ref
def test_unchanged_cluster_sobel_script_runs_two_batches(tmp_path):
  res, out = _drive(tmp_path, "cluster_sobel")
  c = res["calls"]
  # 2 train batches x 2 views + evaluation forwards (2 eval passes x 2 loaders x 2 batches)
  assert c["net_init"] == 1 and c["opt_step"] == 2 and c["loss"] == 2 * 2
  assert c["net_fwd"] == 2 * 2 + 2 * 2 * 2 and c["sobel"] == c["net_fwd"]
  assert c["match"] == 2 * 2 and c["acc"] >= 2 * 2
  assert "Model ind 7 epoch 1 batch: 1" in out          # the script's own progress line


@needs_ref
def test_unchanged_cluster_sobel_twohead_script(tmp_path):
  """cluster_sobel_twohead.py:265-359: head A then head B (--head_A_first), `net(x, head=head)`,
  sub-head selection on the loss and the double evaluation."""
  res, out = _drive(tmp_path, "cluster_sobel_twohead")
  c = res["calls"]
  assert c["net_init"] == 1 and c["opt_step"] == 2 * 2
  assert c["train_fwd_heads"] == ["A"] * 4 + ["B"] * 4               # 2 batches x 2 views per head
  assert c["loss"] >= 2 * 2 * 2 and c["sobel"] == c["net_fwd"]        # + get_subhead_using_loss's evaluations
  assert c["match"] >= 4 and c["acc"] >= 4
  assert "head A head_i_epoch 0 batch 1" in out and "head B head_i_epoch 0 batch 1" in out   # the script's own lines


@needs_ref
def test_unchanged_cluster_greyscale_script(tmp_path):
  res, _ = _drive(tmp_path, "cluster_greyscale")
  c = res["calls"]
  assert c["net_init"] == 1 and c["opt_step"] == 2 and c["loss"] == 2 * 2 and c["sobel"] == 0
  assert c["train_fwd_heads"] == ["<default>"] * 4


@needs_ref
def test_unchanged_cluster_greyscale_twohead_script_keeps_the_missing_head_argument(tmp_path):
  """cluster_greyscale_twohead.py:342-343 calls `net(all_imgs)` WITHOUT head= in BOTH head loops, so the
  reference always trains through ClusterNet6cTwoHead's default head "B" (net6c_two_head.py:75): the
  quirk must survive -- the product's default is checked to be the reference's."""
  res, _ = _drive(tmp_path, "cluster_greyscale_twohead")
  c, b = res["calls"], res["bound"]
  assert b["default_head_6c"] == "B" == b["default_head_ref_6c"]
  assert c["train_fwd_heads"] == ["<default>"] * 8                   # 2 heads x 2 batches x 2 views, none passes head=
  assert c["net_init"] == 1 and c["opt_step"] == 4 and c["loss"] == 2 * 2 * 2 and c["sobel"] == 0


@needs_ref
def test_unchanged_segmentation_script(tmp_path):
  res, _ = _drive(tmp_path, "segmentation")
  c = res["calls"]
  assert c["net_init"] == 1 and c["opt_step"] == 2 and c["seg_loss"] == 2 and c["loss"] == 0
  assert c["seg_loss_positional"] == 2
  assert c["seg_loss_kwargs"] == ["all_affine2_to_1", "all_mask_img1", "half_T_side_dense", "half_T_side_sparse_max",
                                  "half_T_side_sparse_min", "lamb"]
  assert c["sobel"] == c["net_fwd"] and c["match"] >= 2


@needs_ref
def test_unchanged_segmentation_twohead_script_keyword_loss_call(tmp_path):
  """segmentation_twohead.py:262-361: both heads, and the loss called as
  loss_fn(x1_outs[i], x2_outs[i], all_affine2_to_1=..., all_mask_img1=..., lamb=..., half_T_side_dense=...,
  half_T_side_sparse_min=..., half_T_side_sparse_max=...) (:318-325) -- the HIP losses take exactly these
  keywords (checked against their signature)."""
  res, _ = _drive(tmp_path, "segmentation_twohead")
  c, b = res["calls"], res["bound"]
  assert c["train_fwd_heads"] == ["A"] * 4 + ["B"] * 4
  assert c["net_init"] == 1 and c["opt_step"] == 4 and c["seg_loss"] == 4
  assert c["seg_loss_positional"] == 2
  assert set(c["seg_loss_kwargs"]) <= set(b["seg_loss_params"][2:])
  assert b["seg_loss_params"][:2] == ["x1_outs", "x2_outs"]
