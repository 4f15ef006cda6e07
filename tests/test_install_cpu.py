"""install() rebinds the reference's import sites (checked on a miniature fake tree; the
real scripts need torchvision / datasets that are absent here -- SURVEY.md §8b)."""
import os
import sys
import textwrap


def test_install_rebinds_reference_names(tmp_path, monkeypatch):
  root = tmp_path / "ref"
  for d in ("code", "code/utils", "code/utils/cluster", "code/utils/segmentation", "code/archs", "code/archs/cluster",
            "code/archs/segmentation"):
    (root / d).mkdir(parents=True)
    (root / d / "__init__.py").write_text("")
  (root / "code/utils/cluster/IID_losses.py").write_text("def IID_loss(*a, **k):\n  return 'ref'\n")
  (root / "code/utils/cluster/transforms.py").write_text("def sobel_process(*a, **k):\n  return 'ref'\n")
  (root / "code/utils/cluster/eval_metrics.py").write_text(
    "def _original_match(*a):\n  return 'ref'\ndef _hungarian_match(*a):\n  return 'ref'\ndef _acc(*a):\n  return 'ref'\n")
  (root / "code/utils/cluster/cluster_eval.py").write_text(
    "from .IID_losses import IID_loss\nfrom .eval_metrics import _hungarian_match, _original_match, _acc\n"
    "from .transforms import sobel_process\n")
  (root / "code/utils/cluster/general.py").write_text(
    "from torch.optim import Adam\n_opt_dict = {'Adam': Adam}\ndef get_opt(name):\n  return _opt_dict[name]\n")
  (root / "code/utils/segmentation/segmentation_eval.py").write_text(
    "from code.utils.cluster.transforms import sobel_process\n")
  (root / "code/archs/__init__.py").write_text("class ClusterNet5g: pass\nclass ClusterNet5gTwoHead: pass\nclass ClusterNet6c: pass\nclass ClusterNet6cTwoHead: pass\nclass SegmentationNet10a: pass\nclass SegmentationNet10aTwoHead: pass\n")
  (root / "code/archs/segmentation/__init__.py").write_text("class SegmentationNet10a: pass\nclass SegmentationNet10aTwoHead: pass\n")
  (root / "code/utils/segmentation/IID_losses.py").write_text("def IID_segmentation_loss(*a, **k):\n  return 0\ndef IID_segmentation_loss_uncollapsed(*a, **k):\n  return 0\n")
  (root / "code/archs/cluster/__init__.py").write_text("class ClusterNet5g: pass\nclass ClusterNet5gTwoHead: pass\nclass ClusterNet6c: pass\nclass ClusterNet6cTwoHead: pass\n")
  monkeypatch.syspath_prepend(str(root))
  for k in [k for k in sys.modules if k == "code" or k.startswith("code.")]:
    monkeypatch.delitem(sys.modules, k)
  from iic_amd import archs, install, losses, transforms
  install.py2_shims()
  done = install.install(strict=True)
  assert len(done) == len(install.PATCHES) + 1      # + the get_opt table entry
  from code.utils.cluster.general import get_opt
  from iic_amd import optim
  assert get_opt("Adam") is optim.Adam
  script = textwrap.dedent("""
    from code.utils.cluster.IID_losses import IID_loss
    from code.utils.cluster.transforms import sobel_process
    import code.archs as archs
    net_cls = archs.__dict__["ClusterNet5g"]
  """)
  ns = {}
  exec(script, ns)
  assert ns["IID_loss"] is losses.IID_loss
  assert ns["sobel_process"] is transforms.sobel_process
  assert ns["net_cls"] is archs.ClusterNet5g
  from iic_amd import eval_metrics
  import code.utils.cluster.cluster_eval as ce      # binds the names at import: patched in place
  assert ce._original_match is eval_metrics._original_match and ce._acc is eval_metrics._acc
  assert xrange is range  # noqa: F821
  from iic_amd import py2compat
  py2compat.disable()
