"""Helper process of tests/test_py2compat_cpu.py: runs one of the reference's UNCHANGED training scripts
(Python-2 source under /root/reference) for two batches per head plus its evaluation passes on the CPU
of this container:

  code/scripts/cluster/cluster_sobel.py, cluster_sobel_twohead.py, cluster_greyscale.py,
  cluster_greyscale_twohead.py, code/scripts/segmentation/segmentation.py, segmentation_twohead.py

    python tests/ref_script_driver.py <out_root> <script name>

What it proves: the Python-2 -> 3 import hook (iic_amd.py2compat) and the strict installer
(iic_amd.install) make the real script import and run end to end, and the script reaches every patch
point with the call shapes the HIP implementations accept (keyword loss calls, `head=` forwards, the
missing `head=` of cluster_greyscale_twohead.py:342-343).  There is no GPU here and the product has no
CPU path, so after install() has bound -- and this driver has asserted -- the HIP implementations, the
bound names are swapped for CPU stand-ins that COUNT calls and delegate to the reference's own PyTorch
modules / the oracle (test infrastructure); `.cuda()` becomes the identity.  The data layer (torchvision
datasets, out of scope) is replaced by synthetic modules with the same entry points.
"""
import inspect
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("IIC_REFERENCE", "/root/reference")
out_root = sys.argv[1]
SCRIPT = sys.argv[2] if len(sys.argv) > 2 else "cluster_sobel"

import torch  # noqa: E402

from iic_amd import install, py2compat  # noqa: E402

py2compat.enable(REF)
# the reference's own classes / functions, captured BEFORE install() rebinds the names
import code.archs.cluster.net5g as _r5g  # noqa: E402
import code.archs.cluster.net5g_two_head as _r5g2  # noqa: E402
import code.archs.cluster.net6c as _r6c  # noqa: E402
import code.archs.cluster.net6c_two_head as _r6c2  # noqa: E402
import code.archs.segmentation.net10a as _r10a  # noqa: E402
import code.archs.segmentation.net10a_twohead as _r10a2  # noqa: E402
import code.utils.segmentation.IID_losses as ref_seg_losses  # noqa: E402

REF_ARCH = {"ClusterNet5g": _r5g.ClusterNet5g, "ClusterNet5gTwoHead": _r5g2.ClusterNet5gTwoHead,
            "ClusterNet6c": _r6c.ClusterNet6c, "ClusterNet6cTwoHead": _r6c2.ClusterNet6cTwoHead,
            "SegmentationNet10a": _r10a.SegmentationNet10a,
            "SegmentationNet10aTwoHead": _r10a2.SegmentationNet10aTwoHead}
REF_SEG_LOSS = {"IID_segmentation_loss": ref_seg_losses.IID_segmentation_loss,
                "IID_segmentation_loss_uncollapsed": ref_seg_losses.IID_segmentation_loss_uncollapsed}

done = install.install(strict=True, reference_root=REF)
import code.archs as archs  # noqa: E402
import code.utils.cluster.cluster_eval as ce  # noqa: E402
import code.utils.cluster.general as general  # noqa: E402
import code.utils.cluster.IID_losses as ref_losses  # noqa: E402
import code.utils.segmentation.segmentation_eval as se  # noqa: E402

import iic_amd.archs  # noqa: E402
import iic_amd.eval_metrics  # noqa: E402
import iic_amd.losses  # noqa: E402
import iic_amd.optim  # noqa: E402
import iic_amd.seg_losses  # noqa: E402

GT_K = 10
SPECS = {
  "cluster_sobel": dict(
    module="code.scripts.cluster.cluster_sobel", arch="ClusterNet5g", data="cluster", in_ch=1, input_sz=32, model_ind=7,
    argv=["--arch", "ClusterNet5g", "--dataset", "synthetic", "--output_k", "20", "--num_dataloaders", "3",
          "--num_sub_heads", "2", "--input_sz", "32", "--batchnorm_track"]),
  "cluster_sobel_twohead": dict(
    module="code.scripts.cluster.cluster_sobel_twohead", arch="ClusterNet5gTwoHead", data="cluster_twohead", in_ch=1,
    input_sz=32, model_ind=8,
    argv=["--arch", "ClusterNet5gTwoHead", "--dataset", "synthetic", "--output_k_A", "20", "--output_k_B", str(GT_K),
          "--num_dataloaders", "3", "--num_sub_heads", "2", "--input_sz", "32", "--batchnorm_track", "--head_A_first",
          "--double_eval", "--select_sub_head_on_loss"]),
  "cluster_greyscale": dict(
    module="code.scripts.cluster.cluster_greyscale", arch="ClusterNet6c", data="cluster", in_ch=1, input_sz=24,
    model_ind=9,
    argv=["--arch", "ClusterNet6c", "--dataset", "MNIST", "--output_k", "20", "--num_dataloaders", "3",
          "--num_sub_heads", "2", "--input_sz", "24", "--batchnorm_track", "--mode", "IID+"]),
  "cluster_greyscale_twohead": dict(
    module="code.scripts.cluster.cluster_greyscale_twohead", arch="ClusterNet6cTwoHead", data="cluster_twohead",
    in_ch=1, input_sz=24, model_ind=10,
    argv=["--arch", "ClusterNet6cTwoHead", "--dataset", "MNIST", "--output_k_A", "20", "--output_k_B", str(GT_K),
          "--num_dataloaders", "3", "--num_sub_heads", "2", "--input_sz", "24", "--batchnorm_track"]),
  "segmentation": dict(
    module="code.scripts.segmentation.segmentation", arch="SegmentationNet10a", data="segmentation", in_ch=4,
    input_sz=24, model_ind=11, gt_k=3,
    argv=["--arch", "SegmentationNet10a", "--dataset", "Potsdam", "--output_k", "6", "--num_dataloaders", "3",
          "--num_sub_heads", "1", "--input_sz", "24", "--batchnorm_track", "--mode", "IID+", "--include_rgb",
          "--half_T_side_dense", "1", "--use_uncollapsed_loss"]),
  "segmentation_twohead": dict(
    module="code.scripts.segmentation.segmentation_twohead", arch="SegmentationNet10aTwoHead", data="segmentation",
    in_ch=4, input_sz=24, model_ind=12, gt_k=3,
    argv=["--arch", "SegmentationNet10aTwoHead", "--dataset", "Potsdam", "--output_k_A", "6", "--output_k_B", "3",
          "--num_dataloaders", "3", "--num_sub_heads", "1", "--input_sz", "24", "--batchnorm_track", "--include_rgb",
          "--half_T_side_dense", "1"]),
}
spec = SPECS[SCRIPT]
gt_k = spec.get("gt_k", GT_K)
arch = spec["arch"]

bound = {
  "arch": archs.__dict__[arch] is getattr(iic_amd.archs, arch),
  "loss": ref_losses.IID_loss is iic_amd.losses.IID_loss and ce.IID_loss is iic_amd.losses.IID_loss,
  "seg_loss": ref_seg_losses.IID_segmentation_loss is iic_amd.seg_losses.IID_segmentation_loss and
              ref_seg_losses.IID_segmentation_loss_uncollapsed is iic_amd.seg_losses.IID_segmentation_loss_uncollapsed,
  "opt": general.get_opt("Adam") is iic_amd.optim.Adam,
  "eval": ce._original_match is iic_amd.eval_metrics._original_match and
          ce._hungarian_match is iic_amd.eval_metrics._hungarian_match,
  "sobel_eval": se.sobel_process is sys.modules["iic_amd.transforms"].sobel_process,
  "n_patched": len(done), "n_patches": len(install.PATCHES),
  # cluster_greyscale_twohead.py:342-343 calls net(all_imgs) WITHOUT head=: the product's default head
  # must be the reference's ("B", net6c_two_head.py:75)
  "default_head_6c": inspect.signature(iic_amd.archs.ClusterNet6cTwoHead.forward).parameters["head"].default,
  "default_head_ref_6c": inspect.signature(_r6c2.ClusterNet6cTwoHead.forward).parameters["head"].default,
  # the HIP losses accept the keyword call of segmentation_twohead.py:318-325
  "seg_loss_params": list(inspect.signature(iic_amd.seg_losses.IID_segmentation_loss_uncollapsed).parameters),
}

# ---- CPU stand-ins (counting) -------------------------------------------------------------
from oracle import eval_oracle, iid_oracle, net_oracle  # noqa: E402

calls = {"net_init": 0, "net_fwd": 0, "train_fwd_heads": [], "loss": 0, "seg_loss": 0, "seg_loss_kwargs": None,
         "seg_loss_positional": None, "sobel": 0, "match": 0, "acc": 0, "opt_step": 0}


def counting_net(cls):
  class CountingNet(cls):
    def __init__(self, config):
      calls["net_init"] += 1
      super(CountingNet, self).__init__(config)

    def forward(self, *a, **k):
      calls["net_fwd"] += 1
      if self.training and torch.is_grad_enabled():
        calls["train_fwd_heads"].append(k.get("head", "<default>"))
      return super(CountingNet, self).forward(*a, **k)
  CountingNet.__name__ = cls.__name__
  return CountingNet


LOSS_VALUES = []       # every IID_loss call's value, in call order (IIC_DRIVER_FULL: stored in the fixture)


def counting_loss(x_out, x_tf_out, lamb=1.0, EPS=sys.float_info.epsilon):
  calls["loss"] += 1
  r = iid_oracle.IID_loss(x_out, x_tf_out, lamb=lamb, EPS=EPS)
  LOSS_VALUES.append(float(r[0].detach()))
  return r


def counting_seg_loss(name):
  fn = REF_SEG_LOSS[name]                 # the reference's own function (runs on CPU tensors)

  def wrapped(*a, **k):
    calls["seg_loss"] += 1
    calls["seg_loss_positional"] = len(a)
    calls["seg_loss_kwargs"] = sorted(k)
    return fn(*a, **k)
  return wrapped


def counting_sobel(imgs, include_rgb, using_IR=False):
  calls["sobel"] += 1
  return net_oracle.sobel_process(imgs, include_rgb, using_IR=using_IR)


def counting_original(flat_preds, flat_targets, preds_k, targets_k):
  calls["match"] += 1
  return eval_oracle.original_match(flat_preds, flat_targets, preds_k, targets_k)


def counting_hungarian(flat_preds, flat_targets, preds_k, targets_k):
  calls["match"] += 1
  return eval_oracle.hungarian_match(flat_preds, flat_targets, preds_k, targets_k)


def counting_acc(preds, targets, num_k, verbose=0):
  calls["acc"] += 1
  return eval_oracle.acc(preds, targets)


class CountingAdam(torch.optim.Adam):
  def step(self, *a, **k):
    calls["opt_step"] += 1
    return super(CountingAdam, self).step(*a, **k)


for m in (archs, sys.modules["code.archs.cluster"], sys.modules["code.archs.segmentation"]):
  if hasattr(m, arch):
    setattr(m, arch, counting_net(REF_ARCH[arch]))
for m in (ref_losses, ce):
  m.IID_loss = counting_loss
for n in REF_SEG_LOSS:
  setattr(ref_seg_losses, n, counting_seg_loss(n))
for m in (sys.modules["code.utils.cluster.transforms"], ce, se):
  m.sobel_process = counting_sobel
for m in (sys.modules["code.utils.cluster.eval_metrics"], ce):
  m._original_match = counting_original
  m._hungarian_match = counting_hungarian
  m._acc = counting_acc
general._opt_dict["Adam"] = CountingAdam

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

# ---- synthetic stand-ins for the data layer ---------------------------------------------------
# IIC_DRIVER_FULL=<epochs> (oracle/gen_golden_script.py): whole epochs on the data / arguments / seeds of the GPU driver
# (tests/ref_script_gpu_driver.py: 44 images per loader = 5 full batches + a ragged one, batch 24), no --test_code --
# the reference's own modules on the CPU produce the epoch losses the HIP path is held to (tests/golden/script_*.json)
FULL = int(os.environ.get("IIC_DRIVER_FULL", "0"))
NUM_IMGS = 44 if FULL else 24


def _cluster_loaders(config, seed0):
  g = torch.Generator().manual_seed(seed0)
  per = config.dataloader_batch_sz
  assert isinstance(per, int), "py2 integer division of the batch size was not preserved"
  base = torch.rand(NUM_IMGS, spec["in_ch"], config.input_sz, config.input_sz, generator=g)
  labels = torch.randint(0, config.gt_k, (NUM_IMGS,), generator=g)

  def loader(tf_seed):
    imgs = base if tf_seed is None else \
      (torch.flip(base, dims=[3]) * 0.9 + 0.02 * torch.randn(base.shape, generator=torch.Generator().manual_seed(tf_seed))).clamp(0, 1)
    return [(imgs[i:i + per],) for i in range(0, NUM_IMGS, per)]
  dataloaders = [loader(None)] + [loader(seed0 + 1 + d) for d in range(config.num_dataloaders)]
  bs = config.batch_sz
  mapping = [(base[i:i + bs], labels[i:i + bs]) for i in range(0, NUM_IMGS, bs)]
  return dataloaders, mapping


def cluster_create_dataloaders(config):
  # (the fields code/utils/cluster/data.py:114-144 sets for its datasets)
  config.train_partitions, config.mapping_assignment_partitions, config.mapping_test_partitions = [True], [True], [False]
  dl, mapping = _cluster_loaders(config, 0)
  return dl, mapping, mapping


def cluster_twohead_create_dataloaders(config):
  # (code/utils/cluster/data.py:25-29,64-68)
  config.train_partitions_head_A = config.train_partitions_head_B = [True, False]
  config.mapping_assignment_partitions = config.mapping_test_partitions = [True, False]
  dl_a, mapping = _cluster_loaders(config, 0)
  dl_b, _ = _cluster_loaders(config, 100)
  return dl_a, dl_b, mapping, mapping


def segmentation_create_dataloaders(config):
  # (code/utils/segmentation/data.py:31-33)
  config.train_partitions = config.mapping_assignment_partitions = config.mapping_test_partitions = ["all"]
  g = torch.Generator().manual_seed(3)
  per = config.dataloader_batch_sz
  assert isinstance(per, int)
  S = config.input_sz
  pre = config.in_channels - (0 if config.no_sobel else 1)
  base = torch.rand(NUM_IMGS, pre, S, S, generator=g)
  labels = torch.randint(0, config.gt_k, (NUM_IMGS, S, S), generator=g)
  eye = torch.tensor([[1., 0., 0.], [0., 1., 0.]])

  def loader(tf_seed):
    img2 = (base * 0.9 + 0.02 * torch.randn(base.shape, generator=torch.Generator().manual_seed(tf_seed))).clamp(0, 1)
    mask = (torch.rand(NUM_IMGS, S, S, generator=torch.Generator().manual_seed(50 + tf_seed)) > 0.1).float()
    return [(base[i:i + per], img2[i:i + per], eye.expand(min(per, NUM_IMGS - i), 2, 3).clone(), mask[i:i + per])
            for i in range(0, NUM_IMGS, per)]
  dataloaders = [loader(1 + d) for d in range(config.num_dataloaders)]
  bs = config.batch_sz
  emask = torch.ones(NUM_IMGS, S, S, dtype=torch.uint8)     # (the reference's loaders yield uint8 masks)
  mapping = [(base[i:i + bs], labels[i:i + bs], emask[i:i + bs]) for i in range(0, NUM_IMGS, bs)]
  return dataloaders, mapping, mapping


cdata = types.ModuleType("code.utils.cluster.data")
cdata.cluster_create_dataloaders = cluster_create_dataloaders
cdata.cluster_twohead_create_dataloaders = cluster_twohead_create_dataloaders
sys.modules["code.utils.cluster.data"] = cdata
sdata = types.ModuleType("code.utils.segmentation.data")
sdata.segmentation_create_dataloaders = segmentation_create_dataloaders
sys.modules["code.utils.segmentation.data"] = sdata

name = spec["module"].rsplit(".", 1)[1]
sys.argv = [name, "--model_ind", str(spec["model_ind"]), "--dataset_root", "/nonexistent", "--gt_k", str(gt_k),
            "--lr", "0.001", "--num_epochs", "3", "--batch_sz", "12", "--out_root", out_root, "--test_code"] + spec["argv"]
if FULL:
  import random
  import numpy as np
  random.seed(0)
  np.random.seed(0)
  torch.manual_seed(0)
  sys.argv = [name, "--model_ind", str(spec["model_ind"]), "--dataset_root", "/nonexistent", "--gt_k", str(gt_k),
              "--lr", "0.001", "--num_epochs", str(FULL), "--batch_sz", "24", "--out_root", out_root, "--save_freq", "1"] + \
      [a for a in spec["argv"]] + (["--lr_schedule", "2"] if SCRIPT.startswith("cluster_sobel") else [])
rc, err = None, None
try:
  py2compat.run_script(spec["module"])
except SystemExit as e:      # the scripts leave through exit(0) under --test_code
  rc = e.code
except Exception as e:       # noqa: BLE001  (reported to the test, with the traceback on stderr)
  import traceback
  traceback.print_exc()
  err = "%s: %s" % (type(e).__name__, e)
odir = os.path.join(out_root, str(spec["model_ind"]))
res = {"script": SCRIPT, "bound": bound, "calls": calls, "exit": rc, "error": err,
       "files": sorted(os.listdir(odir)) if os.path.isdir(odir) else []}
cp = os.path.join(odir, "config.pickle")
if FULL and os.path.exists(cp):
  import pickle
  with open(cp, "rb") as f:
    cfg = pickle.load(f)
  for k in ("epoch_loss", "epoch_loss_no_lamb", "epoch_acc", "epoch_loss_head_A", "epoch_loss_head_B", "last_epoch"):
    if hasattr(cfg, k):
      v = getattr(cfg, k)
      res[k] = [float(x) for x in v] if isinstance(v, (list, tuple)) else v
  res["argv"] = sys.argv[1:]
  res["loss_calls"] = LOSS_VALUES
print("IIC_DRIVER_RESULT " + json.dumps(res))
