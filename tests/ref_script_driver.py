"""Helper process of tests/test_reference_scripts_cpu.py: runs the reference's UNCHANGED
code/scripts/cluster/cluster_sobel.py (Python-2 source under /root/reference) for two batches
plus both evaluation passes on the CPU of this container.

What it proves: the Python-2 -> 3 import hook (iic_amd.py2compat) and the strict installer
(iic_amd.install) make the real script import and run end to end, and the script reaches every
patch point.  There is no GPU here and the product has no CPU path, so after install() has
bound -- and this driver has asserted -- the HIP implementations, the bound names are swapped
for CPU stand-ins that COUNT calls and delegate to the reference's own PyTorch modules / the
oracle (test infrastructure); `.cuda()` becomes the identity.  The data layer (torchvision
datasets, out of scope) is replaced by a synthetic module with the same entry point.
"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("IIC_REFERENCE", "/root/reference")
out_root = sys.argv[1]

import torch  # noqa: E402

from iic_amd import install, py2compat  # noqa: E402

py2compat.enable(REF)
import code.archs.cluster.net5g as ref_net5g  # noqa: E402  (the reference's own class, pre-patch)
RefClusterNet5g = ref_net5g.ClusterNet5g
done = install.install(strict=True, reference_root=REF)
import code.archs as archs  # noqa: E402
import code.utils.cluster.cluster_eval as ce  # noqa: E402
import code.utils.cluster.general as general  # noqa: E402
import code.utils.cluster.IID_losses as ref_losses  # noqa: E402

import iic_amd.archs  # noqa: E402
import iic_amd.eval_metrics  # noqa: E402
import iic_amd.losses  # noqa: E402
import iic_amd.optim  # noqa: E402

bound = {
  "arch": archs.__dict__["ClusterNet5g"] is iic_amd.archs.ClusterNet5g,
  "loss": ref_losses.IID_loss is iic_amd.losses.IID_loss and ce.IID_loss is iic_amd.losses.IID_loss,
  "opt": general.get_opt("Adam") is iic_amd.optim.Adam,
  "eval": ce._original_match is iic_amd.eval_metrics._original_match,
  "n_patched": len(done), "n_patches": len(install.PATCHES),
}

# ---- CPU stand-ins (counting) -------------------------------------------------------------
from oracle import eval_oracle, iid_oracle, net_oracle  # noqa: E402

calls = {"net_init": 0, "net_fwd": 0, "loss": 0, "sobel": 0, "match": 0, "acc": 0, "opt_step": 0}


class CountingNet(RefClusterNet5g):
  def __init__(self, config):
    calls["net_init"] += 1
    super(CountingNet, self).__init__(config)

  def forward(self, *a, **k):
    calls["net_fwd"] += 1
    return super(CountingNet, self).forward(*a, **k)


def counting_loss(x_out, x_tf_out, lamb=1.0, EPS=sys.float_info.epsilon):
  calls["loss"] += 1
  return iid_oracle.IID_loss(x_out, x_tf_out, lamb=lamb, EPS=EPS)


def counting_sobel(imgs, include_rgb, using_IR=False):
  calls["sobel"] += 1
  return net_oracle.sobel_process(imgs, include_rgb, using_IR=using_IR)


def counting_match(flat_preds, flat_targets, preds_k, targets_k):
  calls["match"] += 1
  return eval_oracle.original_match(flat_preds, flat_targets, preds_k, targets_k)


def counting_acc(preds, targets, num_k, verbose=0):
  calls["acc"] += 1
  return eval_oracle.acc(preds, targets)


class CountingAdam(torch.optim.Adam):
  def step(self, *a, **k):
    calls["opt_step"] += 1
    return super(CountingAdam, self).step(*a, **k)


for m in (archs, sys.modules["code.archs.cluster"]):
  m.ClusterNet5g = CountingNet
for m in (ref_losses, ce):
  m.IID_loss = counting_loss
for m in (sys.modules["code.utils.cluster.transforms"], ce):
  m.sobel_process = counting_sobel
for m in (sys.modules["code.utils.cluster.eval_metrics"], ce):
  m._original_match = counting_match
  m._acc = counting_acc
general._opt_dict["Adam"] = CountingAdam

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

# ---- synthetic stand-in for the data layer (code/utils/cluster/data.py:259-339 entry point) ----
INPUT_SZ, NUM_IMGS, GT_K = 32, 24, 10


def cluster_create_dataloaders(config):
  g = torch.Generator().manual_seed(0)
  per = config.dataloader_batch_sz
  assert isinstance(per, int), "py2 integer division of the batch size was not preserved"
  base = torch.rand(NUM_IMGS, 1, config.input_sz, config.input_sz, generator=g)
  labels = torch.randint(0, config.gt_k, (NUM_IMGS,), generator=g)

  def loader(tf_seed):
    imgs = base if tf_seed is None else \
      (torch.flip(base, dims=[3]) * 0.9 + 0.02 * torch.randn(base.shape, generator=torch.Generator().manual_seed(tf_seed))).clamp(0, 1)
    return [(imgs[i:i + per],) for i in range(0, NUM_IMGS, per)]
  dataloaders = [loader(None)] + [loader(1 + d) for d in range(config.num_dataloaders)]
  bs = config.batch_sz
  mapping = [(base[i:i + bs], labels[i:i + bs]) for i in range(0, NUM_IMGS, bs)]
  return dataloaders, mapping, mapping


data = types.ModuleType("code.utils.cluster.data")
data.cluster_create_dataloaders = cluster_create_dataloaders
sys.modules["code.utils.cluster.data"] = data

sys.argv = ["cluster_sobel", "--model_ind", "7", "--arch", "ClusterNet5g", "--dataset", "synthetic",
            "--dataset_root", "/nonexistent", "--gt_k", str(GT_K), "--output_k", "20", "--lr", "0.001",
            "--num_epochs", "3", "--batch_sz", "12", "--num_dataloaders", "3", "--num_sub_heads", "2",
            "--input_sz", str(INPUT_SZ), "--out_root", out_root, "--test_code", "--batchnorm_track"]
rc = None
try:
  py2compat.run_script("code.scripts.cluster.cluster_sobel")
except SystemExit as e:      # the script leaves through exit(0) under --test_code (cluster_sobel.py:341)
  rc = e.code
print("IIC_DRIVER_RESULT " + json.dumps({"bound": bound, "calls": calls, "exit": rc,
                                           "files": sorted(os.listdir(os.path.join(out_root, "7")))}))
