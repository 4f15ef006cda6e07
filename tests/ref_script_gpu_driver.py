"""Helper process of tests/test_gpu_script.py::test_real_reference_scripts_drive_the_hip_kernels: runs one of the
reference's UNCHANGED training scripts (Python-2 source, read where it lies under $IIC_REFERENCE) on the MI355X
through exactly what `python -m iic_amd.run` does -- import hook, strict install, two-stream forwards, graph replay --
for whole epochs: every batch incl. the ragged last one, the evaluation passes (`double_eval` where the script has
it), the checkpoint round trip (net.module.cpu() / state_dict / torch.save / .cuda()) and, in a second invocation,
`--restart`.

    python tests/ref_script_gpu_driver.py <out_root> <script name> <num_epochs> [--restart]

Nothing here stands in for the product: the architectures, losses, Sobel, evaluation matching and Adam the script
reaches are the HIP implementations install() bound (asserted below).  Only the data layer (torchvision datasets and
PIL augmentation, out of scope: SURVEY.md section 2) is replaced by synthetic modules with the same entry points.
RNGs are seeded before the script module runs so that an eager run and a graph-replay run start from equal weights.
"""
import json
import os
import pickle
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ["IIC_REFERENCE"]
out_root, SCRIPT, NUM_EPOCHS = sys.argv[1], sys.argv[2], int(sys.argv[3])
RESTART = "--restart" in sys.argv[4:]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from iic_amd import install, ops, py2compat  # noqa: E402

py2compat.enable(REF)
done = install.install(strict=True, reference_root=REF)
ops.AUTO_BRANCH[0] = os.environ.get("IIC_AUTO_BRANCH", "1") != "0"          # as iic_amd.run.main
ops.GRAPH_FORWARD[0] = os.environ.get("IIC_GRAPH_FORWARD", "1") != "0"

import code.archs as archs  # noqa: E402
import code.utils.cluster.cluster_eval as ce  # noqa: E402
import code.utils.cluster.general as general  # noqa: E402
import code.utils.cluster.IID_losses as ref_losses  # noqa: E402
import code.utils.segmentation.IID_losses as ref_seg_losses  # noqa: E402

import iic_amd.archs  # noqa: E402
import iic_amd.eval_metrics  # noqa: E402
import iic_amd.losses  # noqa: E402
import iic_amd.optim  # noqa: E402
import iic_amd.seg_losses  # noqa: E402

GT_K = 10
BATCH = 24            # 3 loaders x 8 images
NUM_IMGS = 44         # per loader: 5 full batches of 8 + a ragged one of 4
SPECS = {
  "cluster_sobel": dict(
    module="code.scripts.cluster.cluster_sobel", arch="ClusterNet5g", in_ch=1, model_ind=7,
    argv=["--arch", "ClusterNet5g", "--dataset", "synthetic", "--output_k", "20", "--num_sub_heads", "2",
          "--input_sz", "32", "--batchnorm_track", "--lr_schedule", "2"]),
  "cluster_sobel_twohead": dict(
    module="code.scripts.cluster.cluster_sobel_twohead", arch="ClusterNet5gTwoHead", in_ch=1, model_ind=8,
    argv=["--arch", "ClusterNet5gTwoHead", "--dataset", "synthetic", "--output_k_A", "20", "--output_k_B", str(GT_K),
          "--num_sub_heads", "2", "--input_sz", "32", "--batchnorm_track", "--head_A_first", "--double_eval",
          "--select_sub_head_on_loss", "--lr_schedule", "2"]),
  "cluster_greyscale": dict(
    module="code.scripts.cluster.cluster_greyscale", arch="ClusterNet6c", in_ch=1, model_ind=9,
    argv=["--arch", "ClusterNet6c", "--dataset", "MNIST", "--output_k", "20", "--num_sub_heads", "2",
          "--input_sz", "24", "--batchnorm_track", "--mode", "IID+"]),
  "cluster_greyscale_twohead": dict(
    module="code.scripts.cluster.cluster_greyscale_twohead", arch="ClusterNet6cTwoHead", in_ch=1, model_ind=10,
    argv=["--arch", "ClusterNet6cTwoHead", "--dataset", "MNIST", "--output_k_A", "20", "--output_k_B", str(GT_K),
          "--num_sub_heads", "2", "--input_sz", "24", "--batchnorm_track"]),
  "segmentation": dict(
    module="code.scripts.segmentation.segmentation", arch="SegmentationNet10a", in_ch=4, model_ind=11, gt_k=3,
    argv=["--arch", "SegmentationNet10a", "--dataset", "Potsdam", "--output_k", "6", "--num_sub_heads", "1",
          "--input_sz", "24", "--batchnorm_track", "--mode", "IID+", "--include_rgb", "--half_T_side_dense", "1",
          "--use_uncollapsed_loss"]),
  "segmentation_twohead": dict(
    module="code.scripts.segmentation.segmentation_twohead", arch="SegmentationNet10aTwoHead", in_ch=4, model_ind=12,
    gt_k=3,
    argv=["--arch", "SegmentationNet10aTwoHead", "--dataset", "Potsdam", "--output_k_A", "6", "--output_k_B", "3",
          "--num_sub_heads", "1", "--input_sz", "24", "--batchnorm_track", "--include_rgb", "--half_T_side_dense", "1"]),
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
  "eval": ce._original_match is iic_amd.eval_metrics._original_match,
  "n_patched": len(done), "n_patches": len(install.PATCHES),
}
assert all(v for k, v in bound.items() if k not in ("n_patched", "n_patches")), bound


# ---- synthetic stand-ins for the data layer (same entry points as code/utils/*/data.py) ----------------------------
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
  config.train_partitions, config.mapping_assignment_partitions, config.mapping_test_partitions = [True], [True], [False]
  dl, mapping = _cluster_loaders(config, 0)
  return dl, mapping, mapping


def cluster_twohead_create_dataloaders(config):
  config.train_partitions_head_A = config.train_partitions_head_B = [True, False]
  config.mapping_assignment_partitions = config.mapping_test_partitions = [True, False]
  dl_a, mapping = _cluster_loaders(config, 0)
  dl_b, _ = _cluster_loaders(config, 100)
  return dl_a, dl_b, mapping, mapping


def segmentation_create_dataloaders(config):
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
  emask = torch.ones(NUM_IMGS, S, S, dtype=torch.uint8)
  mapping = [(base[i:i + bs], labels[i:i + bs], emask[i:i + bs]) for i in range(0, NUM_IMGS, bs)]
  return dataloaders, mapping, mapping


cdata = types.ModuleType("code.utils.cluster.data")
cdata.cluster_create_dataloaders = cluster_create_dataloaders
cdata.cluster_twohead_create_dataloaders = cluster_twohead_create_dataloaders
sys.modules["code.utils.cluster.data"] = cdata
sdata = types.ModuleType("code.utils.segmentation.data")
sdata.segmentation_create_dataloaders = segmentation_create_dataloaders
sys.modules["code.utils.segmentation.data"] = sdata

import random  # noqa: E402
random.seed(0)
np.random.seed(0)
torch.manual_seed(0)

name = spec["module"].rsplit(".", 1)[1]
sys.argv = [name, "--model_ind", str(spec["model_ind"]), "--dataset_root", "/nonexistent", "--gt_k", str(gt_k),
            "--lr", "0.001", "--num_epochs", str(NUM_EPOCHS), "--batch_sz", str(BATCH), "--num_dataloaders", "3",
            "--out_root", out_root, "--save_freq", "1"] + spec["argv"] + (["--restart"] if RESTART else [])
rc, err = None, None
_fp32 = ops.fp32_mode() if os.environ.get("IIC_FP32_MODE", "0") == "1" else None      # the exact-fp32 parity kernels
if _fp32 is not None:
  _fp32.__enter__()
try:
  py2compat.run_script(spec["module"])
except SystemExit as e:
  rc = e.code
except Exception as e:       # noqa: BLE001  (reported to the test, with the traceback on stderr)
  import traceback
  traceback.print_exc()
  err = "%s: %s" % (type(e).__name__, e)
if _fp32 is not None:
  _fp32.__exit__(None, None, None)
torch.cuda.synchronize()
odir = os.path.join(out_root, str(spec["model_ind"]))
res = {"script": SCRIPT, "bound": bound, "exit": rc, "error": err,
       "files": sorted(os.listdir(odir)) if os.path.isdir(odir) else []}
cp = os.path.join(odir, "config.pickle")
if os.path.exists(cp):
  with open(cp, "rb") as f:
    cfg = pickle.load(f)
  for k in ("epoch_loss", "epoch_loss_no_lamb", "epoch_acc", "epoch_loss_head_A", "epoch_loss_head_B",
            "epoch_loss_no_lamb_head_A", "epoch_loss_no_lamb_head_B", "double_eval_acc", "last_epoch"):
    if hasattr(cfg, k):
      v = getattr(cfg, k)
      res[k] = [float(x).hex() for x in v] if isinstance(v, (list, tuple)) else v
print("IIC_GPU_DRIVER_RESULT " + json.dumps(res))
