"""Helper of tests/test_gpu_dist.py (run under torchrun, 2 gloo ranks sharing cuda:0): the product's segmentation
losses on a rank's shard of the batch -- FULL-batch masks / affines handed over as the unchanged scripts do
(segmentation_twohead.py:318-325), sliced by iic_amd.dist.shard_like; raw per-shift joints all-reduced -- against the
same kernels on the full batch in one process and against the float64 oracle.  Prints one JSON line per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  torch.cuda.set_device(0)
  dev = torch.device("cuda", 0)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from iic_amd import dist as idist
  from iic_amd import seg_losses
  from oracle import iid_oracle
  bn, k, h, w = 6, 5, 24, 32
  g = torch.Generator().manual_seed(3)
  x1 = torch.softmax(2.0 * torch.randn(bn, k, h, w, generator=g), dim=1)
  x2 = torch.softmax(2.0 * torch.randn(bn, k, h, w, generator=g), dim=1)
  aff = torch.zeros(bn, 2, 3)
  aff[:, 0, 0] = torch.where(torch.rand(bn, generator=g) < 0.5, -1.0, 1.0)
  aff[:, 1, 1] = 1.0
  mask = (torch.rand(bn, h, w, generator=g) < 0.7).float()
  res = []
  for name, fn, ofn in (("uncollapsed", seg_losses.IID_segmentation_loss_uncollapsed, iid_oracle.IID_segmentation_loss_uncollapsed),
                        ("collapsed", seg_losses.IID_segmentation_loss, iid_oracle.IID_segmentation_loss)):
    for T in (1, 2):
      kw = dict(lamb=1.0, half_T_side_dense=T, half_T_side_sparse_min=0, half_T_side_sparse_max=0)
      # full batch, no collectives
      idist.disable()
      idist.SHARD_INPUTS[0] = False
      f1, f2 = x1.to(dev).requires_grad_(True), x2.to(dev).requires_grad_(True)
      full, _ = fn(f1, f2, all_affine2_to_1=aff.to(dev), all_mask_img1=mask.to(dev), **kw)
      full.backward()
      # this rank's shard, full-batch side inputs
      idist.enable()
      idist.SHARD_INPUTS[0] = True
      lo, hi = idist.shard_rows(bn)
      l1, l2 = x1[lo:hi].to(dev).requires_grad_(True), x2[lo:hi].to(dev).requires_grad_(True)
      loc, _ = fn(l1, l2, all_affine2_to_1=aff.to(dev), all_mask_img1=mask.to(dev), **kw)
      loc.backward()
      # float64 oracle, full batch
      o1, o2 = x1.double().requires_grad_(True), x2.double().requires_grad_(True)
      ref, _ = ofn(o1, o2, all_affine2_to_1=aff.double(), all_mask_img1=mask.double(), lamb=1.0, half_T_side_dense=T)
      ref.backward()
      gn = float(o1.grad.norm())
      res.append({"variant": name, "T": T, "loss_local": float(loc), "loss_full": float(full), "loss_ref64": float(ref),
                  "d1_vs_full": float((l1.grad - f1.grad[lo:hi]).norm() / f1.grad[lo:hi].norm()),
                  "d2_vs_full": float((l2.grad - f2.grad[lo:hi]).norm() / f2.grad[lo:hi].norm()),
                  "d1_vs_ref64": float((l1.grad.cpu().double() - o1.grad[lo:hi]).norm() / o1.grad[lo:hi].norm()),
                  "d2_vs_ref64": float((l2.grad.cpu().double() - o2.grad[lo:hi]).norm() / o2.grad[lo:hi].norm()),
                  "rows": [lo, hi], "gnorm": gn})
  print(json.dumps({"rank": rank, "results": res}), flush=True)
  idist.SHARD_INPUTS[0] = False
  idist.disable()
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
