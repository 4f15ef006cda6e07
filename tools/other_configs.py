"""Train-step timing of the other BASELINE.json configs on one MI355X (secondary measurements for
DESIGN.md; bench.py stays the north-star config).  Synthetic resident inputs, same timed region as
the reference scripts' train step (forward x2, loss(es), backward, Adam)."""
import argparse, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch
from iic_amd import archs
from iic_amd.losses import IID_loss_heads, IID_loss
from iic_amd.optim import Adam
from iic_amd.seg_losses import IID_segmentation_loss_uncollapsed

dev = torch.device("cuda:0")


def timeit(step, steps=5, warm=2):
  for _ in range(warm):
    step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / steps


def mnist_6c_twohead(bn=700):
  """BASELINE configs[0]: MNIST 24x24 ClusterNet6cTwoHead, k_A 50 / k_B 10, 5 sub-heads, batch 700
  (cluster_greyscale_twohead.py: head A step + head B step per batch pair)."""
  cfg = types.SimpleNamespace(in_channels=1, input_sz=24, batchnorm_track=True, num_sub_heads=5,
                              output_k_A=50, output_k_B=10)
  net = archs.ClusterNet6cTwoHead(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=1e-4)
  g = torch.Generator().manual_seed(0)
  x = torch.rand(bn, 1, 24, 24, generator=g).to(dev)
  xt = torch.clamp(torch.flip(x, dims=[3]) * 0.9 + 0.05, 0, 1)

  def step():
    for head in ("A", "B"):
      net.zero_grad(set_to_none=True)
      a = net(x, head=head)
      b = net(xt, head=head)
      loss = sum(IID_loss(a[i], b[i])[0] for i in range(5)) / 5
      loss.backward()
      opt.step()
  return bn, timeit(step)


def potsdam_10a_twohead(bn=75, sz=200, T=1):
  """BASELINE configs[3]: Potsdam-3 200x200 SegmentationNet10aTwoHead, k_A 24 / k_B 3, batch 75,
  uncollapsed loss with half_T_side_dense = T."""
  cfg = types.SimpleNamespace(in_channels=4, input_sz=sz, batchnorm_track=True, num_sub_heads=1,
                              output_k_A=24, output_k_B=3)
  net = archs.SegmentationNet10aTwoHead(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=1e-4)
  g = torch.Generator().manual_seed(0)
  x = torch.rand(bn, 4, sz, sz, generator=g).to(dev)
  xt = torch.flip(x, dims=[3]) * 0.9 + 0.05
  aff = torch.zeros(bn, 2, 3, device=dev)
  aff[:, 0, 0] = -1.0
  aff[:, 1, 1] = 1.0
  mask = torch.ones(bn, sz, sz, device=dev)

  def step():
    for head in ("A", "B"):
      net.zero_grad(set_to_none=True)
      a = net(x, head=head)
      b = net(xt, head=head)
      loss, _ = IID_segmentation_loss_uncollapsed(a[0], b[0], all_affine2_to_1=aff, all_mask_img1=mask, lamb=1.0,
                                                  half_T_side_dense=T, half_T_side_sparse_min=0,
                                                  half_T_side_sparse_max=0)
      loss.backward()
      opt.step()
  return bn, timeit(step, steps=3, warm=1)


def cifar_6c(bn=700, k=280):
  """BASELINE configs[2] per-GPU shape: CIFAR 32x32 cropped 20 -> 24x24, --include_rgb (RGB + Sobel =
  5 channels), ClusterNet6c, output_k 280 (commands.txt:41), 5 sub-heads."""
  from iic_amd.transforms import sobel_process
  cfg = types.SimpleNamespace(in_channels=5, input_sz=24, batchnorm_track=True, num_sub_heads=5, output_k=k)
  net = archs.ClusterNet6c(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=1e-4)
  g = torch.Generator().manual_seed(0)
  x = torch.rand(bn, 4, 24, 24, generator=g).to(dev)          # RGB + grey, as the dataloader emits
  xt = torch.clamp(torch.flip(x, dims=[3]) * 0.9 + 0.05, 0, 1)

  def step():
    net.zero_grad(set_to_none=True)
    a = net(sobel_process(x, True))
    b = net(sobel_process(xt, True))
    loss = sum(IID_loss(a[i], b[i])[0] for i in range(5)) / 5
    loss.backward()
    opt.step()
  return bn, timeit(step)


def coco_10a_twohead(bn=120, sz=128, T=10):
  """BASELINE configs[4] shape: COCO-Stuff-3 128x128, RGB + Sobel = 5 channels, k_A 15 / k_B 3,
  batch 120 (commands.txt:74), half_T_side_dense 10, masked loss."""
  cfg = types.SimpleNamespace(in_channels=5, input_sz=sz, batchnorm_track=True, num_sub_heads=1,
                              output_k_A=15, output_k_B=3)
  net = archs.SegmentationNet10aTwoHead(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=1e-4)
  g = torch.Generator().manual_seed(0)
  x = torch.rand(bn, 5, sz, sz, generator=g).to(dev)
  xt = torch.flip(x, dims=[3]) * 0.9 + 0.05
  aff = torch.zeros(bn, 2, 3, device=dev)
  aff[:, 0, 0] = -1.0
  aff[:, 1, 1] = 1.0
  mask = (torch.rand(bn, sz, sz, generator=g) < 0.6).float().to(dev)

  def step():
    for head in ("A", "B"):
      net.zero_grad(set_to_none=True)
      a = net(x, head=head)
      b = net(xt, head=head)
      loss, _ = IID_segmentation_loss_uncollapsed(a[0], b[0], all_affine2_to_1=aff, all_mask_img1=mask, lamb=1.0,
                                                  half_T_side_dense=T, half_T_side_sparse_min=0,
                                                  half_T_side_sparse_max=0)
      loss.backward()
      opt.step()
  return bn, timeit(step, steps=3, warm=1)


def stl_5g_twohead(bn=700, sz=64):
  """commands.txt:35-style two-head STL10 run: ClusterNet5gTwoHead, 64x64, k_A 70 / k_B 10."""
  from iic_amd.transforms import sobel_process
  cfg = types.SimpleNamespace(in_channels=2, input_sz=sz, batchnorm_track=True, num_sub_heads=5,
                              output_k_A=70, output_k_B=10)
  net = archs.ClusterNet5gTwoHead(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=1e-4)
  g = torch.Generator().manual_seed(0)
  x = torch.rand(bn, 1, sz, sz, generator=g).to(dev)
  xt = torch.clamp(torch.flip(x, dims=[3]) * 0.9 + 0.05, 0, 1)

  def step():
    for head in ("A", "B"):
      net.zero_grad(set_to_none=True)
      a = net(sobel_process(x, False), head=head)
      b = net(sobel_process(xt, False), head=head)
      loss = sum(IID_loss(a[i], b[i])[0] for i in range(5)) / 5
      loss.backward()
      opt.step()
  return bn, timeit(step, steps=3, warm=1)


if __name__ == "__main__":
  ap = argparse.ArgumentParser()
  ap.add_argument("--which", default="mnist,cifar,stl2h,potsdam,coco")
  ap.add_argument("--bd-one-wg", type=int, default=0)
  a = ap.parse_args()
  if a.bd_one_wg:
    import ctypes
    from iic_amd import _lib
    ctypes.CDLL(_lib.LIB_PATH).iic_debug_bd_one_wg(1)
  if "mnist" in a.which:
    bn, t = mnist_6c_twohead()
    print("MNIST 24x24 ClusterNet6cTwoHead batch %d (head A + head B steps): %.2f ms -> %.0f paired-images/s" % (bn, 1e3 * t, bn / t))
  if "cifar" in a.which:
    bn, t = cifar_6c()
    print("CIFAR 24x24x5 ClusterNet6c k=280 batch %d: %.2f ms -> %.0f paired-images/s" % (bn, 1e3 * t, bn / t))
  if "stl2h" in a.which:
    bn, t = stl_5g_twohead()
    print("STL10 64x64 ClusterNet5gTwoHead batch %d (head A + head B steps): %.1f ms -> %.0f pairs/s" % (bn, 1e3 * t, bn / t))
  if "coco" in a.which:
    bn, t = coco_10a_twohead()
    print("COCO-Stuff-3 128x128x5 SegmentationNet10aTwoHead batch %d T=10 (head A + head B steps): %.1f ms -> %.1f pairs/s" % (bn, 1e3 * t, bn / t))
  if "potsdam" in a.which:
    for T in (1, 10):
      bn, t = potsdam_10a_twohead(T=T)
      print("Potsdam-3 200x200 SegmentationNet10aTwoHead batch %d T=%d (head A + head B steps): %.1f ms -> %.1f pairs/s" % (bn, T, 1e3 * t, bn / t))
