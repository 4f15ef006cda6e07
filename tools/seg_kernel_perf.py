"""Microbenchmark of the segmentation-loss contraction kernels at the BASELINE shapes (no network):
  python tools/seg_kernel_perf.py [potsdam|coco|potsdamB] [reps]
prints ms per launch of iic_seg_joint_raw and of each iic_seg_grad, and the fp32-MFMA rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iic_amd._lib import check, lib, ptr, stream_ptr

CFG = {"potsdam": (75, 24, 200, 200, 10, 1.0), "coco": (120, 15, 128, 128, 10, 0.6),
       "potsdamB": (75, 3, 200, 200, 10, 1.0), "potsdamT1": (75, 24, 200, 200, 1, 1.0)}


def main():
  which = sys.argv[1] if len(sys.argv) > 1 else "potsdam"
  reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
  bn, k, h, w, T, dens = CFG[which]
  dev = torch.device("cuda:0")
  g = torch.Generator().manual_seed(0)
  x1 = torch.softmax(torch.randn(bn, k, h, w, generator=g), 1).to(dev)
  x2 = torch.softmax(torch.randn(bn, k, h, w, generator=g), 1).to(dev)
  mask = (torch.rand(bn, h, w, generator=g) < dens).float().to(dev)
  flips = torch.tensor([[i & 1, 0] for i in range(bn)], dtype=torch.int32).to(dev)
  L = lib()
  nq = 2 * T + 1
  H = nq * nq
  ns = L.iic_seg_joint_nsplit(bn, h, k, T)
  part = torch.empty((ns, H, k, k), device=dev)
  dR1 = torch.randn(H, k, k, generator=g).to(dev)
  dR2 = torch.randn(H, k, k, generator=g).to(dev)
  g1 = torch.randn(H, generator=g).to(dev)
  g2 = torch.randn(H, generator=g).to(dev)
  ws = torch.empty(L.iic_seg_grad_workspace_bytes(k, T) // 4, device=dev)
  out = torch.empty_like(x1)
  flops = 2.0 * bn * h * w * H * k * k

  def joint():
    check(L.iic_seg_joint_raw(ptr(x1), ptr(x2), ptr(mask), ptr(flips), ptr(part), bn, k, h, w, T, ns, stream_ptr()), "j")

  def grad(wh):
    src = x2 if wh == 0 else x1
    check(L.iic_seg_grad(ptr(src), ptr(mask), ptr(flips), ptr(dR1), ptr(dR2), ptr(g1), ptr(g2), ptr(out), bn, k, h, w, T,
                         wh, 0, ptr(ws), stream_ptr()), "g")

  for name, fn in (("joint", joint), ("grad dx1", lambda: grad(0)), ("grad dx2", lambda: grad(1))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-9s %-9s %8.3f ms  %6.1f TF/s  (%.3f of 157.3)" % (which, name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3))


if __name__ == "__main__":
  main()
