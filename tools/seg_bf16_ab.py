"""Segmentation-loss contraction kernels: exact-fp32 MFMA (16x16x4 f32) against the three-term bf16 split on
v_mfma_f32_16x16x32_bf16 (iic_debug_seg_bf16, round 6) at the BASELINE shapes -- ms per launch, and the difference of
the results against a float64 evaluation of the same joint on a sub-sample."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch  # noqa: E402

from iic_amd import _lib  # noqa: E402
from iic_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

CFG = {"coco3 k15 T10": (120, 15, 128, 128, 10, 0.6), "potsdam3 k24 T10": (75, 24, 200, 200, 10, 1.0),
       "coco3 head B k3 T10": (120, 3, 128, 128, 10, 0.6), "potsdam k24 T5": (75, 24, 200, 200, 5, 1.0)}


def run(which, reps=5):
  bn, k, h, w, T, dens = CFG[which]
  dev = torch.device("cuda:0")
  g = torch.Generator().manual_seed(0)
  x1 = torch.softmax(2 * torch.randn(bn, k, h, w, generator=g), 1).to(dev)
  x2 = torch.softmax(2 * torch.randn(bn, k, h, w, generator=g), 1).to(dev)
  mask = (torch.rand(bn, h, w, generator=g) < dens).float().to(dev)
  flips = torch.tensor([[i & 1, 0] for i in range(bn)], dtype=torch.int32).to(dev)
  L = lib()
  D = ctypes.CDLL(_lib.LIB_PATH)
  nq = 2 * T + 1
  H = nq * nq
  ns = L.iic_seg_joint_nsplit(bn, h, k, T)
  dR1 = torch.randn(H, k, k, generator=g).to(dev)
  dR2 = torch.randn(H, k, k, generator=g).to(dev)
  g1 = torch.randn(H, generator=g).to(dev)
  g2 = torch.randn(H, generator=g).to(dev)
  ws = torch.empty(L.iic_seg_grad_workspace_bytes(k, T) // 4, device=dev)
  flops = 2.0 * bn * h * w * H * k * k
  res = {}
  for mode in (0, 1):
    D.iic_debug_seg_bf16(mode)
    part = torch.empty((ns, H, k, k), device=dev)
    o1, o2 = torch.empty_like(x1), torch.empty_like(x1)

    def joint():
      check(L.iic_seg_joint_raw(ptr(x1), ptr(x2), ptr(mask), ptr(flips), ptr(part), bn, k, h, w, T, ns, stream_ptr()), "j")

    def grad(wh, out):
      src = x2 if wh == 0 else x1
      check(L.iic_seg_grad(ptr(src), ptr(mask), ptr(flips), ptr(dR1), ptr(dR2), ptr(g1), ptr(g2), ptr(out), bn, k, h, w, T,
                           wh, 0, ptr(ws), stream_ptr()), "g")
    t = []
    for fn in (joint, lambda: grad(0, o1), lambda: grad(1, o2)):
      fn(); torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(reps):
        fn()
      e1.record(); torch.cuda.synchronize()
      t.append(e0.elapsed_time(e1) / reps)
    res[mode] = (t, part.double().sum(0).clone(), o1.clone(), o2.clone())
  D.iic_debug_seg_bf16(1)
  a, b = res[0], res[1]
  dj = float((a[1] - b[1]).abs().max() / a[1].abs().max())
  d1 = float((a[2] - b[2]).norm() / a[2].norm())
  d2 = float((a[3] - b[3]).norm() / a[3].norm())
  print("%-22s joint %7.3f -> %7.3f ms (%.2fx)  grad dx1 %7.3f -> %7.3f (%.2fx)  grad dx2 %7.3f -> %7.3f (%.2fx)  | fp32 MFMA %5.1f TF/s -> "
        "%6.1f TF/s effective | bf16-split vs fp32-MFMA: joint max rel %.2e, dx1 rel L2 %.2e, dx2 %.2e" % (
          which, a[0][0], b[0][0], a[0][0] / b[0][0], a[0][1], b[0][1], a[0][1] / b[0][1], a[0][2], b[0][2], a[0][2] / b[0][2],
          3 * flops / sum(a[0]) / 1e9, 3 * flops / sum(b[0]) / 1e9, dj, d1, d2))


if __name__ == "__main__":
  for name in CFG:
    run(name)
