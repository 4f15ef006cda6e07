"""Segmentation-loss contraction kernels at 33 <= k <= 48 (round 6, VERDICT r5 missing #5): the streaming kernels with three
class tiles against the element-wise generic kernels they replace (iic_debug_seg_stream), at the reference's
overclustering shapes -- COCO-Stuff k_A 45, T 10, 128 x 128, batch 60 (commands.txt:80) and Potsdam k_A 36, T 5,
200 x 200, batch 60 (commands.txt:89) -- the runs where the reference's loss takes 19.8 / 21.6 s per batch."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch  # noqa: E402

from iic_amd import _lib  # noqa: E402
from iic_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

CFG = {"coco-stuff k45 T10 128^2 b60": (60, 45, 128, 128, 10, 0.6), "potsdam k36 T5 200^2 b60": (60, 36, 200, 200, 5, 1.0)}


def main():
  L, D = lib(), ctypes.CDLL(_lib.LIB_PATH)
  dev = torch.device("cuda:0")
  for name, (bn, k, h, w, T, dens) in CFG.items():
    g = torch.Generator().manual_seed(0)
    x1 = torch.softmax(2 * torch.randn(bn, k, h, w, generator=g), 1).to(dev)
    x2 = torch.softmax(2 * torch.randn(bn, k, h, w, generator=g), 1).to(dev)
    mask = (torch.rand(bn, h, w, generator=g) < dens).float().to(dev)
    flips = torch.tensor([[i & 1, 0] for i in range(bn)], dtype=torch.int32).to(dev)
    nq = 2 * T + 1
    H = nq * nq
    ns = L.iic_seg_joint_nsplit(bn, h, k, T)
    dR1, dR2 = torch.randn(H, k, k, generator=g).to(dev), torch.randn(H, k, k, generator=g).to(dev)
    g1, g2 = torch.randn(H, generator=g).to(dev), torch.randn(H, generator=g).to(dev)
    ws = torch.empty(L.iic_seg_grad_workspace_bytes(k, T) // 4, device=dev)
    flops = 2.0 * bn * h * w * H * k * k
    t = {}
    for mode in (0, 1):
      D.iic_debug_seg_stream(mode)
      part = torch.empty((ns, H, k, k), device=dev)
      out = torch.empty_like(x1)

      def joint():
        check(L.iic_seg_joint_raw(ptr(x1), ptr(x2), ptr(mask), ptr(flips), ptr(part), bn, k, h, w, T, ns, stream_ptr()), "j")

      def grad(wh):
        check(L.iic_seg_grad(ptr(x2 if wh == 0 else x1), ptr(mask), ptr(flips), ptr(dR1), ptr(dR2), ptr(g1), ptr(g2), ptr(out),
                             bn, k, h, w, T, wh, 0, ptr(ws), stream_ptr()), "g")
      tt = []
      for fn in (joint, lambda: grad(0), lambda: grad(1)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
          fn()
        e1.record(); torch.cuda.synchronize()
        tt.append(e0.elapsed_time(e1) / 3)
      t[mode] = tt
    D.iic_debug_seg_stream(1)
    a, b = t[0], t[1]
    print("%-30s joint %8.2f -> %7.2f ms  grad dx1 %8.2f -> %7.2f  grad dx2 %8.2f -> %7.2f | forward + backward of the loss %8.2f -> %7.2f ms "
          "(%.2fx), %5.1f TF/s = %.2f of the fp32 MFMA peak" % (name, a[0], b[0], a[1], b[1], a[2], b[2], sum(a), sum(b), sum(a) / sum(b),
                                                                 3 * flops / sum(b) / 1e9, 3 * flops / sum(b) / 1e9 / 157.3))


if __name__ == "__main__":
  main()
