"""First-layer kernels of the VGG-style trunks, generation A/B at the BASELINE shapes:
firstconv_fwd / firstconv_wgrad of vgg.hip against firstconv_fwd2 / firstconv_wgrad2 of firstconv2.hip
(iic_debug_firstconv_v2 in the instrumented library), HIP-event times per launch, outputs compared."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch  # noqa: E402

from iic_amd import _lib, ops  # noqa: E402

SHAPES = [("potsdam3 75x4x200x200 3x3", 75, 4, 200, 200, 3), ("coco3 120x5x128x128 3x3", 120, 5, 128, 128, 3),
          ("mnist6c 700x1x24x24 5x5", 700, 1, 24, 24, 5), ("cifar6c 700x5x24x24 5x5", 700, 5, 24, 24, 5)]


def timed(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / n


def main():
  L = ctypes.CDLL(_lib.LIB_PATH)
  dev = torch.device("cuda", 0)
  for name, N, C, H, W, K in SHAPES:
    pad, P = (K - 1) // 2, 2 if K == 5 else 1
    g = torch.Generator().manual_seed(0)
    x = torch.rand(N, C, H, W, generator=g).to(dev)
    w = (torch.randn(64, C, K, K, generator=g) * 0.2).to(dev)
    dy = ops.pt_from_nchw(torch.randn(N, 64, H, W, generator=g).to(dev).to(torch.bfloat16).float(), P)
    res = {}
    for v2 in (0, 1):
      L.iic_debug_firstconv_v2(v2)
      out = torch.zeros((N, H + 2 * P, W + 2 * P, 64), dtype=torch.bfloat16, device=dev)
      st = ops.new_stats(64, dev)
      t_f = timed(lambda: ops.firstconv_fwd(x, w, out, st, K, pad, P))
      t_w = timed(lambda: ops.firstconv_wgrad(x, dy, tuple(w.shape), K, pad, P))
      st = ops.new_stats(64, dev)
      ops.firstconv_fwd(x, w, out, st, K, pad, P)
      dW = ops.firstconv_wgrad(x, dy, tuple(w.shape), K, pad, P)
      torch.cuda.synchronize()
      res[v2] = (t_f, t_w, out.float().clone(), ops.stats_decode(st, 64).clone(), dW.clone())
    fl = 2.0 * N * H * W * C * K * K * 64
    by = 4.0 * N * C * H * W + 2.0 * N * H * W * 64
    a, b = res[0], res[1]
    print("%-28s fwd %7.1f -> %7.1f us (%.2fx; fp32-MFMA floor %.0f us, HBM floor %.0f us)   wgrad+reduce %7.1f -> %7.1f us (%.2fx)   "
          "max|dout| %.3g  rel|dstats| %.2g  rel|ddW| %.2g" % (
            name, a[0], b[0], a[0] / b[0], fl / 157.3e12 * 1e6, by / 6.3e12 * 1e6, a[1], b[1], a[1] / b[1],
            float((a[2] - b[2]).abs().max()), float(((a[3] - b[3]).abs() / (a[3].abs() + 1e-6)).max()),
            float((a[4] - b[4]).abs().max() / a[4].abs().max())))
  L.iic_debug_firstconv_v2(1)


if __name__ == "__main__":
  main()
