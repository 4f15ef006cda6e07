"""Long-K head GEMMs (ClusterNet6c: Linear 4608 -> 5 x k): exact-fp32 MFMA tiled kernel against the three-term bf16
split (iic_debug_gemm_x3) and the 128 x 128-tile fp32 kernel (iic_debug_gemm_t128; round 6), the three products of a head's forward / backward in their operand layouts;
us per launch and the error of each against a float64 product."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch  # noqa: E402

from iic_amd import _lib, ops  # noqa: E402


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
  D = ctypes.CDLL(_lib.LIB_PATH)
  dev = torch.device("cuda", 0)
  g = torch.Generator().manual_seed(0)
  for name, Nb, Kf, Ko in (("cifar6c k=280 x5", 700, 4608, 1400), ("mnist6c k=50 x5", 700, 4608, 250), ("net6c@64 k=10 x5", 700, 32768, 50)):
    feats = torch.randn(Nb, Kf, generator=g).to(dev)
    W = (torch.randn(Ko, Kf, generator=g) * 0.01).to(dev)
    dlog = torch.randn(Nb, Ko, generator=g).to(dev)
    prods = [("logits = feats W^T", lambda C: ops.gemm_f32(feats, Kf, 1, W, 1, Kf, C, Ko, Nb, Ko, Kf), (Nb, Ko), lambda: feats.double() @ W.double().t()),
             ("dfeats = dlogits W", lambda C: ops.gemm_f32(dlog, Ko, 1, W, Kf, 1, C, Kf, Nb, Kf, Ko), (Nb, Kf), lambda: dlog.double() @ W.double()),
             ("dW = dlogits^T feats", lambda C: ops.gemm_f32(dlog, 1, Ko, feats, Kf, 1, C, Kf, Ko, Kf, Nb), (Ko, Kf), lambda: dlog.double().t() @ feats.double())]
    for pn, fn, shape, ref in prods:
      want = ref()
      res = {}
      for mode, (t128, x3) in enumerate(((0, 0), (0, 1), (1, 0))):
        D.iic_debug_gemm_t128(t128)
        D.iic_debug_gemm_x3(x3)
        C = torch.empty(shape, device=dev)
        t = timed(lambda: fn(C))
        fn(C)
        torch.cuda.synchronize()
        res[mode] = (t, float((C.double() - want).abs().max() / want.abs().max()))
      fl = 2.0 * Nb * Kf * Ko
      print("%-18s %-22s 64-tile fp32 MFMA %7.1f us (%5.1f TF/s, err %.1e) | 64-tile bf16 x3 %7.1f us (err %.1e) | 128-tile fp32 MFMA %7.1f us "
            "(%5.1f TF/s = %.2f of the fp32 peak, err %.1e)  %.2fx" % (
              name, pn, res[0][0], fl / res[0][0] / 1e6, res[0][1], res[1][0], res[1][1], res[2][0], fl / res[2][0] / 1e6,
              fl / res[2][0] / 1e6 / 157.3, res[2][1], res[0][0] / res[2][0]))
  D.iic_debug_gemm_x3(0)
  D.iic_debug_gemm_t128(1)


if __name__ == "__main__":
  main()
