"""Weight-gradient DMA kernel: plain 12-wave K loop against the software-pipelined one (iic_debug_wgrad_swp, round 6)
at the north-star layer shapes (660 images), interleaved, same process; the results must be bit-identical (same MFMAs
in the same order)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch  # noqa: E402

from iic_amd import _lib, geom, ops  # noqa: E402

LAYERS = [("l1 3x3 64->64 @49", 64, 64, 49), ("l2 3x3 128->128 @25", 128, 128, 25), ("l3 3x3 256->256 @13", 256, 256, 13),
          ("l4 3x3 512->512 @7", 512, 512, 7)]
COUNT = [6, 7, 11, 5]


def timeit(fn, iters=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / iters


def main():
  L = ctypes.CDLL(_lib.LIB_PATH)
  dev = torch.device("cuda", 0)
  N = int(sys.argv[1]) if len(sys.argv) > 1 else 660
  tot = [0.0, 0.0]
  for (name, cin, cout, H), cnt in zip(LAYERS, COUNT):
    spec = geom.ConvSpec(cin, cout, 3, 1, 1)
    gf = geom.fwd_geom(spec, N, H, H, 1, 1)
    x = torch.randn(N, H + 2, H + 2, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, H + 2, H + 2, cout, device=dev).to(torch.bfloat16)
    x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
    dy[:, 0] = 0; dy[:, -1] = 0; dy[:, :, 0] = 0; dy[:, :, -1] = 0
    flops = 2.0 * N * H * H * cout * cin * 9
    t, out = {}, {}
    for rep in range(2):
      for v in (0, 1):
        L.iic_debug_wgrad_swp(v)
        tt = timeit(lambda: ops.conv_wgrad(gf, x, dy, 9, True))
        t[v] = min(t.get(v, 1e9), tt)
        out[v] = ops.conv_wgrad(gf, x, dy, 9, True).clone()
    torch.cuda.synchronize()
    same = bool(torch.equal(out[0], out[1]))
    print("%-24s plain %7.1f us %6.0f TF/s | pipelined %7.1f us %6.0f TF/s (%.3fx)  bit-identical %s" % (
      name, t[0], flops / t[0] / 1e6, t[1], flops / t[1] / 1e6, t[0] / t[1], same))
    tot[0] += cnt * t[0]; tot[1] += cnt * t[1]
  print("per view (x layer counts, incl. the reduce pass): plain %.2f ms, pipelined %.2f ms" % (tot[0] / 1e3, tot[1] / 1e3))
  L.iic_debug_wgrad_swp(1)


if __name__ == "__main__":
  main()
