"""(needs the ablation build of the library: make -C iic_amd/csrc clean && make -C iic_amd/csrc ABL=1)
Where a tile's cycles go in the persistent 64 -> 64 convolution (conv_igemm_p64_kernel, ResNet layer1 of
ClusterNet5g): the kernel's PROF build (iic_debug_set_ablate(8), results unchanged) sums, per workgroup, the
s_memtime cycles of each phase of its tile loop.

python tools/p64_phases.py [--n 660] [--wide 0|1] [--spread 0|1] [--bwd]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import numpy as np
import torch

from iic_amd import _lib, geom, ops

PH = ["wait patch + barrier A", "store of tile t-1", "DMA issue + row tables", "K loop (9 taps x 4 k-steps)",
      "barrier B", "accumulators -> LDS (+stats)"]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--n", type=int, default=660)
  ap.add_argument("--wide", type=int, default=0)
  ap.add_argument("--spread", type=int, default=1)
  ap.add_argument("--bwd", action="store_true", help="backward-data with the residual gradient + ReLU mask epilogue")
  args = ap.parse_args()
  L = _lib.lib()
  for name in ("iic_debug_p64_prof", "iic_debug_p64_wide", "iic_debug_p64_spread", "iic_debug_set_ablate"):
    getattr(L, name).restype = None
  L.iic_debug_p64_prof.argtypes = [ctypes.c_void_p]
  dev = torch.device("cuda:0")
  N, H, C = args.n, 49, 64
  spec = geom.ConvSpec(C, C, 3, 1, 1)
  g = geom.fwd_geom(spec, N, H, H, 1, 1)
  x = torch.randn(N, H + 2, H + 2, C, device=dev).to(torch.bfloat16)
  x[:, 0], x[:, -1], x[:, :, 0], x[:, :, -1] = 0, 0, 0, 0
  y = torch.zeros(N, H + 2, H + 2, C, device=dev, dtype=torch.bfloat16)
  w = torch.randn(C, C, 3, 3, device=dev) * 0.05
  pw = ops.PreppedWeights(w)
  st = ops.new_stats(C, dev)
  kw = dict(stats=st)
  if args.bwd:
    rg = torch.randn_like(x)
    kw = dict(res_grad=rg, res_act=x, premask=True)
  L.iic_debug_p64_wide(args.wide)
  L.iic_debug_p64_spread(args.spread)
  ncu = 2 * torch.cuda.get_device_properties(0).multi_processor_count
  buf = torch.zeros(ncu * 8, device=dev, dtype=torch.int64)

  def launch():
    ops.conv_igemm(g, x, pw[0], y, **kw)

  def timed(iters=10):
    for _ in range(3):
      launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
      launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
  us_plain = timed()
  L.iic_debug_set_ablate(8)
  L.iic_debug_p64_prof(buf.data_ptr())
  us_prof = timed()
  torch.cuda.synchronize()
  L.iic_debug_set_ablate(0)
  L.iic_debug_p64_prof(None)
  q = buf.cpu().numpy().reshape(ncu, 8).astype(np.float64)
  q = q[q[:, 6] > 0]
  tiles = q[:, 6]
  per_tile = q[:, :6].sum(0) / tiles.sum()
  loop = q[:, 7].sum() / tiles.sum()
  flops = 2.0 * N * H * H * C * C * 9
  print("layer1 3x3 64->64 @49, %d images, %s, wide=%d spread=%d: %.1f us per launch (%.0f TF/s); with stamps %.1f us"
        % (N, "backward-data + residual epilogue" if args.bwd else "forward", args.wide, args.spread, us_plain,
           flops / us_plain / 1e6, us_prof))
  print("%d workgroups, %.1f tiles each; cycles per tile (s_memtime, mean over all tiles): %.0f" % (len(q), tiles.mean(), loop))
  for n, v in zip(PH, per_tile):
    print("  %-34s %8.0f  %5.1f %%" % (n, v, 100 * v / loop))
  mf = 72 * 32 * (2 if not args.wide else 2) * (1 if not args.wide else 1)
  print("  (matrix pipe alone: %d MFMAs per wave and tile x 32 cycles x %d waves per SIMD = %d cycles)"
        % (72 if not args.wide else 144, 2 if not args.wide else 1, 72 * 32 * 2))


if __name__ == "__main__":
  main()
