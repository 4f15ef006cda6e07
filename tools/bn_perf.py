"""A/B timing of the BatchNorm backward passes at the north-star layer shapes (660 images/view):
first-generation (row-per-block) vs second-generation (pixel walkers) kernels.
GB/s = algorithmic bytes: reduce reads dout + y, apply reads dout + y and writes dy (bf16)."""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
from iic_amd import ops, _lib   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=660)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--blocks", type=int, nargs="*", default=[1024])
a = ap.parse_args()
L = ctypes.CDLL(_lib.LIB_PATH)
d = torch.device("cuda:0")
shapes = [(49, 64), (25, 128), (13, 256), (7, 512)]


def timeit(fn):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(a.iters):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / a.iters * 1e3     # us


tot = {}
for (H, C) in shapes:
  N, P = a.n, 1
  shape = (N, H + 2, H + 2, C)
  dout = torch.randn(shape, device=d).to(torch.bfloat16)
  y = torch.randn(shape, device=d).to(torch.bfloat16)
  act = torch.relu(torch.randn(shape, device=d)).to(torch.bfloat16)
  dy = torch.zeros(shape, dtype=torch.bfloat16, device=d)
  coef = torch.randn(4, C, device=d)
  bcoef = torch.randn(3, C, device=d)
  sums = ops.new_stats(C, d)
  nbytes = N * H * H * C * 2
  for label, gen, blocks in [("v1", 0, 0)] + [("v2/%d" % b, 2, b) for b in a.blocks]:
    L.iic_debug_bn_v2(gen, blocks)
    for mode, aa, mc in (("act", act, None), ("from_y", None, coef)):
      t_r = timeit(lambda: ops.bn_bwd_reduce(dout, aa, y, sums, N, H, H, P, C, mask_coef=mc))
      t_a = timeit(lambda: ops.bn_bwd_apply(dout, aa, y, bcoef, dy, N, H, H, P, C, mask_coef=mc))
      rb = (3 if aa is not None else 2) * nbytes
      ab = (4 if aa is not None else 3) * nbytes
      print("H=%2d C=%3d %-8s %-6s reduce %7.1f us %5.2f TB/s | apply %7.1f us %5.2f TB/s"
            % (H, C, label, mode, t_r, rb / t_r / 1e6, t_a, ab / t_a / 1e6))
      tot[(label, "r")] = tot.get((label, "r"), 0) + t_r
      tot[(label, "a")] = tot.get((label, "a"), 0) + t_a
L.iic_debug_bn_v2(1, 1024)
print({k: round(v, 1) for k, v in tot.items()})
