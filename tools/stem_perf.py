"""Timing of the stem kernels at the north-star shape (660 x 2 x 96 x 96); --ablate codes for stem_bwd2."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch
from iic_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=660)
ap.add_argument("--ablate", type=str, default="")
a = ap.parse_args()
d = torch.device("cuda:0")
N, S = a.n, 96
x = torch.randn(N, 2, S, S, device=d)
w = torch.randn(64, 2, 3, 3, device=d) * 0.3
gamma, beta = torch.ones(64, device=d), torch.zeros(64, device=d)
st = ops.new_stats(64, d)
ops.stem_stats(x, w, st)
coef = ops.bn_finalize(st, gamma, beta, None, None, None, 64, N * S * S, True)
So = S // 2 + 1
dp = torch.randn(N, So + 2, So + 2, 64, device=d).to(torch.bfloat16)
sums = ops.new_stats(64, d)
L = ctypes.CDLL(_lib.LIB_PATH)


def t(fn, it=5):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(it):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / it


print("stem_stats %.1f us" % t(lambda: ops.stem_stats(x, w, st)))
out = torch.zeros(N, So + 2, So + 2, 64, device=d, dtype=torch.bfloat16)
print("stem_apply_pool %.1f us" % t(lambda: ops.stem_apply_pool(x, w, coef, out)))
print("stem_bwd_fused (incl. patch sums) %.1f us" % t(lambda: ops.stem_bwd_fused(x, w, coef, dp, sums)))
for c in [int(v) for v in a.ablate.split(",") if v]:
  L.iic_debug_set_ablate(c)
  print("  ablate %2d: %.1f us" % (c, t(lambda: ops.stem_bwd_fused(x, w, coef, dp, sums))))
  L.iic_debug_set_ablate(0)
