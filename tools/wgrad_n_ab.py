"""Does one weight-gradient launch over both views (1320 images) beat two launches of 660?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iic_amd import geom, ops
from tools.conv_perf import LAYERS, COUNT, timeit
dev = torch.device("cuda:0")
tot = {660: 0.0, 1320: 0.0}
for li, (name, cin, cout, K, s, p, H) in enumerate(LAYERS):
  line = "%-28s" % name
  for N in (660, 1320):
    spec = geom.ConvSpec(cin, cout, K, s, p)
    Ho = spec.out_size(H)
    gf = geom.fwd_geom(spec, N, H, H, 1, 1)
    x = torch.randn(N, H + 2, H + 2, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, Ho + 2, Ho + 2, cout, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.conv_wgrad(gf, x, dy, K * K, True), 20)
    flops = 2.0 * N * Ho * Ho * cout * cin * K * K
    line += " | N=%4d %7.1f us %6.0f TF/s" % (N, t, flops / t / 1e6)
    tot[N] += COUNT[li] * t * (2 if N == 660 else 1)
  print(line)
print("per step (both views): 2 x 660: %.2f ms, 1 x 1320: %.2f ms" % (tot[660] / 1e3, tot[1320] / 1e3))
