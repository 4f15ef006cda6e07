"""A/B of the weights-direct kernel's tile height (256-row vs 128-row workgroup tiles) at the
north-star shapes: same-process timing, outputs compared bit for bit."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch
from iic_amd import _lib, geom, ops
from tools.conv_perf import LAYERS, COUNT, timeit

L = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
N = int(os.environ.get("N", "660"))
tot = {2: 0.0, 4: 0.0}
for li, (name, cin, cout, K, s, p, H) in enumerate(LAYERS):
  spec = geom.ConvSpec(cin, cout, K, s, p)
  Ho = spec.out_size(H)
  gf = geom.fwd_geom(spec, N, H, H, 1, 1)
  gb = geom.bwd_data_geoms(spec, N, H, H, 1, 1)
  x = torch.randn(N, H + 2, H + 2, cin, device=dev).to(torch.bfloat16)
  x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
  dy = torch.randn(N, Ho + 2, Ho + 2, cout, device=dev).to(torch.bfloat16)
  dy[:, 0] = 0; dy[:, -1] = 0; dy[:, :, 0] = 0; dy[:, :, -1] = 0
  w = torch.randn(cout, cin, K, K, device=dev) * 0.05
  pw = ops.PreppedWeights(w)
  flops = 2.0 * N * Ho * Ho * cout * cin * K * K
  res = {}
  line = "%-28s" % name
  for ms in (4, 2):
    L.iic_debug_bd_ms(ms)
    for g in [gf] + list(gb):
      g._frag_ok = None
    okf = ops.frag_supported(gf)
    okb = all(ops.frag_supported(g) for g in gb)
    y = torch.zeros(N, Ho + 2, Ho + 2, cout, device=dev, dtype=torch.bfloat16)
    dx = torch.zeros(N, H + 2, H + 2, cin, device=dev, dtype=torch.bfloat16)
    st = ops.new_stats(cout, dev)
    tf = timeit(lambda: ops.conv_igemm(gf, x, pw[0], y, stats=st), 20) if okf else float("nan")
    tb = timeit(lambda: [ops.conv_igemm(g, dy, pw[1], dx) for g in gb], 20) if okb else float("nan")
    res[ms] = (y.clone(), dx.clone())
    line += " | ms%d fwd %7.1f us %6.0f TF  bwd %7.1f us %6.0f TF" % (ms, tf, flops / tf / 1e6, tb, flops / tb / 1e6)
    if tf == tf: tot[ms] += COUNT[li] * tf
    if tb == tb: tot[ms] += COUNT[li] * tb
  same = torch.equal(res[2][0], res[4][0]), torch.equal(res[2][1], res[4][1])
  dmax = float((res[2][0].float() - res[4][0].float()).abs().max()), float((res[2][1].float() - res[4][1].float()).abs().max())
  print(line + " | equal %s maxdiff %s" % (same, dmax))
L.iic_debug_bd_ms(0)
print("per pass sum (x count): ms4 %.2f ms, ms2 %.2f ms" % (tot[4] / 1e3, tot[2] / 1e3))
