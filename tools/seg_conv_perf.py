"""Per-layer timing of the SegmentationNet10a convolutions (PT border 3) at the BASELINE shapes:
  python tools/seg_conv_perf.py [potsdam|coco] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iic_amd import geom, ops
from tools.conv_perf import timeit

SHAPES = {"potsdam": (75, 200), "coco": (120, 128)}


def main():
  which = sys.argv[1] if len(sys.argv) > 1 else "potsdam"
  iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
  N, S = SHAPES[which]
  P = 3
  dev = torch.device("cuda:0")
  layers = [("c2 64->128", 64, 128, 1, S), ("c3 128->256", 128, 256, 1, S // 2), ("c4 256->256", 256, 256, 1, S // 2),
            ("c5 256->512 dil2", 256, 512, 2, S // 2), ("c6 512->512 dil2", 512, 512, 2, S // 2 - 2)]
  print("%-20s %5s | %9s %7s | %9s %7s | %9s %7s | NP NP64 NP256 frag(f,b)" % ("layer", "H", "fwd us", "TF/s", "bwdD us", "TF/s", "wgrad us", "TF/s"))
  tf = tb = tw = 0.0
  for name, cin, cout, dil, H in layers:
    spec = geom.ConvSpec(cin, cout, 3, 1, 1, dil)
    Ho = spec.out_size(H)
    gf = geom.fwd_geom(spec, N, H, H, P, P)
    gb = geom.bwd_data_geoms(spec, N, H, H, P, P)
    x = torch.randn(N, H + 2 * P, H + 2 * P, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, Ho + 2 * P, Ho + 2 * P, cout, device=dev).to(torch.bfloat16)
    for t in (x, dy):
      t[:, :P] = 0; t[:, -P:] = 0; t[:, :, :P] = 0; t[:, :, -P:] = 0
    y = torch.zeros(N, Ho + 2 * P, Ho + 2 * P, cout, device=dev, dtype=torch.bfloat16)
    dx = torch.zeros(N, H + 2 * P, H + 2 * P, cin, device=dev, dtype=torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    pw = ops.PreppedWeights(w)
    st = ops.new_stats(cout, dev)
    flops = 2.0 * N * Ho * Ho * cout * cin * 9
    t_f = timeit(lambda: ops.conv_igemm(gf, x, pw[0], y, stats=st), iters)
    t_b = timeit(lambda: [ops.conv_igemm(g, dy, pw[1], dx) for g in gb], iters)
    t_w = timeit(lambda: ops.conv_wgrad(gf, x, dy, 9, True), iters)
    # backward-data with the BatchNorm-backward reduction fused into its epilogue (what the step runs for c3..c6)
    # against the plain launch + the separate reduction pass it replaces
    extra = ""
    if len(gb) == 1 and ops.red_supported(gb[0], pw[1]):
      yin = torch.randn(N, H + 2 * P, H + 2 * P, cin, device=dev).to(torch.bfloat16)
      coef = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.3, torch.zeros(cin, device=dev),
                          torch.ones(cin, device=dev), torch.zeros(cin, device=dev)])
      s1 = ops.new_stats(cin, dev)
      t_br = timeit(lambda: ops.conv_igemm(gb[0], dy, pw[1], dx, red=(yin, coef, s1, None, None)), iters)
      t_r = timeit(lambda: ops.bn_bwd_reduce(dx, None, yin, s1, N, H, H, P, cin, mask_coef=coef), iters)
      extra = " | bwdD+red %8.1f us vs plain + separate reduce %8.1f + %6.1f = %8.1f us" % (t_br, t_b, t_r, t_b + t_r)
    tf += t_f; tb += t_b; tw += t_w
    print("%-20s %5d | %9.1f %7.1f | %9.1f %7.1f | %9.1f %7.1f | %d %d %d  %s %s" % (
      name, H, t_f, flops / t_f / 1e6, t_b, flops / t_b / 1e6, t_w, flops / t_w / 1e6, gf.NP, gf.NP64, gf.NP256,
      ops.frag_supported(gf), [ops.frag_supported(g) for g in gb]) + extra)
  print("sum per pass: fwd %.2f ms, bwd-data %.2f ms, wgrad %.2f ms" % (tf / 1e3, tb / 1e3, tw / 1e3))


if __name__ == "__main__":
  main()
