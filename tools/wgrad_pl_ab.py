"""Weight-gradient DMA kernel: first-generation K loop (iic_debug_wgrad_planar 0) against the planar-patch K loop (1 =
builtin transposing reads, 2 = inline-asm reads) at the north-star layer shapes (660 images) and the 3x3 layers of the
other configs, interleaved in one process; results must be bit-identical (same MFMAs in the same order).
python tools/wgrad_pl_ab.py [N]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch  # noqa: E402

from iic_amd import _lib, geom, ops  # noqa: E402

# name, cin, cout, H, W, images, dilation, count per view, PT border, conv padding
LAYERS = [("5g l1 64->64 @49", 64, 64, 49, 49, 0, 1, 6, 1, 1), ("5g l2 128->128 @25", 128, 128, 25, 25, 0, 1, 7, 1, 1),
          ("5g l3 256->256 @13", 256, 256, 13, 13, 0, 1, 11, 1, 1), ("5g l4 512->512 @7", 512, 512, 7, 7, 0, 1, 5, 1, 1),
          ("6c 64->128 @12 n700", 64, 128, 12, 12, 700, 1, 0, 2, 1), ("6c 128->256 @6 n700", 128, 256, 6, 6, 700, 1, 0, 2, 1),
          # SegmentationNet10a as the network builds it (archs/seg.py: PT border 3, the dilated convs with padding 1)
          ("pots c2 64->128 @200", 64, 128, 200, 200, 75, 1, 0, 3, 1), ("pots c3 128->256 @100", 128, 256, 100, 100, 75, 1, 0, 3, 1),
          ("pots c4 256->256 @100", 256, 256, 100, 100, 75, 1, 0, 3, 1), ("pots c5 256->512 d2 @100", 256, 512, 100, 100, 75, 2, 0, 3, 1),
          ("pots c6 512->512 d2 @98", 512, 512, 98, 98, 75, 2, 0, 3, 1),
          ("coco c2 64->128 @128", 64, 128, 128, 128, 120, 1, 0, 3, 1), ("coco c3 128->256 @64", 128, 256, 64, 64, 120, 1, 0, 3, 1),
          ("coco c4 256->256 @64", 256, 256, 64, 64, 120, 1, 0, 3, 1), ("coco c5 256->512 d2 @64", 256, 512, 64, 64, 120, 2, 0, 3, 1),
          ("coco c6 512->512 d2 @62", 512, 512, 62, 62, 120, 2, 0, 3, 1)]

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
  N0 = int(sys.argv[1]) if len(sys.argv) > 1 else 660
  tot = {}
  only = os.environ.get("WGRAD_ONLY", "")
  variants = tuple(int(v) for v in os.environ.get("WGRAD_VARIANTS", "0,2,3,4,5").split(","))
  for name, cin, cout, H, W, n, dil, cnt, P, pad in LAYERS:
    if only and only not in name:
      continue
    N = n or N0
    spec = geom.ConvSpec(cin, cout, 3, 1, pad, dil)
    gf = geom.fwd_geom(spec, N, H, W, P, P)
    Ho, Wo = spec.out_size(H), spec.out_size(W)
    x = torch.randn(N, H + 2 * P, W + 2 * P, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, Ho + 2 * P, Wo + 2 * P, cout, device=dev).to(torch.bfloat16)
    for t_ in (x, dy):
      t_[:, :P] = 0; t_[:, -P:] = 0; t_[:, :, :P] = 0; t_[:, :, -P:] = 0
    flops = 2.0 * N * Ho * Wo * cout * cin * 9
    t, out = {}, {}
    for rep in range(2):
      for v in variants:
        L.iic_debug_enable_wgrad_dma(0 if v < 0 else 1)      # -1: the register-staged kernel (conv_wgrad.hip)
        L.iic_debug_wgrad_planar(max(v, 0))
        tt = timeit(lambda: ops.conv_wgrad(gf, x, dy, 9, True))
        t[v] = min(t.get(v, 1e9), tt)
        out[v] = ops.conv_wgrad(gf, x, dy, 9, True).clone()
    torch.cuda.synchronize()
    same = all(bool(torch.equal(out[variants[0]], out[v])) for v in variants)
    names = {-1: "register-staged", 0: "gen-1", 1: "planar", 2: "planar asm", 3: "pipelined", 4: "pipelined swp", 5: "default"}
    v0 = variants[0]
    print("%-26s " % name + " | ".join("%s %7.1f us %5.0f TF/s (%.3fx)" % (names[v], t[v], flops / t[v] / 1e6, t[v0] / t[v])
                                        for v in variants) + "  bit-identical %s" % same, flush=True)
    for v in t:
      tot[v] = tot.get(v, 0.0) + cnt * t[v]
    if os.environ.get("WGRAD_ABL"):      # timing ablations of the planar kernel (results wrong by design)
      L.iic_debug_wgrad_planar(int(os.environ.get("WGRAD_ABL_VARIANT", "5")))
      row = []
      for code in [int(c) for c in os.environ["WGRAD_ABL"].split(",")]:
        L.iic_debug_wgrad_ablate(code)
        row.append("abl%d %.1f" % (code, timeit(lambda: ops.conv_wgrad(gf, x, dy, 9, True))))
      L.iic_debug_wgrad_ablate(0)
      print("    planar asm, ablations (us incl. reduce): " + ", ".join(row), flush=True)
  print("ClusterNet5g per view (x layer counts, incl. the reduce pass): " + ", ".join("variant %d %.2f ms" % (v, tot[v] / 1e3) for v in sorted(tot)))
  L.iic_debug_wgrad_planar(5)
  L.iic_debug_enable_wgrad_dma(1)


if __name__ == "__main__":
  main()
