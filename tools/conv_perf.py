"""(needs the ablation build of the library: make -C iic_amd/csrc clean && make -C iic_amd/csrc ABL=1)
Per-layer timing of the conv kernels at the north-star shapes (660 images per pass).
python tools/conv_perf.py [--n 660] [--iters 20] -> table of us and TFLOP/s per geometry."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch
from iic_amd import geom, ops

LAYERS = [  # name, cin, cout, K, stride, pad, H
  ("l1 3x3 64->64 @49", 64, 64, 3, 1, 1, 49),
  ("l2.0 3x3s2 64->128 @49", 64, 128, 3, 2, 1, 49),
  ("l2.0 1x1s2 64->128 @49", 64, 128, 1, 2, 0, 49),
  ("l2 3x3 128->128 @25", 128, 128, 3, 1, 1, 25),
  ("l3.0 3x3s2 128->256 @25", 128, 256, 3, 2, 1, 25),
  ("l3.0 1x1s2 128->256 @25", 128, 256, 1, 2, 0, 25),
  ("l3 3x3 256->256 @13", 256, 256, 3, 1, 1, 13),
  ("l4.0 3x3s2 256->512 @13", 256, 512, 3, 2, 1, 13),
  ("l4.0 1x1s2 256->512 @13", 256, 512, 1, 2, 0, 13),
  ("l4 3x3 512->512 @7", 512, 512, 3, 1, 1, 7),
]
COUNT = {0: 6, 1: 1, 2: 1, 3: 7, 4: 1, 5: 1, 6: 11, 7: 1, 8: 1, 9: 5}


def timeit(fn, iters):
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


_FLUSH = []


def timeit_cold(fn, iters):
  """Per-launch time with the caches cold: a 1-GB fill runs in front of every timed call (the 256-MB MALL and the L2s hold
  nothing of the operands; in a timeit() loop the same 0.2-0.4 GB of operands are re-read from the MALL every time)."""
  if not _FLUSH:
    _FLUSH.append(torch.empty(1 << 30, dtype=torch.uint8, device="cuda"))
  fn()
  tot = 0.0
  for i in range(iters):
    _FLUSH[0].fill_(i & 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
  return tot * 1e3 / iters


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--n", type=int, default=660)
  ap.add_argument("--iters", type=int, default=20)
  ap.add_argument("--only", type=str, default="")
  ap.add_argument("--ablate", type=int, default=0)
  ap.add_argument("--bm", type=int, default=0)
  ap.add_argument("--frag", type=int, default=1, help="1: weights-direct kernel where supported (A/B column)")
  ap.add_argument("--no-wgrad", action="store_true")
  ap.add_argument("--bd-dma", type=int, default=1, help="weights-direct kernel: 1 = LDS-DMA patch loads, 0 = register-staged")
  ap.add_argument("--frag-ablate", type=str, default="", help="comma list of ablation codes for the frag kernel")
  ap.add_argument("--cold", action="store_true", help="also time every kernel with cold caches (a 1-GB fill in front of each launch)")
  ap.add_argument("--no-pw", action="store_true", help="persistent kernel off: every launch on conv_igemm_bd_kernel (the ablation codes' baseline)")
  ap.add_argument("--dense-key-ab", action="store_true", help="A/B the weights-direct kernel's swizzle key (dense pixel count vs raw index)")
  a = ap.parse_args()
  dev = torch.device("cuda:0")
  N = a.n
  if a.no_pw:
    import ctypes
    from iic_amd import _lib
    ctypes.CDLL(_lib.LIB_PATH).iic_debug_enable_pw(0)
    print("persistent kernel off")
  if not a.bd_dma:
    import ctypes
    from iic_amd import _lib
    ctypes.CDLL(_lib.LIB_PATH).iic_debug_bd_dma(0)
    print("weights-direct kernel: register-staged patch loads")
  if a.bm:
    import ctypes
    from iic_amd import _lib
    ctypes.CDLL(_lib.LIB_PATH).iic_debug_force_bm(a.bm)
    print('FORCE BM', a.bm)
  if a.ablate:
    import ctypes
    from iic_amd import _lib
    ctypes.CDLL(_lib.LIB_PATH).iic_debug_set_ablate(a.ablate)
    print('ABLATE', a.ablate)
  tot = {"fwd": 0.0, "bwd": 0.0, "wg": 0.0}
  print("%-28s %9s %8s | %9s %8s | %9s %8s | NP" % ("layer", "fwd us", "TF/s", "bwdD us", "TF/s", "wgrad us", "TF/s"))
  for li, (name, cin, cout, K, s, p, H) in enumerate(LAYERS):
    if a.only and a.only not in name:
      continue
    spec = geom.ConvSpec(cin, cout, K, s, p)
    Ho = spec.out_size(H)
    gf = geom.fwd_geom(spec, N, H, H, 1, 1)
    gb = geom.bwd_data_geoms(spec, N, H, H, 1, 1)
    x = torch.randn(N, H + 2, H + 2, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, Ho + 2, Ho + 2, cout, device=dev).to(torch.bfloat16)
    y = torch.zeros(N, Ho + 2, Ho + 2, cout, device=dev, dtype=torch.bfloat16)
    dx = torch.zeros(N, H + 2, H + 2, cin, device=dev, dtype=torch.bfloat16)
    w = torch.randn(cout, cin, K, K, device=dev) * 0.05
    wf, wb = ops.weight_prep(w)
    pw = ops.PreppedWeights(w)
    st = ops.new_stats(cout, dev)
    flops = 2.0 * N * Ho * Ho * cout * cin * K * K
    t_f = timeit(lambda: ops.conv_igemm(gf, x, wf, y, stats=st), a.iters)
    t_b = timeit(lambda: [ops.conv_igemm(g, dy, wb, dx) for g in gb], a.iters)
    t_w = 0.0 if a.no_wgrad else timeit(lambda: ops.conv_wgrad(gf, x, dy, K * K, True), a.iters)
    t_w0 = 0.0
    if not a.no_wgrad:    # A/B: register-staged kernel
      import ctypes
      from iic_amd import _lib
      L = ctypes.CDLL(_lib.LIB_PATH)
      L.iic_debug_enable_wgrad_dma(3)     # 64-pixel K-tiles, 3-4 buffers
      t_w0 = timeit(lambda: ops.conv_wgrad(gf, x, dy, K * K, True), a.iters)
      L.iic_debug_enable_wgrad_dma(1)
      L.iic_debug_wgrad_asm(0)            # ds_read_tr builtin: compiler waits for the DMA before it
      t_w1 = timeit(lambda: ops.conv_wgrad(gf, x, dy, K * K, True), a.iters)
      L.iic_debug_wgrad_asm(1)
      t_w2 = timeit(lambda: ops.conv_wgrad(gf, x, dy, K * K, True), a.iters)
    extra = "" if a.no_wgrad else " | wgrad(dma 64-px ring) %7.1f us | builtin tr reads %7.1f us, asm again %7.1f us" % (t_w0, t_w1, t_w2)
    if a.frag:
      if ops.frag_supported(gf):
        t2 = timeit(lambda: ops.conv_igemm(gf, x, pw[0], y, stats=st), a.iters)
        extra += " | frag fwd %7.1f us %7.1f TF/s" % (t2, flops / t2 / 1e6)
        if a.dense_key_ab:
          import ctypes
          from iic_amd import _lib
          L = ctypes.CDLL(_lib.LIB_PATH)
          L.iic_debug_bd_dense_key(0)
          t4 = timeit(lambda: ops.conv_igemm(gf, x, pw[0], y, stats=st), a.iters)
          L.iic_debug_bd_dense_key(1)
          t5 = timeit(lambda: ops.conv_igemm(gf, x, pw[0], y, stats=st), a.iters)
          extra += " | raw-index key %7.1f us, dense again %7.1f us" % (t4, t5)
        for code in [int(c) for c in a.frag_ablate.split(",") if c]:
          import ctypes
          from iic_amd import _lib
          L = ctypes.CDLL(_lib.LIB_PATH)
          L.iic_debug_set_ablate(code)
          # NOTE: only the frag kernel is timed under the flag (the old kernel has its own codes)
          t3 = timeit(lambda: ops.conv_igemm(gf, x, pw[0], y, stats=st), a.iters)
          L.iic_debug_set_ablate(0)
          extra += " | abl%d %6.1f" % (code, t3)
        tot["fwd2"] = tot.get("fwd2", 0.0) + COUNT[li] * (t2 - t_f)
      if all(ops.frag_supported(g) for g in gb):
        t2 = timeit(lambda: [ops.conv_igemm(g, dy, pw[1], dx) for g in gb], a.iters)
        extra += " | frag bwdD %7.1f us %7.1f TF/s" % (t2, flops / t2 / 1e6)
        tot["bwd2"] = tot.get("bwd2", 0.0) + COUNT[li] * (t2 - t_b)
    if a.cold:
      extra += " | COLD:"
      if ops.frag_supported(gf):
        extra += " frag fwd %7.1f" % timeit_cold(lambda: ops.conv_igemm(gf, x, pw[0], y, stats=st), a.iters)
      if all(ops.frag_supported(g) for g in gb):
        extra += " frag bwdD %7.1f" % timeit_cold(lambda: [ops.conv_igemm(g, dy, pw[1], dx) for g in gb], a.iters)
      if not a.no_wgrad:
        extra += " wgrad %7.1f" % timeit_cold(lambda: ops.conv_wgrad(gf, x, dy, K * K, True), a.iters)
    c = COUNT[li]
    tot["fwd"] += c * t_f; tot["bwd"] += c * t_b; tot["wg"] += c * t_w
    print("%-28s %9.1f %8.1f | %9.1f %8.1f | %9.1f %8.1f | %d%s" % (
      name, t_f, flops / t_f / 1e6, t_b, flops / t_b / 1e6, t_w, flops / max(t_w, 1e-9) / 1e6, gf.NP, extra))
  print("per pass (x count): fwd %.2f ms, bwd-data %.2f ms, wgrad(+reduce) %.2f ms; x2 passes = %.2f ms/step" % (
    tot["fwd"] / 1e3, tot["bwd"] / 1e3, tot["wg"] / 1e3, 2 * (tot["fwd"] + tot["bwd"] + tot["wg"]) / 1e3))
  if "fwd2" in tot or "bwd2" in tot:
    print("weights-direct kernel delta per step (x2 passes): fwd %+.2f ms, bwd-data %+.2f ms" % (
      2 * tot.get("fwd2", 0.0) / 1e3, 2 * tot.get("bwd2", 0.0) / 1e3))


if __name__ == "__main__":
  main()
