"""(needs the ablation build of the library: make -C iic_amd/csrc clean && make -C iic_amd/csrc ABL=1)
Per-tile phase timeline of conv_igemm_bd_kernel (the instruction-level substitute for a rocprofv3 ATT
thread trace: the ATT decoder library is not part of this image).  The kernel's PROF build
(iic_debug_set_ablate(128), results unchanged) lets wave 0 of every workgroup stamp s_memtime at its
phase boundaries; this tool launches one layer, decodes the stamps and prints where a tile's cycles go,
how the workgroups of a launch are packed onto the CUs, and what the MFMA loop would need alone.

python tools/bd_timeline.py [--layer l2|l3|l4|all] [--stagger CYCLES_PER_TAP] [--red]
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

LAYERS = {
  "l2": ("layer2 3x3 128->128 @25", 128, 128, 25),
  "l3": ("layer3 3x3 256->256 @13", 256, 256, 13),
  "l4": ("layer4 3x3 512->512 @7", 512, 512, 7),
}


def run_layer(L, key, N, stagger, red, iters):
  name, cin, cout, H = LAYERS[key]
  dev = torch.device("cuda:0")
  spec = geom.ConvSpec(cin, cout, 3, 1, 1)
  g = geom.fwd_geom(spec, N, H, H, 1, 1)
  x = torch.randn(N, H + 2, H + 2, cin, device=dev).to(torch.bfloat16)
  x[:, 0], x[:, -1], x[:, :, 0], x[:, :, -1] = 0, 0, 0, 0
  y = torch.zeros(N, H + 2, H + 2, cout, device=dev, dtype=torch.bfloat16)
  w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
  pw = ops.PreppedWeights(w)
  st = ops.new_stats(cout, dev)
  kw = dict(stats=st)
  if red:
    ry = torch.randn(N, H + 2, H + 2, cout, device=dev).to(torch.bfloat16)
    sums = ops.new_stats(cout, dev)
    kw = dict(red=(ry, None, sums, None, None))
  M = geom.gemm_rows(g)
  tiles = (M + 255) // 256 * (cout // 128)
  slots = L.iic_debug_bd_prof_slots()
  buf = torch.zeros(tiles * slots, device=dev, dtype=torch.int64)
  L.iic_debug_bd_stagger(stagger)

  def launch():
    ops.conv_igemm(g, x, pw[0], y, **kw)

  def timed():
    for _ in range(3):
      launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
      launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

  L.iic_debug_set_ablate(0)
  us_plain = timed()
  L.iic_debug_bd_prof(ctypes.c_void_p(buf.data_ptr()))
  L.iic_debug_set_ablate(128)
  us_prof = timed()
  buf.zero_()
  launch()
  torch.cuda.synchronize()
  L.iic_debug_set_ablate(0)
  L.iic_debug_bd_prof(None)
  L.iic_debug_bd_stagger(0)
  r = buf.view(tiles, slots).cpu().numpy().astype(np.int64)
  t0, t1, t2, t3, t4, t5, bsum, nb, hw, tix, rt = [r[:, i] for i in range(11)]
  base = t0.min()      # (s_memtime is NOT synchronised between XCDs: only per-CU differences below are meaningful)
  flops = 2.0 * N * H * H * cout * cin * 9
  nit = (cin // 64) * 9
  mfma_cyc = nit * 4 * 8 * 32            # one wave's MFMA issue cycles per tile
  span = t5.max() - base
  rts = (rt.max() - rt.min()) / 100.0    # us between first and last end stamp (100 MHz)
  ghz = span / max(us_prof, 1e-9) / 1e3
  print("== %s  N=%d  stagger=%d cyc/tap  %s" % (name, N, stagger, "fused reduction" if red else "fwd + stats"))
  print("   launch %.1f us (%.0f TF/s); PROF build %.1f us; %d tiles; kernel span %d cycles => ~%.2f GHz "
        "(s_memrealtime span of end stamps %.1f us)" % (us_plain, flops / us_plain / 1e6, us_prof, tiles, span, ghz, rts))
  ph = [("setup (row table, keys)", t1 - t0), ("prologue loads (B ring + patch DMA + barrier)", t2 - t1),
        ("K loop incl. boundaries", t3 - t2), ("  of which chunk boundaries (%d per tile)" % int(nb.max()), bsum),
        ("epilogue A: statistics + accumulators -> LDS", t4 - t3), ("epilogue B: tile store (+ fused reads)", t5 - t4),
        ("whole tile", t5 - t0)]
  for nm, v in ph:
    print("   %-52s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f cycles" % (
      nm, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
  loop = (t3 - t2 - bsum).astype(np.float64)
  print("   K loop without boundaries: mean %.0f cycles; one wave's MFMAs alone need %d, two co-resident waves %d "
        "=> MFMA pipe share while in the loop %.2f" % (loop.mean(), mfma_cyc, 2 * mfma_cyc, 2 * mfma_cyc / loop.mean()))
  tile = (t5 - t0).astype(np.float64)
  print("   tile = %.0f cycles; MFMA-only floor for a co-resident pair = %d  => per-tile efficiency %.2f" % (
    tile.mean(), 2 * mfma_cyc, 2 * mfma_cyc / tile.mean()))
  # packing: CU identity from HW_ID (cu 11:8, sh 12, se 15:13) + XCC_ID
  hwid = hw & 0xffffffff
  xcc = (hw >> 32) & 0xf
  cu = ((hwid >> 8) & 0xf) | (((hwid >> 12) & 0xf) << 4) | (xcc << 8)
  tg = (hwid >> 16) & 0xf
  ucu = np.unique(cu)
  per_cu = np.array([(cu == c).sum() for c in ucu])
  busy = np.array([(t5[cu == c] - t0[cu == c]).sum() for c in ucu], dtype=np.float64)
  print("   CUs seen %d; tiles per CU min %d / mean %.2f / max %d; threadgroup slots used %s" % (
    len(ucu), per_cu.min(), per_cu.mean(), per_cu.max(), sorted(set(int(v) for v in tg))))
  print("   slot occupancy: sum of tile cycles / (2 slots x CUs x span) = %.2f   (quantisation + ramp + tail)" % (
    busy.sum() / (2.0 * len(ucu) * span)))
  st_rel = np.sort(t0 - base)
  print("   start stamps: first round (%d tiles) all started by %d cycles; last tile starts at %d, ends at %d" % (
    min(tiles, 2 * len(ucu)), st_rel[min(tiles, 2 * len(ucu)) - 1], st_rel[-1], span))
  # per CU (its own s_memtime is consistent): how much of the CU's busy span has 2 / 1 / 0 workgroups in
  # their K loop?  0 = the matrix pipes idle behind prologues, chunk reloads' neighbours and epilogues
  c2 = c1 = c0 = tot = 0.0
  for c in ucu:
    m = cu == c
    ev = []
    for a_, b_ in zip(t2[m], t3[m]):
      ev.append((int(a_), 1))
      ev.append((int(b_), -1))
    ev.sort()
    lo, hi = int(t0[m].min()), int(t5[m].max())
    cur, last = 0, lo
    for t_, d_ in ev:
      dt = t_ - last
      if cur >= 2: c2 += dt
      elif cur == 1: c1 += dt
      else: c0 += dt
      cur += d_
      last = t_
    c0 += hi - last
    tot += hi - lo
  print("   per-CU busy span: both workgroups in the K loop %.2f, one %.2f, none %.2f of the time; mean CU span %.0f cycles" % (
    c2 / tot, c1 / tot, c0 / tot, tot / len(ucu)))
  print("   MFMA floor for the CU's tiles / CU span = %.2f" % (per_cu.mean() * mfma_cyc / (tot / len(ucu))))
  # overlap of the two slots of a CU: how far apart do co-resident tiles start?
  d = []
  for c in ucu[:64]:
    s = np.sort(t0[cu == c])
    if len(s) >= 2:
      d.append(s[1] - s[0])
  if d:
    print("   start offset between the first two tiles of a CU: median %d cycles (tile/2 = %d would interleave them)" % (
      int(np.median(d)), int(tile.mean() / 2)))
  return us_plain


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--layer", default="all")
  ap.add_argument("--n", type=int, default=660)
  ap.add_argument("--iters", type=int, default=10)
  ap.add_argument("--stagger", type=str, default="0")
  ap.add_argument("--red", action="store_true")
  a = ap.parse_args()
  L = ctypes.CDLL(_lib.LIB_PATH)
  L.iic_debug_bd_prof.argtypes = [ctypes.c_void_p]
  _lib.lib()
  for key in (LAYERS if a.layer == "all" else [a.layer]):
    for stg in [int(v) for v in a.stagger.split(",")]:
      run_layer(L, key, a.n, stg, False, a.iters)
    if a.red:
      run_layer(L, key, a.n, 0, True, a.iters)


if __name__ == "__main__":
  main()
