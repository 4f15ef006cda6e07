"""A/B of the persistent weights-direct conv kernel (csrc/conv_igemm_pw.hip) against conv_igemm_bd_kernel at the
north-star shapes: outputs bit-identical?, statistics equal to rounding?, us / TF/s per launch class (forward + BN
statistics, backward-data, backward-data with the fused BatchNorm-backward reduction), the same with ONE workgroup
per CU (a wave alone on its SIMD), and the per-phase cycle sums of the PROF build.
  python tools/pw_perf.py [--n 660] [--iters 20] [--only l3]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import numpy as np
import torch

from iic_amd import _lib, geom, ops

LAYERS = [  # name, cin, cout, H, launches per pass (fwd, bwd plain, bwd fused-red)
  ("l2 3x3 128->128 @25", 128, 128, 25),
  ("l3 3x3 256->256 @13", 256, 256, 13),
  ("l4 3x3 512->512 @7", 512, 512, 7),
]


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


def decode_stats(st, C):
  """exact accumulators [stripe][C][2][8] int64 -> float64 [C][2] (csrc/common.h)."""
  a = st.view(torch.int64).view(-1, C, 2, 8).cpu().numpy().astype(np.float64).sum(0)
  v = np.zeros((C, 2))
  for b in range(7):
    v += a[:, :, b] * 2.0 ** (-96 + 24 * b)
  return v


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--n", type=int, default=660)
  ap.add_argument("--iters", type=int, default=20)
  ap.add_argument("--only", type=str, default="")
  ap.add_argument("--no-prof", action="store_true")
  ap.add_argument("--prof-stagger", type=int, default=0, help="start offset used in the PROF run")
  ap.add_argument("--dbg", type=int, default=0, help="iic_debug_pw_dbg for the PROF run and an extra timing column")
  ap.add_argument("--stagger", type=str, default="", help="comma list of start offsets (cycles) of the odd-slot workgroups to time")
  a = ap.parse_args()
  dev = torch.device("cuda:0")
  L = _lib.lib()
  N = a.n
  print("%-22s %-10s | %9s %8s | %9s %8s | %s" % ("layer", "class", "bd us", "TF/s", "pw us", "TF/s", "pw alone (1 WG/CU) us / bd alone us"))
  for name, cin, cout, H in LAYERS:
    if a.only and a.only not in name:
      continue
    spec = geom.ConvSpec(cin, cout, 3, 1, 1)
    gf = geom.fwd_geom(spec, N, H, H, 1, 1)
    gb = geom.bwd_data_geoms(spec, N, H, H, 1, 1)
    assert len(gb) == 1
    gb = gb[0]
    torch.manual_seed(0)
    x = torch.zeros(N, H + 2, H + 2, cin, device=dev, dtype=torch.bfloat16)
    x[:, 1:-1, 1:-1] = torch.randn(N, H, H, cin, device=dev).to(torch.bfloat16)
    dy = torch.zeros(N, H + 2, H + 2, cout, device=dev, dtype=torch.bfloat16)
    dy[:, 1:-1, 1:-1] = torch.randn(N, H, H, cout, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    pw = ops.PreppedWeights(w)
    # fused-reduction inputs (a conv1 backward-data launch of a BasicBlock): residual gradient, mask activation, y, coef
    rg = torch.zeros_like(x); rg[:, 1:-1, 1:-1] = torch.randn(N, H, H, cin, device=dev).to(torch.bfloat16)
    ra = torch.zeros_like(x); ra[:, 1:-1, 1:-1] = torch.randn(N, H, H, cin, device=dev).to(torch.bfloat16)
    yy = torch.zeros_like(x); yy[:, 1:-1, 1:-1] = torch.randn(N, H, H, cin, device=dev).to(torch.bfloat16)
    coef = torch.randn(5, cin, device=dev)
    flops = 2.0 * N * H * H * cout * cin * 9

    def run(cls, out, st, sums):
      if cls == "fwd+stats":
        ops.conv_igemm(gf, x, pw[0], out, stats=st)
      elif cls == "bwd":
        ops.conv_igemm(gb, dy, pw[1], out)
      else:
        ops.conv_igemm(gb, dy, pw[1], out, res_grad=rg, res_act=ra, premask=True, red=(yy, coef, sums, None, None))

    for cls in ("fwd+stats", "bwd", "bwd+red"):
      outs, stats_v, sums_v = {}, {}, {}
      for use_pw in (0, 1):
        L.iic_debug_enable_pw(use_pw)
        out = torch.zeros(N, H + 2, H + 2, cout if cls == "fwd+stats" else cin, device=dev, dtype=torch.bfloat16)
        st = ops.new_stats(cout, dev)
        sums = ops.new_stats(cin, dev)
        run(cls, out, st, sums)
        torch.cuda.synchronize()
        outs[use_pw] = out
        stats_v[use_pw] = decode_stats(st, cout)
        sums_v[use_pw] = decode_stats(sums, cin)
      same = torch.equal(outs[0], outs[1])
      nz = float((outs[1].float().abs() > 0).float().mean())
      dstat = np.abs(stats_v[0] - stats_v[1]).max() / max(np.abs(stats_v[0]).max(), 1e-30) if cls == "fwd+stats" else 0.0
      dsum = np.abs(sums_v[0] - sums_v[1]).max() / max(np.abs(sums_v[0]).max(), 1e-30) if cls == "bwd+red" else 0.0
      t = {}
      scratch = torch.zeros_like(outs[0])
      st = ops.new_stats(cout, dev)
      sums = ops.new_stats(cin, dev)
      for rep in range(2):                       # interleaved A/B in one process
        for use_pw in (0, 1):
          L.iic_debug_enable_pw(use_pw)
          t.setdefault(use_pw, []).append(timeit(lambda: run(cls, scratch, st, sums), a.iters))
      L.iic_debug_enable_pw(1)
      L.iic_debug_pw_one_wg(1)
      t_alone = timeit(lambda: run(cls, scratch, st, sums), a.iters)
      L.iic_debug_pw_one_wg(0)
      L.iic_debug_enable_pw(0)
      tb, tp = min(t[0]), min(t[1])
      print("%-22s %-10s | %9.1f %8.1f | %9.1f %8.1f | %7.1f | out identical %s (nonzero %.2f) stats rel %.1e sums rel %.1e | runs bd %s pw %s" % (
        name, cls, tb, flops / tb / 1e6, tp, flops / tp / 1e6, t_alone, same, nz, dstat, dsum,
        ["%.1f" % v for v in t[0]], ["%.1f" % v for v in t[1]]))
      L.iic_debug_enable_pw(1)
      if a.stagger:
        row = []
        for sg in [int(v) for v in a.stagger.split(",")]:
          L.iic_debug_pw_stagger(sg)
          row.append("%d: %.1f" % (sg, timeit(lambda: run(cls, scratch, st, sums), a.iters)))
        L.iic_debug_pw_stagger(0)
        print("    stagger (cycles: us)  " + "   ".join(row))
      if not a.no_prof and cls != "bwd":
        slots = L.iic_debug_pw_prof_slots()
        grid = L.iic_debug_pw_grid(ctypes.byref(gf if cls == "fwd+stats" else gb))
        buf = torch.zeros(grid * slots, device=dev, dtype=torch.int64)
        L.iic_debug_pw_prof(ctypes.c_void_p(buf.data_ptr()))
        L.iic_debug_pw_stagger(a.prof_stagger)
        L.iic_debug_pw_dbg(a.dbg)
        run(cls, scratch, st, sums)
        if a.dbg:
          print("    dbg %d: %.1f us (2 WG/CU)" % (a.dbg, timeit(lambda: run(cls, scratch, st, sums), a.iters)))
          L.iic_debug_pw_one_wg(1)
          print("    dbg %d: %.1f us (1 WG/CU)" % (a.dbg, timeit(lambda: run(cls, scratch, st, sums), a.iters)))
          L.iic_debug_pw_one_wg(0)
        L.iic_debug_pw_dbg(0)
        L.iic_debug_pw_stagger(0)
        torch.cuda.synchronize()
        L.iic_debug_pw_prof(None)
        p = buf.view(grid, slots).cpu().numpy().astype(np.float64)
        tiles = p[:, 5].sum()
        hw = p[:, 7].astype(np.int64)
        cu_key = ((hw >> 32) & 0xf) * 100000 + ((hw >> 13) & 7) * 1000 + ((hw >> 12) & 1) * 100 + ((hw >> 8) & 0xf)   # XCC, SE, SH, CU
        tg = (hw >> 16) & 0xf
        per_cu = {}
        for k, t in zip(cu_key.tolist(), tg.tolist()):
          per_cu.setdefault(k, []).append(t)
        pairs = [tuple(sorted(v)) for v in per_cu.values()]
        xcd_ok = float(np.mean(((hw >> 32) & 0xf) == (np.arange(grid) & 7)))
        print("    placement: %d CUs seen, workgroups per CU %s, TG_ID sets %s, block b on XCD b%%8: %.2f" % (
          len(per_cu), sorted(set(len(v) for v in per_cu.values())), sorted(set(pairs))[:6], xcd_ok))
        nit = (cin // 64) * 9
        mfma_pair = nit * 4 * 8 * 32 * 2
        # timeline of the two workgroups of three CUs (cycles from the earlier one's start)
        shown = 0
        for k, v in per_cu.items():
          idx = [i for i in range(grid) if cu_key[i] == k]
          if len(idx) != 2 or shown >= 3:
            continue
          shown += 1
          base = min(p[i, 8] for i in idx)
          for i in idx:
            segs = []
            for t in range(4):
              if p[i, 8 + 4 * t + 3] > 0:
                segs.append("top %6d K %6d..%6d E..%6d" % tuple(int(p[i, 8 + 4 * t + j] - base) for j in range(4)))
            print("      CU %d block %3d TG %d: %s" % (k, i, tg[i], " | ".join(segs)))
        live = p[:, 5] > 0
        ghz = (p[live, 4] / np.maximum(p[live, 25] - p[live, 24], 1.0)) * 0.1
        span_us = (p[live, 25].max() - p[live, 24].min()) / 100.0
        t_prof = timeit(lambda: run(cls, scratch, st, sums), a.iters) if False else 0.0
        print("    clock: %.2f GHz mean (min %.2f max %.2f); first start -> last end %.1f us; workgroup lifetimes %.1f..%.1f us" % (
          ghz.mean(), ghz.min(), ghz.max(), span_us, ((p[live, 25] - p[live, 24]) / 100.0).min(), ((p[live, 25] - p[live, 24]) / 100.0).max()))
        print("    PROF %d workgroups, %d tiles: per tile  wait-patch %6.0f | K loop %7.0f (boundaries %6.0f) | epilogue %6.0f | "
              "all %7.0f cycles; MFMA floor of a co-resident pair %d => %.2f" % (
                grid, tiles, p[:, 0].sum() / tiles, p[:, 1].sum() / tiles, p[:, 2].sum() / tiles, p[:, 3].sum() / tiles,
                p[:, 4].sum() / tiles, mfma_pair, mfma_pair / (p[:, 4].sum() / tiles)))


if __name__ == "__main__":
  main()
