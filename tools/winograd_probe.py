"""Go / no-go measurement for Winograd F(2x2, 3x3) (kernel: iic_amd/csrc/probes/wino_probe.hip).

  python tools/winograd_probe.py                 # parity + A/B timing at the north-star shapes (660 images)
  python tools/winograd_probe.py --dbg           # also decode workgroup 0's checkpoints (LDS image, accumulators, Z)
  python tools/winograd_probe.py --hold l3,wino,6    # run one kernel back to back for 6 s (tools/power_probe_wino.sh samples rocm-smi)

Replaces nothing in the product: the A/B partner is the product's own launch (ops.conv_igemm -> conv_igemm_bd / _pw).
Reference: float64 F.conv2d on the SAME bf16 operands (weights rounded to bf16 for the direct kernel's operand); the
Winograd kernel rounds G g G^T instead, so its error is also quoted against the unrounded-weight convolution.
"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from iic_amd import _lib, geom, ops

LAYERS = {  # name: (C, H, launches of this shape per forward pass)
  "l2": (128, 25, 7),
  "l3": (256, 13, 11),
  "l4": (512, 7, 5),
}
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
VPITCH, VBYTES = 1296, 32 * 1296


def probe_lib():
  L = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libiic_probe.so"))   # make -C iic_amd/csrc probes
  L.iic_probe_wino_fwd.restype = ctypes.c_int
  L.iic_probe_wino_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
  L.iic_probe_wino_weight_prep.restype = ctypes.c_int
  L.iic_probe_wino_weight_prep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
  L.iic_probe_wino_lds_bytes.restype = ctypes.c_long
  L.iic_probe_wino_lds_bytes.argtypes = [ctypes.c_int] * 5
  return L


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


def make_case(C, H, N, dev, seed=0):
  g = torch.Generator(device="cpu").manual_seed(seed)
  x = torch.zeros(N, H + 2, H + 2, C)
  x[:, 1:-1, 1:-1, :] = torch.randn(N, H, H, C, generator=g).relu()      # post-ReLU-like activations
  w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
  return x.to(torch.bfloat16).to(dev), w.to(dev)


def ufrag_torch(w):
  """G g G^T (fp32) -> bf16, [pos][ks][Cout/32][lane][8] (the prep kernel's contract)."""
  Co, Ci = w.shape[:2]
  Gf = G.float().to(w.device)
  U = torch.einsum("ar,oirs,bs->oiab", Gf, w.float(), Gf)            # [Co][Ci][4][4]
  # same association as the kernel: rows first (t = G g), then columns
  t = torch.einsum("ar,oirs->oias", Gf, w.float())
  U = torch.einsum("oias,bs->oiab", t, Gf)
  U = U.reshape(Co, Ci, 16).permute(2, 1, 0)                           # [pos][k = Ci][n = Co]
  U = U.reshape(16, Ci // 16, 2, 8, Co // 32, 32)                      # pos, ks, g5, e, nb, l31
  U = U.permute(0, 1, 4, 2, 5, 3).contiguous()                         # pos, ks, nb, g5, l31, e
  return U.to(torch.bfloat16).reshape(-1)


def run_layer(L, name, N, iters, dev, dbg):
  C, H, cnt = LAYERS[name]
  x, w = make_case(C, H, N, dev)
  lds = L.iic_probe_wino_lds_bytes(N, H, H, C, C)
  uf = torch.empty(16 * C * C, dtype=torch.bfloat16, device=dev)
  _lib.check(L.iic_probe_wino_weight_prep(w.data_ptr(), uf.data_ptr(), C, C, 0, _lib.stream_ptr()), "wino weight prep")
  torch.cuda.synchronize()
  uf_t = ufrag_torch(w)
  prep_diff = (uf.float() - uf_t.float()).abs().max().item()
  prep_bits = (uf.view(torch.int16) != uf_t.view(torch.int16)).float().mean().item()
  yw = torch.zeros(N, H + 2, H + 2, C, dtype=torch.bfloat16, device=dev)
  yd = torch.zeros_like(yw)
  st_w, st_d = ops.new_stats(C, dev), ops.new_stats(C, dev)

  def wino(stats=st_w, abl=0, dbgbuf=None):
    _lib.check(L.iic_probe_wino_fwd(x.data_ptr(), uf.data_ptr(), yw.data_ptr(), None if stats is None else stats.data_ptr(),
                                    N, H, H, C, C, None if dbgbuf is None else dbgbuf.data_ptr(), abl, _lib.stream_ptr()),
               "wino fwd")
  spec = geom.ConvSpec(C, C, 3, 1, 1)
  gf = geom.fwd_geom(spec, N, H, H, 1, 1)
  pw = ops.PreppedWeights(w)

  def direct(stats=st_d):
    ops.conv_igemm(gf, x, pw[0], yd, stats=stats)

  wino()
  direct()
  torch.cuda.synchronize()
  # ---- parity: float64 convolution of a sample of images on the same bf16 operands ----
  idx = sorted(set([0, 1, 2, 3, N // 2, N // 2 + 1, N - 2, N - 1]))
  xs = x[idx][:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).double().cpu()
  ref_b = F.conv2d(xs, w.to(torch.bfloat16).double().cpu(), padding=1)      # bf16-rounded weights (direct kernel's operand)
  ref_f = F.conv2d(xs, w.double().cpu(), padding=1)                         # master weights
  ow = yw[idx][:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).double().cpu()
  od = yd[idx][:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).double().cpu()
  mx = ref_b.abs().max().item()
  e = lambda a, b: ((a - b).abs().max().item(), (a - b).pow(2).mean().sqrt().item())
  ew_b, ed_b, ew_f, ed_f = e(ow, ref_b), e(od, ref_b), e(ow, ref_f), e(od, ref_f)
  border_clean = bool((yw[:, 0].abs().max() == 0) & (yw[:, -1].abs().max() == 0) & (yw[:, :, 0].abs().max() == 0) & (yw[:, :, -1].abs().max() == 0))
  whole = (yw.float() - yd.float()).abs().max().item()
  sw, sd = ops.stats_decode(st_w, C), ops.stats_decode(st_d, C)
  cnt_px = float(N * H * H)
  mean_w, mean_d = sw[0] / cnt_px, sd[0] / cnt_px
  var_w, var_d = sw[1] / cnt_px - mean_w ** 2, sd[1] / cnt_px - mean_d ** 2
  # BatchNorm statistics as their consumer sees them: mean shift in units of the channel's std, variance ratio
  st_rel = max(((mean_w - mean_d).abs() / var_d.sqrt()).max().item(), (var_w / var_d - 1).abs().max().item())
  # each kernel's statistics against the sums of its own stored (bf16-rounded) outputs
  def self_check(y, st):
    yi = y[:, 1:-1, 1:-1, :].double()
    s1, s2 = yi.sum((0, 1, 2)), (yi * yi).sum((0, 1, 2))
    return max(((st[0] - s1).abs() / (s2.sqrt() + 1e-9)).max().item(), ((st[1] - s2).abs() / s2).max().item())
  self_w, self_d = self_check(yw, sw), self_check(yd, sd)
  import math
  print("%s %d->%d @%d x %d images: LDS %d B | weight prep vs torch: max diff %.2e, differing bf16 %.4f%%" % (name, C, C, H, N, lds, prep_diff, 100 * prep_bits))
  print("  max|y| %.3f | vs fp64 conv (bf16 weights): winograd max %.3e = 2^%.2f max|y| rms %.3e | direct max %.3e rms %.3e" % (
    mx, ew_b[0], math.log2(max(ew_b[0], 1e-30) / mx), ew_b[1], ed_b[0], ed_b[1]))
  print("  vs fp64 conv (fp32 master weights): winograd max %.3e rms %.3e | direct max %.3e rms %.3e" % (ew_f[0], ew_f[1], ed_f[0], ed_f[1]))
  print("  winograd vs direct over all %d images: max |diff| %.3e | border untouched %s | BatchNorm mean (in std) / variance (ratio) diff %.2e | statistics vs own stored outputs: winograd %.2e direct %.2e" % (N, whole, border_clean, st_rel, self_w, self_d))
  ok = ew_b[0] <= 2.0 ** -6 * mx and border_clean and st_rel < 1e-2 and self_w < max(1e-2, 1.5 * self_d)
  print("  PARITY %s (gate: max error <= 2^-6 max|y| = %.3e; statistics no further from the stored outputs than the direct kernel's)" % ("ok" if ok else "FAILED", 2.0 ** -6 * mx))
  if dbg or not ok:
    decode_checkpoints(L, name, x, w, uf, N, H, C, dev, wino)
  # ---- timing, interleaved twice ----
  flops = 2.0 * N * H * H * C * C * 9
  tw, td = [], []
  for _ in range(2):
    td.append(timeit(direct, iters))
    tw.append(timeit(wino, iters))
  abls = {}
  for a in (1, 2, 3, 4, 7):
    abls[a] = timeit(lambda: wino(abl=a), iters)
  t_w, t_d = min(tw), min(td)
  phase_profile(L, name, N, H, C, dev, wino)
  print("  TIME direct %.1f us (%.0f TF/s)  winograd %.1f us (%.0f TF/s direct-equivalent)  speed-up %.2fx  | runs direct %s wino %s" % (
    t_d, flops / t_d / 1e6, t_w, flops / t_w / 1e6, t_d / t_w, ["%.1f" % v for v in td], ["%.1f" % v for v in tw]))
  print("  ablations (us): no input transform %.1f | no main MFMAs %.1f | neither %.1f | no stage-2 epilogue %.1f | loads + barriers only %.1f" % (
    abls[1], abls[2], abls[3], abls[4], abls[7]))
  return {"name": name, "t_direct": t_d, "t_wino": t_w, "count": cnt, "ok": ok}


def phase_profile(L, name, N, H, C, dev, wino):
  """s_memtime sums of wave 0 of every workgroup (kernel template PROF): where a workgroup's cycles go."""
  TH = (H + 1) // 2
  grid = ((N * TH * TH + 63) // 64) * (C // 64)
  buf = torch.zeros(grid * 16, dtype=torch.int64, device=dev)
  wino(abl=8, dbgbuf=buf)
  torch.cuda.synchronize()
  q = buf.cpu().reshape(grid, 16).double()
  ks = q[:, 9].mean().item()
  m = q.mean(0)
  rt = q[:, 10]
  span_us = (rt.max() - rt.min()).item() / 100.0          # s_memrealtime: 100 MHz
  names = ["set-up", "prologue", "top wait+barrier", "issue+A latency", "main MFMAs", "transform", "stage 1", "stage 2"]
  per_step = {2: m[2].item() / (ks + 1), 3: m[3].item() / ks, 4: m[4].item() / ks, 5: m[5].item() / ks}
  print("  PROF (cycles per workgroup, mean of %d; %d k-steps): %s | whole %.0f" % (
    grid, int(ks), ", ".join("%s %.0f" % (names[i], m[i].item()) for i in range(8)), m[8].item()))
  print("       per k-step: top wait+barrier %.0f, issue+A latency %.0f, main MFMAs %.0f (floor 512), transform %.0f (floor 512) | last-end minus first-end %.1f us" % (
    per_step[2], per_step[3], per_step[4], per_step[5], span_us))


def decode_checkpoints(L, name, x, w, uf, N, H, C, dev, wino):
  """Workgroup 0 (tiles 0..63, couts 0..63): compare the LDS image / accumulators / Z with torch."""
  dbgbuf = torch.zeros(3 * 512 * 1024, dtype=torch.uint8, device=dev)
  st = ops.new_stats(C, dev)
  wino(stats=st, dbgbuf=dbgbuf)
  torch.cuda.synchronize()
  raw = dbgbuf.cpu()
  Hp = Wp = H + 2
  TH = TW = (H + 1) // 2
  lds = L.iic_probe_wino_lds_bytes(N, H, H, C, C)
  # recover rawb from the LDS size: lds = tab_off + 768 + 2048 + npmax16;  data = 2 rawb + 2 VBYTES
  # (tab_off = max(data, 128 KB)) -- solve by trying both
  rawb = None
  for cand in range(1024, 41 * 1024, 1024):
    data = 2 * cand + 2 * VBYTES
    tab = max(data, 4 * 2 * 64 * 64 * 4)
    if tab + 768 + 2048 + ((cand // 64 + 15) & ~15) == lds:
      rawb = cand
  assert rawb is not None, "cannot recover the raw buffer size from %d" % lds
  data = 2 * rawb + 2 * VBYTES
  tab = max(data, 4 * 2 * 64 * 64 * 4)
  tpin = raw[tab:tab + 256].view(torch.int32)
  tout = raw[tab + 256:tab + 512].view(torch.int32)
  tflag = raw[tab + 512:tab + 768].view(torch.int32)
  exp_pin = []
  for m in range(64):
    n, rem = divmod(m, TH * TW)
    ti, tj = divmod(rem, TW)
    exp_pin.append((n * Hp + 2 * ti) * Wp + 2 * tj)
  print("  [dbg] tile tables: pin ok %s (first %s)" % (tpin.tolist() == exp_pin, tpin[:4].tolist()))
  p_lo = int(tpin[0])
  npix = int(tpin[63]) + 3 * Wp + 3 - p_lo + 1
  ypar = raw[tab + 768 + 2048:tab + 768 + 2048 + npix]
  exp_par = torch.tensor([(((p_lo + r) // Wp) % Hp) & 1 for r in range(npix)], dtype=torch.uint8)
  print("  [dbg] row-parity table ok %s" % bool((ypar == exp_par).all()))
  # raw[0]: pixel r at r*64, physical slot ps holds logical slot ps ^ (2*par) of channels 0..31
  xr = x.reshape(-1, C)[p_lo:p_lo + npix, :32].cpu().view(torch.int16).reshape(npix, 4, 8)
  got = raw[:npix * 64].view(torch.int16).reshape(npix, 4, 8)
  exp = torch.empty_like(xr)
  for ps in range(4):
    ls = ps ^ (2 * exp_par.long())
    exp[:, ps, :] = xr[torch.arange(npix), ls, :]
  bad = (got != exp).any(-1).any(-1)
  print("  [dbg] raw patch (chunk 0) in LDS: %d of %d pixels differ%s" % (int(bad.sum()), npix, "" if not bad.any() else " first bad %d" % int(bad.nonzero()[0])))
  # V[0]: k-step 0 (channels 0..15): V[pos][khalf][slot = tile + pos][16 B = 8 channels]
  vb = raw[2 * rawb:2 * rawb + VBYTES]
  xt = x.float().cpu()
  tiles = torch.zeros(64, 4, 4, 16)
  for m in range(64):
    n, rem = divmod(m, TH * TW)
    ti, tj = divmod(rem, TW)
    patch = torch.zeros(4, 4, 16)
    ys, xs_ = min(4, Hp - 2 * ti), min(4, Wp - 2 * tj)
    patch[:ys, :xs_] = xt[n, 2 * ti:2 * ti + ys, 2 * tj:2 * tj + xs_, :16]
    tiles[m] = patch
  Vexp = torch.einsum("ar,mrsc,bs->mabc", BT.float(), tiles, BT.float()).to(torch.bfloat16)      # [tile][xi][nu][c]
  Vgot = torch.zeros(64, 4, 4, 16, dtype=torch.bfloat16)
  for pos in range(16):
    for kh in range(2):
      row = vb[(pos * 2 + kh) * VPITCH:(pos * 2 + kh + 1) * VPITCH].view(torch.bfloat16).reshape(-1, 8)
      Vgot[:, pos // 4, pos % 4, kh * 8:(kh + 1) * 8] = row[pos:pos + 64]
  # positions whose patch runs off the image (last tile row / column) read a neighbour's pixels: only compare the rest
  okmask = torch.ones(64, 4, 4, dtype=torch.bool)
  for m in range(64):
    n, rem = divmod(m, TH * TW)
    ti, tj = divmod(rem, TW)
    if 2 * ti + 3 >= Hp: okmask[m, 3, :] = False
    if 2 * tj + 3 >= Wp: okmask[m, :, 3] = False
  dv = (Vgot.float() - Vexp.float()).abs().amax(-1)
  print("  [dbg] V (k-step 0) vs torch: max diff %.3e over in-image positions (%d of %d (tile, position) pairs wrong); off-image positions max %.3e" % (
    dv[okmask].max().item(), int((dv[okmask] > 0).sum()), int(okmask.sum()), dv[~okmask].max().item() if (~okmask).any() else 0.0))
  if (dv[okmask] > 0).any():
    bad = (dv * okmask).nonzero()[:6]
    print("        first wrong (tile, xi, nu):", bad.tolist())
    # which hypothesis explains it?  transposed roles / swapped xi, nu
    Vt = Vexp.transpose(1, 2)
    print("        matches with xi <-> nu swapped: max diff %.3e" % (Vgot.float() - Vt.float()).abs().amax(-1)[okmask & okmask.transpose(1, 2)].max().item())
  # accumulators: [wave][lane][nu][mb][nb][16] fp32 -> M[xi = wave][nu][tile][cout]
  acc = raw[256 * 1024:256 * 1024 + 256 * 256 * 4].view(torch.float32).reshape(4, 64, 4, 2, 2, 16)
  Mgot = torch.zeros(4, 4, 64, 64)
  lane = torch.arange(64)
  for r in range(16):
    rows = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    for mb in range(2):
      for nb in range(2):
        Mgot[:, :, mb * 32 + rows, nb * 32 + (lane & 31)] = acc[:, :, :, mb, nb, r].permute(0, 2, 1)
  # expected M from torch: full-K product of bf16 V and bf16 U
  tiles_all = torch.zeros(64, 4, 4, C)
  for m in range(64):
    n, rem = divmod(m, TH * TW)
    ti, tj = divmod(rem, TW)
    ys, xs_ = min(4, Hp - 2 * ti), min(4, Wp - 2 * tj)
    tiles_all[m, :ys, :xs_] = xt[n, 2 * ti:2 * ti + ys, 2 * tj:2 * tj + xs_, :]
  Vall = torch.einsum("ar,mrsc,bs->mabc", BT.float(), tiles_all, BT.float()).to(torch.bfloat16).float()
  Gf = G.float()
  wc = w.float().cpu()
  U = torch.einsum("oias,bs->oiab", torch.einsum("ar,oirs->oias", Gf, wc), Gf).to(torch.bfloat16).float()   # [Co][Ci][4][4]
  Mexp = torch.einsum("mabc,ocab->abmo", Vall, U[:64])
  dm = (Mgot - Mexp).abs()
  okm = okmask.permute(1, 2, 0)[..., None].expand(4, 4, 64, 64)
  print("  [dbg] accumulators vs torch: max diff %.3e (max |M| %.3f) over in-image positions" % (dm[okm].max().item(), Mexp.abs().max().item()))
  if dm[okm].max().item() > 1e-2 * Mexp.abs().max().item():
    for xi in range(4):
      print("        xi %d: per-nu max diff %s" % (xi, ["%.2e" % dm[xi, nu][okm[xi, nu]].max().item() for nu in range(4)]))
  # Z[xi][b][tile][cout]
  Z = raw[1024 * 1024:1024 * 1024 + 4 * 2 * 64 * 64 * 4].view(torch.float32).reshape(4, 2, 64, 64)
  Zexp = torch.stack([Mgot[:, 0] + Mgot[:, 1] + Mgot[:, 2], Mgot[:, 1] - Mgot[:, 2] - Mgot[:, 3]], 1)
  print("  [dbg] Z vs nu-reduction of the dumped accumulators: max diff %.3e" % (Z - Zexp).abs().max().item())


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--n", type=int, default=660)
  ap.add_argument("--iters", type=int, default=30)
  ap.add_argument("--layers", type=str, default="l3,l2,l4")
  ap.add_argument("--dbg", action="store_true")
  ap.add_argument("--hold", type=str, default="", help="layer,kernel(wino|direct),seconds")
  a = ap.parse_args()
  dev = torch.device("cuda:0")
  L = probe_lib()
  if a.hold:
    name, kern, secs = a.hold.split(",")
    C, H, _ = LAYERS[name]
    x, w = make_case(C, H, a.n, dev)
    uf = torch.empty(16 * C * C, dtype=torch.bfloat16, device=dev)
    _lib.check(L.iic_probe_wino_weight_prep(w.data_ptr(), uf.data_ptr(), C, C, 0, _lib.stream_ptr()))
    y = torch.zeros(a.n, H + 2, H + 2, C, dtype=torch.bfloat16, device=dev)
    st = ops.new_stats(C, dev)
    gf = geom.fwd_geom(geom.ConvSpec(C, C, 3, 1, 1), a.n, H, H, 1, 1)
    pw = ops.PreppedWeights(w)
    if kern == "wino":
      fn = lambda: _lib.check(L.iic_probe_wino_fwd(x.data_ptr(), uf.data_ptr(), y.data_ptr(), st.data_ptr(), a.n, H, H, C, C, None, 0, _lib.stream_ptr()))
    else:
      fn = lambda: ops.conv_igemm(gf, x, pw[0], y, stats=st)
    t_end = time.time() + float(secs)
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() < t_end:
      for _ in range(50):
        fn()
      n += 50
      torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("HOLD %s %s: %d launches, %.1f us each (%.0f TF/s direct-equivalent)" % (name, kern, n, us, 2.0 * a.n * H * H * C * C * 9 / us / 1e6))
    return
  res = []
  for name in a.layers.split(","):
    res.append(run_layer(L, name, a.n, a.iters, dev, a.dbg))
  tot_d = sum(r["t_direct"] * r["count"] for r in res)
  tot_w = sum(r["t_wino"] * r["count"] for r in res)
  print("forward launches of these shapes per pass: direct %.2f ms, winograd %.2f ms (x %.2f); all parity gates %s" % (
    tot_d / 1e3, tot_w / 1e3, tot_d / max(tot_w, 1e-9), "ok" if all(r["ok"] for r in res) else "FAILED"))


if __name__ == "__main__":
  main()
