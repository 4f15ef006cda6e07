"""Host-side geometry of the implicit-GEMM convolution (include/iic_hip.h:iic_conv_geom).

A convolution pass (forward, or one stride-parity class of backward-data) is described
to the HIP kernel as: GEMM row m = (n, y, x)  ->  an input pixel index, an output pixel
index (both in the padded-flat "PT" index space) and a list of taps (pixel offset, weight
slice).  Everything here is pure Python/numpy so that the mapping is unit-tested on CPU
against torch.nn.functional.conv2d (tests/test_geom_cpu.py) with ``emulate_*`` below --
the emulators are test helpers, never a product fallback.

Reference semantics reproduced: nn.Conv2d(k, stride, padding, dilation, bias=False) as
used in /root/reference/code/archs/cluster/residual.py:4-7,54-55 and vgg.py:24-26.
"""
import numpy as np

from ._lib import ConvGeom, IIC_MAX_TAPS

BM = 128


class ConvSpec(object):
  """One nn.Conv2d: square kernel K, stride s, padding p, dilation d."""

  def __init__(self, cin, cout, K, stride=1, pad=0, dil=1):
    self.cin, self.cout, self.K, self.s, self.p, self.d = cin, cout, K, stride, pad, dil

  def out_size(self, h):
    return (h + 2 * self.p - self.d * (self.K - 1) - 1) // self.s + 1

  @property
  def taps(self):
    return self.K * self.K


def rows_per_image(g):
  """GEMM rows per image: the plane MY*MX, or the padded plane g.MP (see _finish)."""
  return int(g.MP) if g.MP > 0 else g.MY * g.MX


def gemm_rows(g):
  return g.N * rows_per_image(g)


def _nr(g, m):
  """GEMM row -> (image, in-plane index clamped to the last pixel, valid)."""
  plane, mp = g.MY * g.MX, rows_per_image(g)
  n = m // mp
  r = m - n * mp
  valid = (n < g.N) & (r < plane)
  r = np.minimum(r, plane - 1)
  r = np.where(n >= g.N, plane - 1, r)
  n = np.minimum(n, g.N - 1)
  return n, r, valid


def _pin(g, m):
  n, r, _ = _nr(g, np.asarray(m))
  y = r // g.MX
  x = r - y * g.MX
  return (n * g.in_Hp + y * g.sy + g.oy) * g.in_Wp + x * g.sx + g.ox


def _pout(g, m):
  n, r, _ = _nr(g, np.asarray(m))
  y = r // g.MX
  x = r - y * g.MX
  return (n * g.out_Hp + y * g.ty + g.py) * g.out_Wp + x * g.tx + g.px


def _finish(g, taps):
  assert 1 <= len(taps) <= IIC_MAX_TAPS
  g.ntaps = len(taps)
  for i, (off, tw) in enumerate(taps):
    assert off >= 0
    g.tap_off[i] = off
    g.tap_w[i] = tw
  def spans():
    M = gemm_rows(g)
    for bm, field in ((BM, "NP"), (2 * BM, "NP256"), (BM // 2, "NP64")):
      m0 = np.arange(0, M, bm, dtype=np.int64)
      m1 = np.minimum(m0 + bm, M) - 1
      span = int((_pin(g, m1) - _pin(g, m0)).max())
      setattr(g, field, span + max(off for off, _ in taps) + 1)
  g.MP = 0
  spans()
  # Large images: a dense row numbering lets a tile straddle two images, and its input span then
  # contains the 2*pad border rows between them (200x200 with pad 3: +1236 pixels = 178 KB of
  # LDS).  Padding the per-image row count to a multiple of 256 keeps every tile inside one
  # image (rows >= plane are invalid: never stored, zero weight-gradient contribution) at
  # a few % extra rows.
  plane = g.MY * g.MX
  mp = (plane + 255) // 256 * 256
  if g.NP * 144 > 96 * 1024 and mp * 8 <= plane * 9:      # big span, <= 12.5 % padding rows
    dense = (g.NP, g.NP256, g.NP64)
    g.MP = mp
    spans()
    if g.NP >= dense[0]:                                    # no gain (span is all halo): stay dense
      g.MP = 0
      g.NP, g.NP256, g.NP64 = dense
  return g


def fwd_geom(spec, N, H, W, pad_in, pad_out):
  """Forward conv: in PT [N, H+2pad_in, W+2pad_in, cin] -> out PT [N, Ho+2pad_out, ...]."""
  assert pad_in >= spec.p, "input PT border must cover the conv padding"
  Ho, Wo = spec.out_size(H), spec.out_size(W)
  g = ConvGeom()
  g.N, g.MY, g.MX = N, Ho, Wo
  g.in_Hp, g.in_Wp, g.Cin = H + 2 * pad_in, W + 2 * pad_in, spec.cin
  g.sy = g.sx = spec.s
  g.oy = g.ox = pad_in - spec.p
  g.out_Hp, g.out_Wp, g.Cout = Ho + 2 * pad_out, Wo + 2 * pad_out, spec.cout
  g.ty = g.tx = 1
  g.py = g.px = pad_out
  taps = [(kh * spec.d * g.in_Wp + kw * spec.d, kh * spec.K + kw)
          for kh in range(spec.K) for kw in range(spec.K)]
  # last tap of the last row must stay inside the padded input
  assert (Ho - 1) * spec.s + g.oy + (spec.K - 1) * spec.d < g.in_Hp
  assert (Wo - 1) * spec.s + g.ox + (spec.K - 1) * spec.d < g.in_Wp
  return _finish(g, taps)


def bwd_data_geoms(spec, N, H, W, pad_dy, pad_dx):
  """Backward-data: dY PT [N, Ho+2pad_dy, ., cout] -> dX PT [N, H+2pad_dx, ., cin].
  One geometry per stride-parity class of the input pixel (s*s classes, empty ones dropped);
  uses the [T][cin][cout] weight copy.  Returns a list of ConvGeom writing disjoint pixels."""
  Ho, Wo = spec.out_size(H), spec.out_size(W)
  s, p, d, K = spec.s, spec.p, spec.d, spec.K
  geoms = []
  for ph in range(s):
    ah = [((ph + p - kh * d) // s, kh) for kh in range(K) if (ph + p - kh * d) % s == 0]
    for pw in range(s):
      aw = [((pw + p - kw * d) // s, kw) for kw in range(K) if (pw + p - kw * d) % s == 0]
      MY = (H - ph + s - 1) // s
      MX = (W - pw + s - 1) // s
      if not ah or not aw or MY <= 0 or MX <= 0:
        continue
      ah_min = min(a for a, _ in ah)
      aw_min = min(a for a, _ in aw)
      g = ConvGeom()
      g.N, g.MY, g.MX = N, MY, MX
      g.in_Hp, g.in_Wp, g.Cin = Ho + 2 * pad_dy, Wo + 2 * pad_dy, spec.cout
      g.sy = g.sx = 1
      g.oy, g.ox = ah_min + pad_dy, aw_min + pad_dy
      assert g.oy >= 0 and g.ox >= 0, "dY border too small for backward-data"
      ah_max = max(a for a, _ in ah)
      aw_max = max(a for a, _ in aw)
      assert (MY - 1) + ah_max + pad_dy < g.in_Hp and (MX - 1) + aw_max + pad_dy < g.in_Wp, \
        "dY border too small for backward-data"
      g.out_Hp, g.out_Wp, g.Cout = H + 2 * pad_dx, W + 2 * pad_dx, spec.cin
      g.ty = g.tx = s
      g.py, g.px = ph + pad_dx, pw + pad_dx
      taps = [((a - ah_min) * g.in_Wp + (b - aw_min), kh * K + kw) for a, kh in ah for b, kw in aw]
      geoms.append(_finish(g, taps))
  return geoms


def bwd_data_covers_all(spec):
  """True when every input-pixel parity class receives at least one tap (else the caller
  must zero / accumulate the untouched classes)."""
  s, p, d, K = spec.s, spec.p, spec.d, spec.K
  for ph in range(s):
    if not any((ph + p - kh * d) % s == 0 for kh in range(K)):
      return False
  return True


def geom_key(g):
  return (g.N, g.MY, g.MX, g.in_Hp, g.in_Wp, g.Cin, g.sy, g.sx, g.oy, g.ox, g.out_Hp, g.out_Wp,
          g.Cout, g.ty, g.tx, g.py, g.px, g.ntaps, tuple(g.tap_off[:g.ntaps]),
          tuple(g.tap_w[:g.ntaps]), g.NP, g.NP256, g.NP64, g.MP)


# ---------------------------------------------------------------------------------------
# numpy emulators of the kernels' contract (TEST HELPERS for the host logic above)
# ---------------------------------------------------------------------------------------

def emulate_igemm(g, x_pt, w_tco_ci, out_pt=None, accumulate=False):
  """out[pout(m)] (+)= sum_t x[pin(m)+off_t] @ w[tap_w[t]].T ; arrays are float64 numpy."""
  m = np.arange(gemm_rows(g), dtype=np.int64)
  m = m[_nr(g, m)[2]]                      # valid rows only
  M = len(m)
  pin, pout = _pin(g, m), _pout(g, m)
  xf = x_pt.reshape(-1, g.Cin)
  if out_pt is None:
    out_pt = np.zeros((g.N, g.out_Hp, g.out_Wp, g.Cout))
  of = out_pt.reshape(-1, g.Cout)
  acc = np.zeros((M, g.Cout))
  for t in range(g.ntaps):
    acc += xf[pin + g.tap_off[t]] @ w_tco_ci[g.tap_w[t]].T
  if accumulate:
    of[pout] += acc
  else:
    of[pout] = acc
  return out_pt


def emulate_wgrad(g, x_pt, dy_pt, wtaps):
  """dW[t][co][ci] = sum_m dy[pout(m)][co] * x[pin(m)+off_t][ci] (forward geometry)."""
  m = np.arange(gemm_rows(g), dtype=np.int64)
  m = m[_nr(g, m)[2]]                      # valid rows only
  pin, pout = _pin(g, m), _pout(g, m)
  xf = x_pt.reshape(-1, g.Cin)
  dyf = dy_pt.reshape(-1, g.Cout)
  dW = np.zeros((wtaps, g.Cout, g.Cin))
  for t in range(g.ntaps):
    dW[g.tap_w[t]] += dyf[pout].T @ xf[pin + g.tap_off[t]]
  return dW


def to_pt(x_nchw, pad):
  """NCHW float array -> PT layout [N, H+2p, W+2p, C] (numpy)."""
  n, c, h, w = x_nchw.shape
  out = np.zeros((n, h + 2 * pad, w + 2 * pad, c), dtype=x_nchw.dtype)
  out[:, pad:pad + h, pad:pad + w, :] = np.transpose(x_nchw, (0, 2, 3, 1))
  return out


def from_pt(x_pt, pad):
  n, hp, wp, c = x_pt.shape
  return np.transpose(x_pt[:, pad:hp - pad, pad:wp - pad, :], (0, 3, 1, 2))
