"""Host logic of the implicit-GEMM conv geometry vs torch conv2d (CPU, float64)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from iic_amd import geom

CASES = [  # cin, cout, K, stride, pad, dil, N, H, W, pad_in, pad_out
  (8, 16, 3, 1, 1, 1, 3, 7, 7, 1, 1),
  (8, 16, 3, 1, 1, 1, 5, 13, 13, 1, 1),
  (8, 8, 3, 2, 1, 1, 3, 13, 13, 1, 1),
  (8, 8, 3, 2, 1, 1, 2, 49, 49, 1, 1),
  (8, 16, 1, 2, 0, 1, 3, 13, 13, 1, 1),
  (8, 8, 5, 1, 2, 1, 2, 12, 12, 2, 2),
  (8, 8, 3, 1, 1, 2, 2, 12, 12, 2, 2),   # dilated, pad 1 => output shrinks by 2
  (4, 4, 3, 1, 1, 1, 2, 66, 66, 3, 3),   # large image, wide border: padded per-image row count (MP)
  (4, 4, 3, 1, 1, 2, 2, 70, 70, 3, 3),   # same, dilated (segmentation trunk)
]


def _mk(cin, cout, K, s, p, d, N, H, W):
  rng = np.random.default_rng(0)
  x = rng.standard_normal((N, cin, H, W))
  w = rng.standard_normal((cout, cin, K, K))
  return x, w


@pytest.mark.parametrize("case", CASES)
def test_forward_geometry(case):
  cin, cout, K, s, p, d, N, H, W, pi, po = case
  spec = geom.ConvSpec(cin, cout, K, s, p, d)
  x, w = _mk(cin, cout, K, s, p, d, N, H, W)
  ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=s, padding=p, dilation=d).numpy()
  g = geom.fwd_geom(spec, N, H, W, pi, po)
  w_t = np.transpose(w, (2, 3, 0, 1)).reshape(K * K, cout, cin)
  out = geom.emulate_igemm(g, geom.to_pt(x, pi), w_t)
  assert np.abs(geom.from_pt(out, po) - ref).max() < 1e-10
  if po > 0:  # border untouched (zero)
    assert not out[:, :po].any() and not out[:, :, :po].any()
  # patch bound: every row's taps stay inside [p_lo, p_lo + NP)
  M = geom.gemm_rows(g)
  if H * W >= 4096:
    assert g.MP > 0 and g.MP % 256 == 0 and g.MP >= g.MY * g.MX   # padded planes: tiles never straddle images
  m = np.arange(M)
  pin = geom._pin(g, m)
  p_lo = pin[(m // 128) * 128]
  assert ((pin - p_lo + max(g.tap_off[:g.ntaps])) < g.NP).all()
  assert (pin - p_lo >= 0).all()


@pytest.mark.parametrize("case", CASES)
def test_backward_data_geometry(case):
  cin, cout, K, s, p, d, N, H, W, pi, po = case
  spec = geom.ConvSpec(cin, cout, K, s, p, d)
  x, w = _mk(cin, cout, K, s, p, d, N, H, W)
  xt = torch.from_numpy(x).requires_grad_(True)
  y = F.conv2d(xt, torch.from_numpy(w), stride=s, padding=p, dilation=d)
  dy = torch.from_numpy(np.random.default_rng(1).standard_normal(tuple(y.shape)))
  y.backward(dy)
  pad_dy = max(po, (K - 1) * d - p)
  geoms = geom.bwd_data_geoms(spec, N, H, W, pad_dy, pi)
  w_b = np.transpose(w, (2, 3, 1, 0)).reshape(K * K, cin, cout)
  dx = np.zeros((N, H + 2 * pi, W + 2 * pi, cin))
  for g in geoms:
    geom.emulate_igemm(g, geom.to_pt(dy.numpy(), pad_dy), w_b, dx)
  assert np.abs(geom.from_pt(dx, pi) - xt.grad.numpy()).max() < 1e-10
  if pi > 0:
    assert not dx[:, :pi].any() and not dx[:, :, :pi].any()


@pytest.mark.parametrize("case", CASES)
def test_weight_grad_geometry(case):
  cin, cout, K, s, p, d, N, H, W, pi, po = case
  spec = geom.ConvSpec(cin, cout, K, s, p, d)
  x, w = _mk(cin, cout, K, s, p, d, N, H, W)
  wt = torch.from_numpy(w).requires_grad_(True)
  y = F.conv2d(torch.from_numpy(x), wt, stride=s, padding=p, dilation=d)
  dy = torch.from_numpy(np.random.default_rng(1).standard_normal(tuple(y.shape)))
  y.backward(dy)
  g = geom.fwd_geom(spec, N, H, W, pi, po)
  dW = geom.emulate_wgrad(g, geom.to_pt(x, pi), geom.to_pt(dy.numpy(), po), K * K)
  ref = np.transpose(wt.grad.numpy(), (2, 3, 0, 1)).reshape(K * K, cout, cin)
  assert np.abs(dW - ref).max() < 1e-9
