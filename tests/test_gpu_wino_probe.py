"""The Winograd F(2x2, 3x3) probe kernel (iic_amd/csrc/probes/wino_probe.hip; LAB.md section R5.1) is not part of the
product, but its measured no-go is only worth something if the kernel is RIGHT: forward convolution + BatchNorm
statistics against float64 F.conv2d on the same bf16 operands, at small batches of the three layer shapes, odd sizes
(tile quantisation: the last tile row / column runs off the image) and a ragged last workgroup tile; the transformed
weights against their torch restatement, bit for bit."""
import ctypes
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _probe():
  from iic_amd import _lib
  path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libiic_probe.so")
  if not os.path.exists(path):
    pytest.skip("libiic_probe.so not built (make -C iic_amd/csrc probes)")
  L = ctypes.CDLL(path)
  L.iic_probe_wino_fwd.restype = ctypes.c_int
  L.iic_probe_wino_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
  L.iic_probe_wino_weight_prep.restype = ctypes.c_int
  L.iic_probe_wino_weight_prep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
  return L


@pytest.mark.parametrize("C,H,N", [(128, 25, 5), (256, 13, 21), (512, 7, 37), (128, 8, 3), (64, 9, 7)])
def test_winograd_probe_forward_and_statistics(C, H, N):
  from iic_amd import _lib, ops
  from tools.winograd_probe import ufrag_torch
  L = _probe()
  dev = torch.device("cuda:0")
  g = torch.Generator().manual_seed(C + H)
  x = torch.zeros(N, H + 2, H + 2, C)
  x[:, 1:-1, 1:-1, :] = torch.randn(N, H, H, C, generator=g).relu()
  w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
  x, w = x.to(torch.bfloat16).to(dev), w.to(dev)
  uf = torch.empty(16 * C * C, dtype=torch.bfloat16, device=dev)
  _lib.check(L.iic_probe_wino_weight_prep(w.data_ptr(), uf.data_ptr(), C, C, 0, _lib.stream_ptr()), "weight prep")
  y = torch.zeros(N, H + 2, H + 2, C, dtype=torch.bfloat16, device=dev)
  st = ops.new_stats(C, dev)
  rc = L.iic_probe_wino_fwd(x.data_ptr(), uf.data_ptr(), y.data_ptr(), st.data_ptr(), N, H, H, C, C, None, 0, _lib.stream_ptr())
  if rc == -3:
    pytest.skip("geometry outside the probe's LDS budget")
  _lib.check(rc, "wino fwd")
  torch.cuda.synchronize()
  assert torch.equal(uf.view(torch.int16).cpu(), ufrag_torch(w).view(torch.int16).cpu())      # G g G^T, rounded once
  ref = F.conv2d(x[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).double().cpu(), w.to(torch.bfloat16).double().cpu(), padding=1)
  got = y[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).double().cpu()
  mx = ref.abs().max().item()
  err = (got - ref).abs().max().item()
  assert err <= 2.0 ** -6 * mx, (err, mx, math.log2(err / mx))
  assert y[:, 0].abs().max() == 0 and y[:, -1].abs().max() == 0 and y[:, :, 0].abs().max() == 0 and y[:, :, -1].abs().max() == 0
  # statistics: sum y, sum y^2 of the fp32 results before rounding -- against the stored (bf16) outputs
  s = ops.stats_decode(st, C).cpu()
  s1, s2 = got.sum((0, 2, 3)), (got * got).sum((0, 2, 3))
  assert ((s[0] - s1).abs() / (s2.sqrt() + 1e-9)).max().item() <= 2e-2
  assert ((s[1] - s2).abs() / s2).max().item() <= 2e-2
