"""fp32 GEMM of the heads (iic_gemm_f32, iic_amd/csrc/head.hip): every kernel family behind the one entry point
-- one-wave tiles, K-split waves, and the LDS-tiled kernel in its four operand-stride modes and three K-split
group counts -- against a float64 product of the same operands.  The reference computes these products with
nn.Linear / autograd in fp32 (/root/reference/code/archs/cluster/net6c.py:47-50, net5g.py:51-54);
tolerance: 2e-5 of the result's scale (fp32 accumulation over K <= 4608 terms)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(M, N, K, a_kc, b_kc, bias, accumulate, seed=0):
  from iic_amd import ops
  g = torch.Generator(device="cpu").manual_seed(seed)
  dev = "cuda:0"
  # operands stored so that the unit stride runs along k (a_kc / b_kc) or along m / n
  A = (torch.randn(M, K, generator=g) if a_kc else torch.randn(K, M, generator=g)).to(dev)
  B = (torch.randn(N, K, generator=g) if b_kc else torch.randn(K, N, generator=g)).to(dev)
  bv = torch.randn(N, generator=g).to(dev) if bias else None
  C0 = torch.randn(M, N, generator=g).to(dev)
  C = C0.clone()
  sam, sak = (K, 1) if a_kc else (1, M)
  sbk, sbn = (1, K) if b_kc else (N, 1)
  ops.gemm_f32(A, sam, sak, B, sbk, sbn, C, N, M, N, K, bias=bv, accumulate=accumulate)
  C2 = C0.clone()
  ops.gemm_f32(A, sam, sak, B, sbk, sbn, C2, N, M, N, K, bias=bv, accumulate=accumulate)
  torch.cuda.synchronize()
  Ad = (A if a_kc else A.t()).double()
  Bd = (B.t() if b_kc else B).double()
  want = Ad @ Bd
  if bias:
    want = want + bv.double()
  if accumulate:
    want = want + C0.double()
  assert torch.equal(C, C2), "two launches over the same operands must agree bit for bit"
  err = (C.double() - want).abs().max().item()
  scale = want.abs().max().item()
  assert err <= 2e-5 * scale, (M, N, K, a_kc, b_kc, err, scale)


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("b_kc", [True, False])
@pytest.mark.parametrize("shape", [
  (700, 1400, 4608),      # ClusterNet6c k = 280 logits: many tiles, one group
  (700, 250, 4608),       # k = 50: 44 tiles, four K-split groups
  (700, 700, 2048),       # 121 tiles: two groups
  (333, 517, 1031),       # ragged in every dimension
  (64, 512, 256),         # the smallest product the tiled kernel takes
])
def test_tiled_gemm_modes(shape, a_kc, b_kc):
  M, N, K = shape
  _run(M, N, K, a_kc, b_kc, bias=a_kc, accumulate=b_kc)


@pytest.mark.parametrize("shape", [(660, 70, 512), (32, 10, 4608), (700, 50, 4608), (5, 3, 17)])
def test_small_gemm_families(shape):
  M, N, K = shape
  _run(M, N, K, True, True, bias=True, accumulate=False)
  _run(M, N, K, False, False, bias=False, accumulate=True)
