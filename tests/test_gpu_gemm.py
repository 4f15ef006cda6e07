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


@pytest.mark.gpu
def test_k_split_workspace_is_never_freed_when_it_grows():
  """the K-split GEMM workspace of a branch used to be re-allocated when a later call needed
  more (head A's 660 x 350 logits after head B's 660 x 50): the old tensor went back to the caching allocator with
  its address still baked into head B's captured graphs.  Superseded workspaces are now kept alive."""
  import torch
  from iic_amd import ops
  dev = torch.device("cuda:0")
  ops._GEMM_WS.clear()
  del ops._GEMM_WS_RETIRED[:]

  def logits(n, k_out, kdim=512):
    x = torch.randn(n, kdim, device=dev)
    w = torch.randn(k_out, kdim, device=dev)
    c = torch.empty(n, k_out, device=dev)
    ops.gemm_f32(x, kdim, 1, w, 1, kdim, c, k_out, n, k_out, kdim)
    torch.cuda.synchronize()
    assert torch.allclose(c, x @ w.t(), rtol=2e-4, atol=2e-3)
  logits(660, 50)
  key = (ops.BRANCH[0], dev.index)
  first = ops._GEMM_WS.get(key)
  if first is None:
    pytest.skip("this shape takes no K split on this build")
  p0, n0 = first.data_ptr(), first.numel()
  logits(4096, 700)                       # a larger request on the same branch
  second = ops._GEMM_WS[key]
  if second.data_ptr() != p0:
    assert second.numel() > n0
    assert any(t.data_ptr() == p0 for t in ops._GEMM_WS_RETIRED), "the superseded workspace was freed"
  logits(660, 50)                         # and the small shape still computes right
