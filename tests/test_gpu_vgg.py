"""ClusterNet6c pieces on the HIP path (through the C ABI): first-layer fp32-MFMA conv and
its weight gradient, 2x2 max-pool, VGG stages teacher-forced against the bf16-emulating
oracle, and the whole net against the reference golden.  pytest -m gpu."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def dev():
  return torch.device("cuda:0")


def _cos(a, b):
  a, b = a.double().flatten(), b.double().flatten()
  return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("cin,K,S,N", [(1, 5, 24, 5), (5, 5, 24, 3), (4, 3, 40, 2), (5, 3, 64, 2), (2, 3, 96, 2),
                                       # round 6 (firstconv2.hip: bands of rows in LDS): the Potsdam / COCO widths, a band
                                       # count that does not divide the height, rectangular images, one image row per
                                       # 32-pixel tile boundary case (W = 32), and widths the banded kernels do not take
                                       # (W % 4 != 0: first-generation kernels)
                                       (4, 3, 200, 2), (5, 3, 128, 1), (3, 5, 28, 3), (1, 3, (20, 36), 2), (8, 3, (36, 32), 2),
                                       (4, 3, 30, 2), (2, 5, (24, 26), 2)])
def test_firstconv_forward_and_wgrad(cin, K, S, N):
  from iic_amd import ops
  pad, P = (K - 1) // 2, 2
  rng = np.random.default_rng(cin * 10 + K)
  SH, SW = S if isinstance(S, tuple) else (S, S)
  x = torch.from_numpy(rng.standard_normal((N, cin, SH, SW)).astype(np.float32))
  w = torch.from_numpy((rng.standard_normal((64, cin, K, K)) * 0.2).astype(np.float32))
  wt = w.clone().requires_grad_(True)
  y = F.conv2d(x, wt, padding=pad)
  d = dev()
  out = torch.zeros((N, SH + 2 * P, SW + 2 * P, 64), dtype=torch.bfloat16, device=d)
  st = ops.new_stats(64, d)
  ops.firstconv_fwd(x.to(d), w.to(d), out, st, K, pad, P)
  torch.cuda.synchronize()
  got = ops.pt_to_nchw(out, P).cpu()
  scale = float(y.abs().max())
  assert float((got - y.detach()).abs().max()) <= 1e-2 * scale
  assert out[:, :P].abs().max() == 0 and out[:, :, -P:].abs().max() == 0
  assert out[:, -P:].abs().max() == 0 and out[:, :, :P].abs().max() == 0
  cnt = N * SH * SW
  ssum = ops.stats_decode(st, 64).float().cpu()
  assert torch.allclose(ssum[0] / cnt, y.detach().mean((0, 2, 3)), atol=1e-4 * scale)
  assert torch.allclose(ssum[1] / cnt, (y.detach() ** 2).mean((0, 2, 3)), rtol=1e-4, atol=1e-5)
  dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).to(torch.bfloat16).float()
  y.backward(dy)
  dW = ops.firstconv_wgrad(x.to(d), ops.pt_from_nchw(dy.to(d), P), tuple(w.shape), K, pad, P)
  torch.cuda.synchronize()
  assert float((dW.cpu() - wt.grad).abs().max()) <= 1e-3 * float(wt.grad.abs().max())


@pytest.mark.parametrize("C,S", [(64, 24), (128, 12), (256, 6)])
def test_maxpool2_forward_backward(C, S):
  from iic_amd import ops
  N, P = 3, 2
  rng = np.random.default_rng(C)
  x = torch.from_numpy(rng.standard_normal((N, C, S, S)).astype(np.float32)).to(torch.bfloat16).float()
  xt = x.clone().requires_grad_(True)
  y = F.max_pool2d(xt, 2, 2)
  dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).to(torch.bfloat16).float()
  y.backward(dy)
  d = dev()
  xp = ops.pt_from_nchw(x.to(d), P)
  out = torch.zeros((N, S // 2 + 2 * P, S // 2 + 2 * P, C), dtype=torch.bfloat16, device=d)
  ops.maxpool2_fwd(xp, out, N, S, S, P, P, C)
  din = torch.zeros_like(xp)
  ops.maxpool2_bwd(xp, ops.pt_from_nchw(dy.to(d), P), din, N, S, S, P, P, C)
  torch.cuda.synchronize()
  assert torch.equal(ops.pt_to_nchw(out, P).cpu(), y.detach())
  assert torch.equal(ops.pt_to_nchw(din, P).cpu(), xt.grad)


@pytest.mark.parametrize("C,H,W", [(64, 24, 24), (128, 12, 12), (256, 6, 6), (64, 7, 9), (128, 200, 10)])
def test_fused_bn_relu_maxpool_is_bit_identical_to_the_two_pass_path(C, H, W):
  """The pooled stages never store relu(bn(y)): the pool recomputes it from (y, coef) -- forward and backward against
  iic_bn_apply followed by the plain pool, bit for bit, on activations full of ties (ReLU zeros) and with odd sizes
  (rows / columns no window covers get zero gradient), borders untouched."""
  from iic_amd import ops
  N, P = 3, 2
  d = dev()
  g = torch.Generator().manual_seed(C + H)
  y = torch.zeros((N, H + 2 * P, W + 2 * P, C), dtype=torch.bfloat16)
  y[:, P:P + H, P:P + W] = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16)
  y = y.to(d)
  coef = torch.zeros(5, C)
  coef[0] = torch.randn(C, generator=g)          # scales of both signs
  coef[1] = torch.randn(C, generator=g) * 0.5
  coef = coef.to(d)
  Ho, Wo = H // 2, W // 2
  dout = torch.zeros((N, Ho + 2 * P, Wo + 2 * P, C), dtype=torch.bfloat16)
  dout[:, P:P + Ho, P:P + Wo] = torch.randn(N, Ho, Wo, C, generator=g).to(torch.bfloat16)
  dout = dout.to(d)
  a = torch.zeros_like(y)
  ops.bn_apply(y, coef, a, N, H, W, P, C, relu=True)
  o_ref = torch.full((N, Ho + 2 * P, Wo + 2 * P, C), 7.0, dtype=torch.bfloat16, device=d)
  o_fus = o_ref.clone()
  ops.maxpool2_fwd(a, o_ref, N, H, W, P, P, C)
  ops.bn_relu_maxpool2_fwd(y, coef, o_fus, N, H, W, P, P, C)
  d_ref = torch.full_like(y, 7.0)
  d_fus = torch.full_like(y, 7.0)
  ops.maxpool2_bwd(a, dout, d_ref, N, H, W, P, P, C)
  ops.bn_relu_maxpool2_bwd(y, coef, dout, d_fus, N, H, W, P, P, C)
  torch.cuda.synchronize()
  assert torch.equal(o_ref, o_fus) and torch.equal(d_ref, d_fus)
  assert float((a == 0).float().mean()) > 0.2          # the ties are really there
  border = d_fus.clone()
  border[:, P:P + H, P:P + W] = 7.0
  assert (border == 7.0).all()


def test_net6c_step_with_and_without_the_fused_pool_is_bit_identical():
  """Whole ClusterNet6c train step (three pooled stages): IIC_FUSE_POOL on / off -- outputs, loss and every gradient."""
  from iic_amd import archs
  from iic_amd.archs import vgg
  from iic_amd.losses import IID_loss
  from oracle import net_oracle
  cfg = types.SimpleNamespace(in_channels=1, input_sz=24, batchnorm_track=True, num_sub_heads=2, output_k=10)
  params = net_oracle.make_net6c_params(1, 24, 10, 2, True, seed=4, randomize_bn=True, head_std=0.05)
  x6, x6t = net_oracle.make_paired_batch(24, 24, 3, seed=6)
  res = {}
  try:
    for fused in (True, False):
      vgg.FUSE_POOL[0] = fused
      net = archs.ClusterNet6c(cfg)
      net.load_state_dict(params, strict=True)
      net.to(dev()).train()
      xo, xt = net(x6.to(dev())), net(x6t.to(dev()))
      tot = sum(IID_loss(xo[i], xt[i], lamb=1.0)[0] for i in range(2)) / 2
      tot.backward()
      torch.cuda.synchronize()
      res[fused] = ([o.detach().clone() for o in xo + xt], tot.detach().clone(),
                    [p.grad.detach().clone() for p in net.parameters()])
  finally:
    vgg.FUSE_POOL[0] = True
  for a, b in zip(res[True][0] + [res[True][1]] + res[True][2], res[False][0] + [res[False][1]] + res[False][2]):
    assert torch.equal(a, b)


@pytest.mark.parametrize("idx,first,pool,cin,S", [(0, True, True, 1, 24), (4, False, True, 64, 12), (12, False, False, 256, 3)])
def test_vgg_stage_teacher_forced(idx, first, pool, cin, S):
  """One conv-BN-ReLU(-pool) stage Function, forward + backward, vs the bf16-emulating oracle
  on the same input and upstream gradient."""
  from iic_amd import archs, ops
  from oracle import net_oracle
  N = 16
  cfg = types.SimpleNamespace(in_channels=1, input_sz=24, batchnorm_track=True, num_sub_heads=2, output_k=10)
  full = net_oracle.make_net6c_params(1, 24, 10, 2, True, seed=4, randomize_bn=True)
  net = archs.ClusterNet6c(cfg)
  net.load_state_dict(full, strict=True)
  net.to(dev()).train()
  st = [s for s in net.trunk._stages if s.conv is net.trunk.features[idx]][0]
  rng = np.random.default_rng(idx)
  x = torch.from_numpy(rng.standard_normal((N, cin, S, S)).astype(np.float32))
  if not first:
    x = x.relu().to(torch.bfloat16).float()
  params = {k: v.clone() for k, v in full.items()}
  for k in ("trunk.features.%d.weight" % idx, "trunk.features.%d.weight" % (idx + 1), "trunk.features.%d.bias" % (idx + 1)):
    params[k].requires_grad_(True)
  xe = x.clone().requires_grad_(True)
  oe = net_oracle.vgg_stage_bf16emu(params, idx, xe, 2, 1, pool, first, True)
  dout = torch.from_numpy(rng.standard_normal(tuple(oe.shape)).astype(np.float32)).to(torch.bfloat16).float()
  oe.backward(dout)
  from iic_amd.archs.vgg import _StageFn
  xin = x.to(dev()) if first else ops.pt_from_nchw(x.to(dev()), 2).requires_grad_(True)
  o = _StageFn.apply(xin, st.conv.weight, st.bn.weight, st.bn.bias, st)
  o.backward(ops.pt_from_nchw(dout.to(dev()), 2))
  torch.cuda.synchronize()
  got = ops.pt_to_nchw(o.detach(), 2).cpu()
  scale = float(oe.abs().max())
  assert float((got - oe.detach()).abs().max()) <= 2e-2 * scale
  assert float((got - oe.detach()).abs().mean()) <= 2e-3 * scale
  if not first:
    gx = ops.pt_to_nchw(xin.grad, 2).cpu()
    assert _cos(gx, xe.grad) >= 0.999
  for p_, key in ((st.conv.weight, "trunk.features.%d.weight" % idx), (st.bn.weight, "trunk.features.%d.weight" % (idx + 1)),
                  (st.bn.bias, "trunk.features.%d.bias" % (idx + 1))):
    c = _cos(p_.grad.cpu(), params[key].grad)
    r = float(p_.grad.norm().cpu() / params[key].grad.norm())
    assert c >= 0.998 and abs(r - 1) < 2e-2, (key, c, r)


def test_net6c_vs_reference_golden_and_twohead():
  from iic_amd import archs
  from iic_amd.losses import IID_loss
  from oracle import iid_oracle, net_oracle
  g = np.load(os.path.join(G, "nets.npz"))
  cfg = types.SimpleNamespace(in_channels=1, input_sz=24, batchnorm_track=True, num_sub_heads=2, output_k=10)
  params = net_oracle.make_net6c_params(1, 24, 10, 2, True, seed=4, randomize_bn=True, head_std=0.05)
  net = archs.ClusterNet6c(cfg)
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  x6, x6t = net_oracle.make_paired_batch(24, 24, 3, seed=6)
  xo, xt = net(x6.to(dev())), net(x6t.to(dev()))
  tot = sum(IID_loss(xo[i], xt[i], lamb=1.0)[0] for i in range(2)) / 2
  tot.backward()
  torch.cuda.synchronize()
  out = np.stack([o.detach().cpu().numpy() for o in xo])
  # bf16-emulating oracle
  ep = {k: v.clone() for k, v in params.items()}
  for k, v in ep.items():
    if v.dtype.is_floating_point and "running" not in k:
      v.requires_grad_(True)
  exo = net_oracle.net6c_forward_bf16emu(ep, x6, True, "head", 2)
  ext = net_oracle.net6c_forward_bf16emu(ep, x6t, True, "head", 2)
  eloss = sum(iid_oracle.IID_loss(exo[i], ext[i], 1.0)[0] for i in range(2)) / 2
  eloss.backward()
  eout = np.stack([o.detach().numpy() for o in exo])
  rep = {"mean_abs_vs_emu": float(np.abs(out - eout).mean()), "max_abs_vs_emu": float(np.abs(out - eout).max()),
         "mean_abs_vs_ref": float(np.abs(out - g["net6c_out"]).mean()),
         "emu_vs_ref_mean": float(np.abs(eout - g["net6c_out"]).mean()),
         "loss": float(tot), "loss_emu": float(eloss), "loss_ref": float(g["net6c_loss"][0])}
  os.makedirs("gpurun_out", exist_ok=True)
  open("gpurun_out/net6c_report.txt", "w").write("%s\n" % rep)
  assert rep["mean_abs_vs_emu"] <= 3 * rep["emu_vs_ref_mean"] + 2e-3, rep
  assert abs(rep["loss"] - rep["loss_emu"]) <= 5e-2 * abs(rep["loss_emu"]) + 1e-4, rep
  assert abs(rep["loss"] - rep["loss_ref"]) <= 1e-1 * abs(rep["loss_ref"]), rep
  cs = []
  for n, p in net.named_parameters():
    ref = ep[n].grad
    if float(ref.norm()) > 1e-7:
      cs.append((n, _cos(p.grad.cpu(), ref)))
  open("gpurun_out/net6c_grads.txt", "w").write("\n".join("%s %.4f" % c for c in cs))
  assert np.median([c for _, c in cs]) >= 0.97 and min(c for _, c in cs) >= 0.9, cs
  # two-head variant + eval / no_grad
  net2 = archs.ClusterNet6cTwoHead(types.SimpleNamespace(
    in_channels=1, input_sz=24, batchnorm_track=True, num_sub_heads=3, output_k_A=50, output_k_B=10)).to(dev())
  net2.eval()
  with torch.no_grad():
    oa, ob = net2(x6.to(dev()), head="A"), net2(x6.to(dev()))
    tf = net2(x6.to(dev()), trunk_features=True)
  assert oa[0].shape == (24, 50) and ob[2].shape == (24, 10) and tf.shape == (24, 4608)
  assert torch.allclose(oa[1].sum(1), torch.ones(24, device=dev()), atol=1e-5)


def test_net6c_fp32_mode_vs_reference_golden():
  """Parity tier T2 for the VGG-style trunk: the whole ClusterNet6c train step on the exact-fp32
  kernels (ops.fp32_mode()) against the reference's own fp32 golden -- outputs, loss, gradients."""
  from iic_amd import archs, ops
  from iic_amd.losses import IID_loss
  from oracle import net_oracle
  g = np.load(os.path.join(G, "nets.npz"))
  cfg = types.SimpleNamespace(in_channels=1, input_sz=24, batchnorm_track=True, num_sub_heads=2, output_k=10)
  params = net_oracle.make_net6c_params(1, 24, 10, 2, True, seed=4, randomize_bn=True, head_std=0.05)
  net = archs.ClusterNet6c(cfg)
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  x6, x6t = net_oracle.make_paired_batch(24, 24, 3, seed=6)
  with ops.fp32_mode():
    xo, xt = net(x6.to(dev())), net(x6t.to(dev()))
  tot = sum(IID_loss(xo[i], xt[i], lamb=1.0)[0] for i in range(2)) / 2
  tot.backward()
  torch.cuda.synchronize()
  out = np.stack([o.detach().cpu().numpy() for o in xo])
  assert np.abs(out - g["net6c_out"]).max() <= 2e-4, np.abs(out - g["net6c_out"]).max()
  lref = float(g["net6c_loss"][0])
  assert abs(float(tot.detach()) - lref) <= 5e-4 * abs(lref) + 1e-7, (float(tot.detach()), lref)
  for n, p in net.named_parameters():
    gn = g["net6c_grad/" + n][0]
    assert abs(float(p.grad.double().norm()) - gn) <= 1e-2 * max(gn, 1e-6) + 1e-9, (n, float(p.grad.double().norm()), gn)


def test_net6c_input_sz_64_fp32_mode_vs_oracle():
  """ClusterNet6c's second supported geometry (net6c.py:42-45: input_sz 64 -> 8 x 8 x 512 features into the heads):
  whole train step on the exact-fp32 kernels against the CPU restatement of the reference on the same parameters
  and batch -- outputs, loss, gradient norms -- and the bf16 path's outputs inside the bf16 tier."""
  from iic_amd import archs, ops
  from iic_amd.losses import IID_loss
  from oracle import net_oracle, iid_oracle
  cfg = types.SimpleNamespace(in_channels=1, input_sz=64, batchnorm_track=True, num_sub_heads=2, output_k=10)
  params = net_oracle.make_net6c_params(1, 64, 10, 2, True, seed=9, randomize_bn=True, head_std=0.05)
  x6, x6t = net_oracle.make_paired_batch(12, 64, 3, seed=10)
  # oracle (CPU, fp32 autograd)
  rp = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
        for k, v in params.items()}
  ro, rt = net_oracle.net6c_forward(rp, x6, True, "head", 2), net_oracle.net6c_forward(rp, x6t, True, "head", 2)
  rtot = sum(iid_oracle.IID_loss(ro[i], rt[i], lamb=1.0)[0] for i in range(2)) / 2
  rtot.backward()
  net = archs.ClusterNet6c(cfg)
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  with ops.fp32_mode():
    xo, xt = net(x6.to(dev())), net(x6t.to(dev()))
  tot = sum(IID_loss(xo[i], xt[i], lamb=1.0)[0] for i in range(2)) / 2
  tot.backward()
  torch.cuda.synchronize()
  for i in range(2):
    assert (xo[i].detach().cpu() - ro[i].detach()).abs().max().item() <= 2e-4
    assert (xt[i].detach().cpu() - rt[i].detach()).abs().max().item() <= 2e-4
  lref = float(rtot.detach())
  assert abs(float(tot.detach()) - lref) <= 5e-4 * abs(lref) + 1e-7, (float(tot.detach()), lref)
  for n, p in net.named_parameters():
    gn = float(rp[n].grad.double().norm())
    assert abs(float(p.grad.double().norm()) - gn) <= 1e-2 * max(gn, 1e-6) + 1e-9, (n, float(p.grad.double().norm()), gn)
  # bf16 path (the measured configuration): same batch, outputs inside the bf16 tier, backward finite
  net.zero_grad()
  bo = net(x6.to(dev()))
  (sum(IID_loss(bo[i], net(x6t.to(dev()))[i], lamb=1.0)[0] for i in range(2)) / 2).backward()
  torch.cuda.synchronize()
  for i in range(2):
    d = (bo[i].detach().cpu() - ro[i].detach()).abs()
    assert d.mean().item() <= 1e-2, d.mean().item()
    assert (bo[i].detach().cpu().argmax(1) == ro[i].detach().argmax(1)).float().mean().item() >= 0.9
  assert all(torch.isfinite(p.grad).all() for p in net.parameters())
