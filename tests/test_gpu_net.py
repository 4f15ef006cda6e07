"""End-to-end parity of the HIP ClusterNet5g train step (through the C ABI) against the
CPU oracle and the reference-generated golden fixtures.  Needs an MI355X: pytest -m gpu.

bf16 activations cannot meet the fp32 loss clause across a 36-layer conv stack (SURVEY.md
§8c T3).  Tiers used here:
  * per BasicBlock, teacher-forced (same input / same upstream gradient as the bf16-storage
    emulation of the oracle): outputs within 2e-2*max (2e-3*max mean), dx and every parameter
    gradient cosine >= 0.998, norms within 2 %;
  * stem parameters: fp32 CPU reference of conv+BN+ReLU+maxpool fed OUR upstream gradient,
    cosine >= 0.995;
  * whole net vs the bf16 emulation and the reference golden: robust aggregates only (loss
    within 5 % / 10 %, mean |dprob|, argmax agreement) -- a 33-layer batch-stat-BN net on a
    24-image batch amplifies 1-ulp differences chaotically (measured CPU-only yardstick in
    the test body).
The fp32 loss clause itself is tested on identical (x, x_tf) loss inputs in
test_gpu_kernels.py.
"""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def dev():
  return torch.device("cuda:0")


def _cfg(**kw):
  base = dict(in_channels=2, input_sz=32, batchnorm_track=True, num_sub_heads=2, output_k=10)
  base.update(kw)
  return types.SimpleNamespace(**base)


def _cos(a, b):
  a, b = a.double().flatten(), b.double().flatten()
  return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("use_tr", [True, False])
def test_net5g_small_vs_reference_golden(use_tr):
  """Same weights / inputs as tests/golden/nets.npz (produced by the reference's own
  ClusterNet5g + IID_loss): forward probabilities, loss and all parameter gradients."""
  from iic_amd import archs
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle
  g = np.load(os.path.join(G, "nets.npz"))
  params = net_oracle.make_net5g_params(2, 10, 2, True, seed=3, randomize_bn=True, head_std=0.3)
  net = archs.ClusterNet5g(_cfg())
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  net.set_wgrad_tr(use_tr)
  imgs, imgs_tf = net_oracle.make_paired_batch(24, 32, 3, seed=5)
  a = sobel_process(imgs.to(dev()), False)
  b = sobel_process(imgs_tf.to(dev()), False)
  # capture d(loss)/d(stem output) of both passes to check the stem backward in isolation
  dpools = []
  def _pre(m, inp):
    inp[0].register_hook(lambda gr: dpools.append(gr.detach().clone()))
    return None
  hook = net.trunk.layer1.register_forward_pre_hook(_pre)
  xo, xt = net(a), net(b)
  hook.remove()
  tot = None
  for i in range(2):
    l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
    tot = l if tot is None else tot + l
  tot = tot / 2
  tot.backward()
  torch.cuda.synchronize()
  out = np.stack([o.detach().cpu().numpy() for o in xo])
  out_tf = np.stack([o.detach().cpu().numpy() for o in xt])
  assert np.allclose(out.sum(-1), 1.0, atol=1e-5)
  # (1) the bf16-storage emulation of the oracle (same rounding points as the HIP path):
  #     what separates it from the GPU result is accumulation order only.
  eparams = {k: v.clone() for k, v in params.items()}
  for k, v in eparams.items():
    if v.dtype.is_floating_point and "running" not in k:
      v.requires_grad_(True)
  exo = net_oracle.net5g_forward_bf16emu(eparams, net_oracle.sobel_process(imgs, False), True, 32, "head", 2)
  ext = net_oracle.net5g_forward_bf16emu(eparams, net_oracle.sobel_process(imgs_tf, False), True, 32, "head", 2)
  from oracle import iid_oracle
  eloss = sum(iid_oracle.IID_loss(exo[i], ext[i], 1.0)[0] for i in range(2)) / 2
  eloss.backward()
  eout = np.stack([o.detach().numpy() for o in exo])
  eout_tf = np.stack([o.detach().numpy() for o in ext])
  report = {"out_err_vs_bf16emu": float(np.abs(out - eout).max()),
            "out_tf_err_vs_bf16emu": float(np.abs(out_tf - eout_tf).max()),
            "out_err_vs_fp32_reference": float(np.abs(out - g["net5g_out"]).max()),
            "bf16emu_vs_fp32_reference": float(np.abs(eout - g["net5g_out"]).max()),
            "loss": float(tot.detach()), "loss_bf16emu": float(eloss), "loss_fp32_reference": float(g["net5g_loss"][0])}
  os.makedirs("gpurun_out", exist_ok=True)
  with open("gpurun_out/net5g_small_report_tr%d.txt" % int(use_tr), "w") as f:
    f.write("%s\n" % report)
  # A 33-layer net with batch-statistics BN on a 24-image batch amplifies 1-ulp differences
  # (accumulation order) chaotically, so whole-net agreement is judged on robust aggregates;
  # exact per-block parity is test_basic_block_teacher_forced below.
  mean_emu = float(np.abs(out - eout).mean())
  mean_inherent = float(np.abs(eout - g["net5g_out"]).mean())
  report["mean_abs_vs_bf16emu"], report["mean_abs_bf16emu_vs_fp32"] = mean_emu, mean_inherent
  with open("gpurun_out/net5g_small_report_tr%d.txt" % int(use_tr), "w") as f:
    f.write("%s\n" % report)
  assert mean_emu <= 3.0 * mean_inherent + 2e-3, report
  assert (out.argmax(-1) == eout.argmax(-1)).mean() >= 0.9, report
  # The statistics are accumulated exactly (round 2), so a given build is bit-reproducible run to run; the VALUE depends
  # on how the launches group their statistic partials: -0.014923 in rounds 2-3, -0.014511 since round 4's 128-row tiles
  # for launches that fill less than half of the chip's workgroup slots (csrc/conv_igemm_bd.hip bd_pick_ms; with
  # iic_debug_bd_ms=4 the instrumented library gives -0.014923 again: profiles/r05_smoke_bisect.txt) -- 4.5 % / 1.8 %
  # from the fp32 reference.  MI ~ 0 on this 24-image fixture: the loss is a difference of nearly equal terms and a
  # one-ulp change in a partial sum moves it by percents, which is why the 5 % gates below are all this fixture can carry
  # and why smoke() and the tight gates use the 96-image fixture (loss -0.40) instead.
  assert abs(report["loss"] - report["loss_bf16emu"]) < 5e-2 * abs(report["loss_bf16emu"]), report
  assert abs(report["loss"] - report["loss_fp32_reference"]) < 5e-2 * abs(report["loss_fp32_reference"]), report
  # gradients vs the bf16-emulating oracle (straight-through rounding)
  table = []
  for n, p in net.named_parameters():
    ref = eparams[n].grad
    if float(ref.norm()) < 1e-7:
      continue
    c = _cos(p.grad.cpu(), ref)
    r = float(p.grad.double().norm().cpu() / ref.double().norm())
    table.append((n, c, r))
  with open("gpurun_out/net5g_small_grads_tr%d.txt" % int(use_tr), "w") as f:
    for n, c, r in table:
      f.write("%-45s cos %.4f  norm ratio %.4f\n" % (n, c, r))
  # the fp32 oracle must agree with the reference golden (ties the checker to the reference)
  oparams = {k: v.clone() for k, v in params.items()}
  for k, v in oparams.items():
    if v.dtype.is_floating_point and "running" not in k:
      v.requires_grad_(True)
  loss_o, _, _, _ = net_oracle.net5g_train_step_loss(oparams, imgs, imgs_tf, 1.0, 32, 2)
  loss_o.backward()
  for n, p in net.named_parameters():
    gn = g["net5g_grad/" + n][0]
    assert abs(float(oparams[n].grad.double().norm()) - gn) <= 2e-3 * max(gn, 1e-3), n
  # whole-net gradients (chaotic regime, see above).  Yardstick measured on CPU with NO GPU
  # code involved: the bf16-emulating oracle vs the fp32 oracle on this very fixture gives
  # cosine 0.96 (layer4) -> 0.82 (layer1), median 0.87, min 0.76.  The HIP path must be in
  # that class w.r.t. the emulation; tight gradient parity is the teacher-forced block test.
  cs = np.array([c for _, c, _ in table])
  assert np.median(cs) >= 0.85 and cs.min() >= 0.75, (float(np.median(cs)), float(cs.min()))
  # stem parameters: feed OUR upstream gradient into an fp32 CPU reference of the stem
  # (conv3x3 + BN + ReLU + maxpool) -- isolates the stem kernels from upstream bf16 noise.
  import torch.nn.functional as F
  from iic_amd import ops
  assert len(dpools) == 2   # backward order: x_tf pass first or second, match by position
  w0 = params["trunk.conv1.weight"].clone().requires_grad_(True)
  g0 = params["trunk.bn1.weight"].clone().requires_grad_(True)
  b0 = params["trunk.bn1.bias"].clone().requires_grad_(True)
  best = None
  for order in ((0, 1), (1, 0)):
    for t in (w0, g0, b0):
      t.grad = None
    for inp, di in zip((a, b), order):
      y = F.conv2d(inp.cpu(), w0, padding=1)
      pl = F.max_pool2d(F.relu(F.batch_norm(y, None, None, g0, b0, True, 0.1, 1e-5)), 2, 2, padding=1)
      pl.backward(ops.pt_to_nchw(dpools[di], 1).cpu())
    c = _cos(net.trunk.conv1.weight.grad.cpu(), w0.grad)
    if best is None or c > best[0]:
      best = (c, _cos(net.trunk.bn1.weight.grad.cpu(), g0.grad), _cos(net.trunk.bn1.bias.grad.cpu(), b0.grad),
              float(net.trunk.conv1.weight.grad.norm().cpu() / w0.grad.norm()))
  assert best[0] >= 0.995 and best[1] >= 0.995 and best[2] >= 0.995 and 0.98 <= best[3] <= 1.02, best
  # running statistics follow nn.BatchNorm2d (two updates: x pass, x_tf pass)
  sd = net.state_dict()
  assert np.allclose(sd["trunk.bn1.running_mean"].cpu().numpy(), g["net5g_rm_bn1"], atol=2e-3)
  assert np.allclose(sd["trunk.bn1.running_var"].cpu().numpy(), g["net5g_rv_bn1"], rtol=2e-2)
  assert int(sd["trunk.bn1.num_batches_tracked"]) == 2


def test_net5g_eval_nograd_and_twohead():
  from iic_amd import archs
  net = archs.ClusterNet5gTwoHead(types.SimpleNamespace(
    in_channels=2, input_sz=32, batchnorm_track=True, num_sub_heads=3, output_k_A=20,
    output_k_B=10)).to(dev())
  x = torch.rand(8, 2, 32, 32, device=dev())
  net.train()
  with torch.no_grad():
    oa = net(x, head="A")
    ob = net(x)   # default head B
  assert len(oa) == 3 and oa[0].shape == (8, 20) and ob[0].shape == (8, 10)
  net.eval()
  with torch.no_grad():
    oe = net(x, head="B")
    feats = net(x, trunk_features=True)
  assert feats.shape == (8, 512) and torch.isfinite(oe[0]).all()
  assert torch.allclose(oe[0].sum(1), torch.ones(8, device=dev()), atol=1e-5)


def test_net5g_96_step_runs_and_decreases_loss():
  """North-star shape (96x96, k=70, 5 sub-heads) at a small batch: a few Adam steps of the
  full train step (sobel -> 2 forwards -> 5x IID_loss -> backward -> Adam) reduce the loss."""
  from iic_amd import archs
  from iic_amd.losses import IID_loss_heads
  from iic_amd.optim import Adam
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle
  torch.manual_seed(0)
  net = archs.ClusterNet5g(_cfg(input_sz=96, num_sub_heads=5, output_k=70)).to(dev()).train()
  opt = Adam(net.parameters(), lr=1e-4)
  imgs, imgs_tf = net_oracle.make_paired_batch(24, 96, 3, seed=0)
  imgs, imgs_tf = imgs.to(dev()), imgs_tf.to(dev())
  losses = []
  for step in range(4):
    net.zero_grad()
    xo = net.forward_packed(sobel_process(imgs, False))
    xt = net.forward_packed(sobel_process(imgs_tf, False))
    l, _ = IID_loss_heads(xo, xt, lamb=1.0)
    loss = l.mean()
    loss.backward()
    opt.step()
    losses.append(loss.item())
  assert all(np.isfinite(losses)), losses
  assert losses[-1] < losses[0], losses


def test_replica_deduplication():
  """all_imgs = r exact replicas of the unique images
  (cluster_sobel.py:215-226).  Forwarding the unique rows once and repeating the features
  reproduces the full replicated forward/backward: BN batch statistics are replication-invariant,
  the backward is linear in the upstream gradient, the unbiased running_var factor uses the true
  batch size.
  (a) exactness where depth cannot amplify rounding: the stem (conv + BN + ReLU + pool) alone;
  (b) whole net: agreement class of two runs that differ only in rounding (see the golden test)."""
  from iic_amd import archs, ops
  from iic_amd.archs import cluster as cl
  from iic_amd.losses import IID_loss_heads
  from iic_amd.transforms import sobel_process
  torch.manual_seed(1)
  g = torch.Generator().manual_seed(11)
  base = torch.rand(64, 1, 64, 64, generator=g)
  imgs = base.repeat(3, 1, 1, 1).to(dev())
  imgs_tf = torch.clamp(torch.flip(imgs, dims=[3]) * 0.9 + 0.05 * torch.rand(192, 1, 64, 64, generator=g).to(dev()), 0, 1)
  ref_net = archs.ClusterNet5g(_cfg(input_sz=64, num_sub_heads=2, output_k=10)).to(dev()).train()
  with torch.no_grad():
    for h in ref_net.head.heads:
      h[0].weight.normal_(0, 0.3)
  state = {k: v.clone() for k, v in ref_net.state_dict().items()}

  # ---- (a) stem only: full replicated batch vs unique batch with summed upstream gradients
  stem = {}
  gup = torch.randn(192, 35, 35, 64, generator=g).to(dev()).to(torch.bfloat16)   # PT layout, border 1
  gup[:, 0] = 0; gup[:, -1] = 0; gup[:, :, 0] = 0; gup[:, :, -1] = 0
  gsum = (gup[:64].float() + gup[64:128].float() + gup[128:].float())
  for mode in ("full", "unique"):
    net = archs.ClusterNet5g(_cfg(input_sz=64, num_sub_heads=2, output_k=10)).to(dev()).train()
    net.load_state_dict(state)
    t = net.trunk
    x = sobel_process(imgs if mode == "full" else imgs[:64], False)
    ops.BN_REPLICAS[0] = 1 if mode == "full" else 3
    try:
      out = cl._StemFn.apply(x, t.conv1.weight, t.bn1.weight, t.bn1.bias, t)
    finally:
      ops.BN_REPLICAS[0] = 1
    o = out.detach().float().clone()
    out.backward(gup if mode == "full" else gsum.to(torch.bfloat16))
    torch.cuda.synchronize()
    stem[mode] = (o, t.conv1.weight.grad.clone(), t.bn1.weight.grad.clone(), t.bn1.bias.grad.clone(),
                  t.bn1.running_mean.clone(), t.bn1.running_var.clone())
  of, wf, gf, bf, rmf, rvf = stem["full"]
  ou, wu, gu, bu, rmu, rvu = stem["unique"]
  d = (of[:64] - ou).abs()
  assert float(d.max()) <= 2 ** -7 * float(ou.abs().max()) and float((d > 0).float().mean()) < 1e-3
  assert torch.equal(of[:64], of[64:128]) and torch.equal(of[:64], of[128:])
  assert torch.allclose(rmf, rmu, rtol=1e-5, atol=1e-7) and torch.allclose(rvf, rvu, rtol=1e-5, atol=1e-8)
  # the summed gradient is rounded to bf16 once more than the three replicas' gradients
  assert _cos(wf, wu) > 0.9995 and _cos(gf, gu) > 0.9995 and _cos(bf, bu) > 0.9995
  assert abs(float(wu.norm() / wf.norm()) - 1) < 5e-3

  # ---- (b) whole net
  res = {}
  for r in (1, 3, "asserted"):
    net = archs.ClusterNet5g(_cfg(input_sz=64, num_sub_heads=2, output_k=10)).to(dev()).train()
    net.load_state_dict(state)
    if r == "asserted":      # caller-asserted replication: no row comparison, no host sync
      with cl.replicated(3):
        xo = net.forward_packed(sobel_process(imgs, False))
      xt = net.forward_packed(sobel_process(imgs_tf, False))
    else:
      cl.DEDUP[0] = r
      try:
        xo = net.forward_packed(sobel_process(imgs, False))
        xt = net.forward_packed(sobel_process(imgs_tf, False))
      finally:
        cl.DEDUP[0] = 1
    assert xo.shape == (192, 2, 10)
    loss, _ = IID_loss_heads(xo, xt, lamb=1.0)
    loss.mean().backward()
    torch.cuda.synchronize()
    res[r] = (xo.detach().clone(), float(loss.mean().detach()), {n: p.grad.clone() for n, p in net.named_parameters()},
              {k: v.clone() for k, v in net.state_dict().items() if "running" in k})
  (o1, l1, g1, s1), (o3, l3, g3, s3) = res[1], res[3]
  # the asserted path is the same computation as the compared one: bit-identical
  oa, la, ga, sa = res["asserted"]
  assert torch.equal(oa, o3) and la == l3 and all(torch.equal(ga[n], g3[n]) for n in g3)
  # independent reference: the oracle (bf16-storage emulation of the reference net) on the FULL
  # replicated batch vs the de-duplicated HIP forward
  from oracle import net_oracle
  cpu_state = {k: v.detach().cpu() for k, v in state.items()}
  with torch.no_grad():
    ref = net_oracle.net5g_forward_bf16emu(cpu_state, net_oracle.sobel_process(imgs.cpu(), False), True, 64,
                                           "head", 2)
  ref = torch.stack(ref, dim=1)
  assert float((o3.cpu() - ref).abs().mean()) < 6e-3, float((o3.cpu() - ref).abs().mean())
  assert torch.equal(o3[:64], o3[64:128]) and torch.equal(o3[:64], o3[128:])
  assert float((o1 - o3).abs().mean()) < 4e-3 and abs(l1 - l3) < 3e-4, (l1, l3)    # loss ~ -4e-4 here
  # Gradients of the bf16 production path: the two runs differ by where bf16 rounding happens (the
  # de-duplicated backward sums three replicas' upstream gradients before they enter the trunk), and
  # the loss of this fixture is ~ -4e-4 (MI ~ 0: dL/dz is a difference of nearly equal terms), so
  # element-level agreement is not meaningful in bf16 -- the EXACT statement (outputs, loss and every
  # parameter gradient of the de-duplicated path against the reference's own fp32 golden of the
  # replicated batch) is test_replica_dedup_fp32_mode_vs_reference_golden below; here only the
  # summary is recorded and loosely bounded.
  names = [n for n in g1 if float(g1[n].norm()) > 1e-8]
  cs = {n: _cos(g1[n], g3[n]) for n in names}
  rs = {n: float(g3[n].norm() / g1[n].norm()) for n in names}
  os.makedirs("gpurun_out", exist_ok=True)
  with open("gpurun_out/dedup_grads.txt", "w") as f:
    for n in names:
      f.write("%-45s cos %.4f  norm ratio %.4f\n" % (n, cs[n], rs[n]))
  assert abs(np.median(list(rs.values())) - 1.0) < 0.25, float(np.median(list(rs.values())))
  k = "trunk.bn1.running_var"
  assert torch.allclose(s1[k], s3[k], rtol=1e-4, atol=1e-6), (s1[k] - s3[k]).abs().max()


def test_net5g_bf16_large_batch_vs_reference_golden():
  """The bf16 PRODUCTION kernels (MFMA convolutions, bf16 activations, fused epilogues) across the whole
  ClusterNet5g against the reference's own fp32 result on a fixture OUTSIDE the chaotic regime
  (VERDICT r2 weak #3 / next 4c): 96 images (32 x 3 replicas), 64 x 64, 2 sub-heads, loss -0.40
  (tests/golden/net5g_large.npz, oracle/gen_golden_large.py).  The 24-image 32 x 32 fixture above has
  MI ~ 0 and only supports robust aggregates; here the loss is held to 1 % and every parameter
  gradient to a cosine against the reference's gradient (complete for small parameters, a fixed
  strided 16 384-element sample for the large convolution weights)."""
  from iic_amd import archs
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle
  from oracle.gen_golden_large import HEADS, INPUT_SZ, K, N_PAIRS, sample_stride
  g = np.load(os.path.join(G, "net5g_large.npz"))
  params = net_oracle.make_net5g_params(2, K, HEADS, True, seed=13, randomize_bn=True, head_std=0.03)
  for k in g.files:
    if k.startswith("param/"):
      params[k[6:]] = torch.from_numpy(g[k])
  net = archs.ClusterNet5g(_cfg(input_sz=INPUT_SZ, num_sub_heads=HEADS, output_k=K))
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  imgs, imgs_tf = net_oracle.make_mild_pair(N_PAIRS, INPUT_SZ, 3, seed=21)
  xo = net(sobel_process(imgs.to(dev()), False))
  xt = net(sobel_process(imgs_tf.to(dev()), False))
  tot = None
  for i in range(HEADS):
    l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
    tot = l if tot is None else tot + l
  tot = tot / HEADS
  tot.backward()
  torch.cuda.synchronize()
  out = np.stack([o.detach().cpu().numpy() for o in xo])
  out_tf = np.stack([o.detach().cpu().numpy() for o in xt])
  lref = float(g["loss"][0])
  # the bf16-storage emulation of the oracle on the same fixture (CPU, no HIP code): the yardstick for
  # what bf16 activations alone do to a 33-BatchNorm net, and -- same rounding points as the HIP path --
  # the tight reference for the gradients
  from oracle import iid_oracle
  eparams = {k: v.clone() for k, v in params.items()}
  for k, v in eparams.items():
    if v.dtype.is_floating_point and "running" not in k:
      v.requires_grad_(True)
  exo = net_oracle.net5g_forward_bf16emu(eparams, net_oracle.sobel_process(imgs, False), True, INPUT_SZ, "head", HEADS)
  ext = net_oracle.net5g_forward_bf16emu(eparams, net_oracle.sobel_process(imgs_tf, False), True, INPUT_SZ, "head", HEADS)
  eloss = sum(iid_oracle.IID_loss(exo[i], ext[i], 1.0)[0] for i in range(HEADS)) / HEADS
  eloss.backward()
  eout = np.stack([o.detach().numpy() for o in exo])
  rows, cs, rs, ce, cy = [], [], [], [], []
  for n, p in net.named_parameters():
    gd = p.grad.detach().flatten()
    st = sample_stride(gd.numel())
    ours = gd[::st].double().cpu()
    ref = torch.from_numpy(g["grad/" + n]).double()
    emu = eparams[n].grad.detach().flatten()
    gn = float(g["gnorm/" + n][0])
    if gn < 1e-9:
      continue
    c, r = _cos(ours, ref), float(gd.double().norm()) / gn
    c_emu = _cos(gd.double().cpu(), emu.double())             # vs the emulation: the FULL gradient
    c_yard = _cos(emu[::st].double(), ref)                    # emulation vs fp32 reference (no HIP code)
    rows.append((n, c, r, c_emu, c_yard))
    cs.append(c); rs.append(r); ce.append(c_emu); cy.append(c_yard)
  cs, rs, ce, cy = np.array(cs), np.array(rs), np.array(ce), np.array(cy)
  os.makedirs("gpurun_out", exist_ok=True)
  with open("gpurun_out/net5g_large_report.txt", "w") as f:
    f.write("loss %.6f, bf16-emulating oracle %.6f, fp32 reference %.6f (rel %.2e); mean|dprob| vs reference %.2e / %.2e "
            "(emulation vs reference %.2e), vs emulation %.2e; argmax agreement %.3f\n" % (
      float(tot.detach()), float(eloss), lref, abs(float(tot.detach()) - lref) / abs(lref), np.abs(out - g["out"]).mean(),
      np.abs(out_tf - g["out_tf"]).mean(), np.abs(eout - g["out"]).mean(), np.abs(out - eout).mean(),
      float((out.argmax(-1) == g["out"].argmax(-1)).mean())))
    f.write("gradient cosine vs fp32 reference: median %.4f min %.4f (bf16 emulation vs reference: median %.4f min %.4f); "
            "vs bf16 emulation: median %.4f min %.4f; norm ratio vs reference median %.4f min %.4f max %.4f\n" % (
      np.median(cs), cs.min(), np.median(cy), cy.min(), np.median(ce), ce.min(), np.median(rs), rs.min(), rs.max()))
    for n, c, r, c_emu, c_yard in rows:
      f.write("%-45s cos vs ref %.4f (emulation vs ref %.4f)  vs emulation %.4f  norm ratio %.4f\n" % (n, c, c_yard, c_emu, r))
  assert abs(float(tot.detach()) - lref) <= 1e-2 * abs(lref), (float(tot.detach()), lref)
  assert abs(float(tot.detach()) - float(eloss)) <= 5e-3 * abs(lref), (float(tot.detach()), float(eloss))
  assert np.abs(out - g["out"]).mean() <= 1.5 * np.abs(eout - g["out"]).mean() + 5e-4
  assert np.abs(out - eout).mean() <= 6e-3, float(np.abs(out - eout).mean())
  assert (out.argmax(-1) == g["out"].argmax(-1)).mean() >= 0.97
  # against the fp32 reference bf16 STORAGE itself costs cosine (emulation vs reference, no HIP code involved:
  # median 0.88, min 0.77 on this fixture); the HIP path must be in the emulation's class there ...
  assert np.median(cs) >= np.median(cy) - 0.02 and cs.min() >= cy.min() - 0.10, (float(np.median(cs)), float(np.median(cy)), cs.min(), cy.min())
  assert abs(np.median(rs) - 1.0) <= 0.03 and rs.min() >= 0.85 and rs.max() <= 1.15, (float(np.median(rs)), rs.min(), rs.max())
  # ... and in the same class against the emulation itself.  Measured: median 0.92, min 0.86 -- and the
  # emulation run on two different CPUs (BLAS summation order) differs from ITSELF by about as much
  # (loss -0.402629 in the build container, -0.402053 on the GPU box's host): at 96 images a 1-ulp
  # difference in accumulation order still flips bf16 rounding decisions that 33 batch-statistics
  # BatchNorm layers amplify, so a cosine of 0.97 between two valid bf16 executions of this net does not
  # exist; gradient NORMS (median ratio 1.000), the loss (0.14 %) and the probabilities (4e-3) are the
  # quantities bf16 storage preserves.  Exact gradient parity is the fp32-mode test (1e-3, every parameter).
  assert np.median(ce) >= 0.88 and ce.min() >= 0.75, (float(np.median(ce)), float(ce.min()))


def test_net5g_fp32_mode_large_batch_gradients_vs_reference_golden():
  """The exact-fp32 kernels (`ops.fp32_mode()`) on the fixture OUTSIDE the chaotic regime (tests/golden/net5g_large.npz:
  the reference's own ClusterNet5g + IID_loss, fp32, CPU; 96 images of 64 x 64, loss -0.40), gradients element by
  element.  What the gate can be was MEASURED (oracle/gen_golden_large_f64.py): the reference's own
  float32 gradients sit 3.4e-3 (median relative L2 per parameter, worst 9.1e-3) from the float64 evaluation of the
  same graph -- 33 batch-statistics BatchNorm layers amplify float32 rounding in backward although loss and
  probabilities agree to 2e-6 / 1.5e-5 -- so no float32 implementation with another summation order can meet the
  float32 golden to 2e-4.  The gate that exists: against FLOAT64 (tests/golden/net5g_large_f64.npz) the HIP fp32 path
  must be in the class of the reference's own float32 run -- measured: median error 3.8e-3 against the reference's
  3.4e-3 (ratio of the medians 1.11), worst 9.8e-3 against 9.1e-3; per parameter the ratio has median 1.45 and
  exceeds 3 for three BatchNorm parameters of the LAST block (layer4.2.bn2.weight: 1.2e-3 against 1.6e-5), where
  both runs are far below everyone else's error and one ReLU-mask decision at a pre-activation of ~0 moves a
  ~900-term sum by 1e-3.  Gates: per parameter <= 2.5 x the reference's error + 2.5e-3, median <= 1.3 x the
  reference's median; outputs to 2e-5 and loss to 1e-5 of the float32 golden."""
  from iic_amd import archs, ops
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle
  from oracle.gen_golden_large import HEADS, INPUT_SZ, K, N_PAIRS, sample_stride
  from oracle.gen_golden_large_f64 import SUB
  g = np.load(os.path.join(G, "net5g_large.npz"))
  g64 = np.load(os.path.join(G, "net5g_large_f64.npz"))
  params = net_oracle.make_net5g_params(2, K, HEADS, True, seed=13, randomize_bn=True, head_std=0.03)
  for k in g.files:
    if k.startswith("param/"):
      params[k[6:]] = torch.from_numpy(g[k])
  net = archs.ClusterNet5g(_cfg(input_sz=INPUT_SZ, num_sub_heads=HEADS, output_k=K))
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  imgs, imgs_tf = net_oracle.make_mild_pair(N_PAIRS, INPUT_SZ, 3, seed=21)
  with ops.fp32_mode():
    xo = net(sobel_process(imgs.to(dev()), False))
    xt = net(sobel_process(imgs_tf.to(dev()), False))
  tot = None
  for i in range(HEADS):
    l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
    tot = l if tot is None else tot + l
  tot = tot / HEADS
  tot.backward()
  torch.cuda.synchronize()
  out = np.stack([o.detach().cpu().numpy() for o in xo])
  out_tf = np.stack([o.detach().cpu().numpy() for o in xt])
  d_out = max(np.abs(out - g["out"]).max(), np.abs(out_tf - g["out_tf"]).max())
  lref = float(g["loss"][0])
  d_loss = abs(float(tot.detach()) - lref) / abs(lref)
  rows = []
  for n, p in net.named_parameters():
    if "grad64/" + n not in g64.files:
      continue
    gd = p.grad.detach().flatten()
    ours = gd[::sample_stride(gd.numel())][::SUB].double().cpu().numpy()
    exact = g64["grad64/" + n].astype(np.float64)
    assert ours.shape == exact.shape, (n, ours.shape, exact.shape)
    e_ours = float(np.linalg.norm(ours - exact) / max(np.linalg.norm(exact), 1e-30))
    rows.append((e_ours, float(g64["ref_err/" + n][0]), n))
  e_o = np.array([r[0] for r in rows])
  e_r = np.array([r[1] for r in rows])
  os.makedirs("gpurun_out", exist_ok=True)
  with open("gpurun_out/net5g_large_fp32_mode.txt", "w") as f:
    f.write("max|dprob| %.3e, loss %.9f vs float32 golden %.9f (rel %.2e), float64 %.9f\n" % (
      d_out, float(tot.detach()), lref, d_loss, float(g64["loss64"][0])))
    f.write("gradient relative L2 error vs FLOAT64 per parameter (%d parameters): HIP fp32 mode median %.3e worst %.3e | "
            "the reference's own float32 run median %.3e worst %.3e | worst ratio ours / reference's %.2f\n" % (
              len(rows), np.median(e_o), e_o.max(), np.median(e_r), e_r.max(), float((e_o / np.maximum(e_r, 1e-12)).max())))
    for e1, e2, n in sorted(rows, reverse=True):
      f.write("%.3e (reference float32: %.3e)  %s\n" % (e1, e2, n))
  assert d_out <= 2e-5, d_out
  assert d_loss <= 1e-5, (float(tot.detach()), lref)
  assert np.all(e_o <= 2.5 * e_r + 2.5e-3), sorted(rows, reverse=True)[:5]
  assert np.median(e_o) <= 1.3 * np.median(e_r), (float(np.median(e_o)), float(np.median(e_r)))
  sd = net.state_dict()
  assert np.allclose(sd["trunk.bn1.running_mean"].cpu().numpy(), g["state/trunk.bn1.running_mean"], atol=1e-5)
  assert np.allclose(sd["trunk.bn1.running_var"].cpu().numpy(), g["state/trunk.bn1.running_var"], rtol=1e-4, atol=1e-7)
  assert np.allclose(sd["trunk.layer4.2.bn2.running_var"].cpu().numpy(), g["state/trunk.layer4.2.bn2.running_var"], rtol=1e-3, atol=1e-7)


def test_replica_dedup_fp32_mode_vs_reference_golden():
  """SURVEY.md 8f rank 3, exact form.  tests/golden/nets.npz was produced by the
  reference's own ClusterNet5g + IID_loss on a batch whose first view is 8 base images replicated 3x
  (net_oracle.make_paired_batch(24, 32, 3): exactly cluster_sobel.py:215-226).  The de-duplicated
  forward (`with replicated(3)`: the trunk sees the 8 unique images, features are repeated, autograd
  sums the replicas' gradients, BatchNorm's unbiased running-variance factor uses the true batch size)
  runs here on the exact-fp32 kernels (`ops.fp32_mode()`), so nothing but fp32 summation order
  separates it from the reference's full 24-image forward: outputs, loss, EVERY parameter gradient
  and the running statistics must agree with the golden of the replicated batch."""
  from iic_amd import archs, ops
  from iic_amd.archs import cluster as cl
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle
  g = np.load(os.path.join(G, "nets.npz"))
  params = net_oracle.make_net5g_params(2, 10, 2, True, seed=3, randomize_bn=True, head_std=0.3)
  net = archs.ClusterNet5g(_cfg(input_sz=32, num_sub_heads=2, output_k=10))
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  imgs, imgs_tf = net_oracle.make_paired_batch(24, 32, 3, seed=5)
  assert torch.equal(imgs[:8], imgs[8:16]) and torch.equal(imgs[:8], imgs[16:])
  with ops.fp32_mode():
    with cl.replicated(3):
      xo = net(sobel_process(imgs.to(dev()), False))          # trunk runs on the 8 unique images
    xt = net(sobel_process(imgs_tf.to(dev()), False))
  tot = None
  for i in range(2):
    l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
    tot = l if tot is None else tot + l
  tot = tot / 2
  tot.backward()
  torch.cuda.synchronize()
  out = np.stack([o.detach().cpu().numpy() for o in xo])
  out_tf = np.stack([o.detach().cpu().numpy() for o in xt])
  assert np.abs(out - g["net5g_out"]).max() <= 2e-4, np.abs(out - g["net5g_out"]).max()
  assert np.abs(out_tf - g["net5g_out_tf"]).max() <= 2e-4
  assert np.array_equal(out[:, :8], out[:, 8:16]) and np.array_equal(out[:, :8], out[:, 16:])
  lref = float(g["net5g_loss"][0])
  assert abs(float(tot.detach()) - lref) <= 1e-3 * abs(lref), (float(tot.detach()), lref)
  worst = 0.0
  for n, p in net.named_parameters():
    gn, gs, g0 = g["net5g_grad/" + n]
    gd = p.grad.detach().double()
    rel = abs(float(gd.norm()) - gn) / max(gn, 1e-12)
    assert rel <= 1e-3 or abs(float(gd.norm()) - gn) <= 1e-9, (n, float(gd.norm()), gn)
    rms = gn / max(gd.numel() ** 0.5, 1)
    assert abs(float(gd.flatten()[0]) - g0) <= 3e-2 * max(rms, abs(g0)) + 1e-9, (n, float(gd.flatten()[0]), g0)
    assert abs(float(gd.sum()) - gs) <= 3e-2 * (abs(gs) + gn), (n, float(gd.sum()), gs)
    worst = max(worst, rel)
  # ... and the gradients THEMSELVES (tests/golden/net5g_grads.npz, oracle/gen_golden_grads.py: whole tensors up to
  # 40960 elements, 8192 evenly spaced elements of the larger ones), not only their norms: relative L2 error of the
  # stored elements <= 1e-2 per parameter, median <= 4e-3 (fp32 summation order is all that differs)
  gg = np.load(os.path.join(G, "net5g_grads.npz"))
  worst_el, n_el, errs = 0.0, 0, []
  os.makedirs("gpurun_out", exist_ok=True)
  for n, p in net.named_parameters():
    ref = gg["grad/" + n].astype(np.float64)
    gd = p.grad.detach().double().cpu().numpy().reshape(-1)
    if gd.size > 40960:
      gd = gd[(np.arange(8192, dtype=np.int64) * gd.size) // 8192]
    assert gd.shape == ref.shape, (n, gd.shape, ref.shape)
    err = np.linalg.norm(gd - ref) / max(np.linalg.norm(ref), 1e-30)
    if np.linalg.norm(gd - ref) <= 1e-9:
      err = 0.0
    errs.append((err, n))
    worst_el, n_el = max(worst_el, err), n_el + ref.size
  errs.sort(reverse=True)
  with open("gpurun_out/dedup_grads.txt", "w") as f:
    f.write("".join("%.3e  %s\n" % e for e in errs))
  # measured (gpurun_out/dedup_grads.txt): median 2.5e-3, worst 5.1e-3 (layer1.1.bn2.weight) -- the loss of this fixture
  # sits at MI ~ 0, where the gradient is a difference of nearly equal terms and fp32 summation order (GPU kernels vs
  # the reference's CPU BLAS) shows at the 1e-3 level element by element while the norms agree to 1e-3
  assert errs[0][0] <= 1e-2 and errs[len(errs) // 2][0] <= 4e-3, errs[:5]
  sd = net.state_dict()
  assert np.abs(sd["trunk.bn1.running_mean"].cpu().numpy() - g["net5g_rm_bn1"]).max() <= 1e-5
  assert np.allclose(sd["trunk.bn1.running_var"].cpu().numpy(), g["net5g_rv_bn1"], rtol=1e-4, atol=1e-7)
  assert np.allclose(sd["trunk.layer4.2.bn2.running_var"].cpu().numpy(), g["net5g_rv_l4"], rtol=1e-3, atol=1e-7)
  os.makedirs("gpurun_out", exist_ok=True)
  with open("gpurun_out/dedup_fp32_mode.txt", "w") as f:
    f.write("max|dprob| %.3e, loss %.9f vs %.9f, worst grad-norm rel err %.3e over %d parameters; "
            "worst relative L2 error of the gradient elements %.3e over %d stored values\n"
            % (np.abs(out - g["net5g_out"]).max(), float(tot.detach()), lref, worst,
               len(list(net.named_parameters())), worst_el, n_el))


def test_north_star_full_size_properties():
  """BASELINE.json configs[1] at FULL size (660 pairs, 96x96, k=70, 5 sub-heads), checked
  through size-independent properties of the domain:
    * every sub-head row is a distribution (sums to 1, >= 0);
    * all_imgs = 220 base images replicated 3x (cluster_sobel.py:215-226): BN batch statistics
      are invariant to exact replication and every kernel treats rows independently, so the
      replicas' softmax rows must be IDENTICAL (bit-exact);
    * loss_no_lamb == loss at lamb = 1; the loss is invariant to permuting the pairs;
    * the IID loss evaluated by the float64 oracle on OUR softmax outputs matches OUR loss to
      the north-star clause (1e-5 relative + 2e-7);
    * a 2-way split of the batch reproduces the loss through the raw-joint sum (the DP algebra);
    * one Adam step changes every parameter and keeps everything finite."""
  from iic_amd import archs
  from iic_amd.losses import IID_loss_heads
  from iic_amd.optim import Adam
  from iic_amd.transforms import sobel_process
  from oracle import iid_oracle
  torch.manual_seed(0)
  H, K, NP = 5, 70, 660
  net = archs.ClusterNet5g(_cfg(input_sz=96, num_sub_heads=H, output_k=K)).to(dev()).train()
  with torch.no_grad():   # un-trivial heads so the loss is not ~0
    for h in net.head.heads:
      h[0].weight.normal_(0, 0.3)
  opt = Adam(net.parameters(), lr=1e-4)
  g = torch.Generator().manual_seed(5)
  base = torch.rand(NP // 3, 1, 96, 96, generator=g)
  base = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(base, (2, 2, 2, 2), mode="replicate"), 5, 1)
  imgs = base.repeat(3, 1, 1, 1)
  gain = torch.rand(NP, 1, 1, 1, generator=g) * 0.8 + 0.6
  imgs_tf = torch.clamp(torch.flip(imgs, dims=[3]) * gain + 0.05 * torch.randn(imgs.shape, generator=g), 0, 1)
  imgs, imgs_tf = imgs.to(dev()), imgs_tf.to(dev())
  before = [p.detach().clone() for p in net.parameters()]
  net.zero_grad()
  xo = net.forward_packed(sobel_process(imgs, False))       # [660, 5, 70]
  xt = net.forward_packed(sobel_process(imgs_tf, False))
  assert xo.shape == (NP, H, K) and xt.shape == (NP, H, K)
  rows = xo.detach()
  assert float(rows.min()) >= 0 and torch.allclose(rows.sum(2), torch.ones(NP, H, device=dev()), atol=1e-5)
  rep = xo.detach().reshape(3, NP // 3, H * K)
  assert torch.equal(rep[0], rep[1]) and torch.equal(rep[0], rep[2])
  loss_h, loss_nl_h = IID_loss_heads(xo, xt, lamb=1.0)
  assert torch.equal(loss_h, loss_nl_h)
  z = xo.detach().double().cpu().numpy()
  zt = xt.detach().double().cpu().numpy()
  perm = torch.randperm(NP, generator=g).to(dev())
  with torch.no_grad():
    loss_p, _ = IID_loss_heads(xo.detach()[perm].contiguous(), xt.detach()[perm].contiguous(), lamb=1.0)
  for h in range(H):
    ref = iid_oracle.iid_loss_np(z[:, h], zt[:, h], 1.0)[0]
    assert abs(float(loss_h[h]) - ref) <= 1e-5 * abs(ref) + 2e-7, (h, float(loss_h[h]), ref)
    assert abs(float(loss_p[h]) - ref) <= 1e-5 * abs(ref) + 2e-7
    R = iid_oracle.raw_joint_np(z[:330, h], zt[:330, h]) + iid_oracle.raw_joint_np(z[330:, h], zt[330:, h])
    assert abs(iid_oracle.loss_and_grad_from_raw_np(R, 1.0)[0] - ref) < 1e-12
  loss = loss_h.mean()
  loss.backward()
  assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
  opt.step()
  assert np.isfinite(loss.item()) and loss.item() <= 1e-6   # loss = -MI <= 0
  changed = [bool((p.detach() != b).any()) for p, b in zip(net.parameters(), before)]
  assert all(changed) and all(bool(torch.isfinite(p).all()) for p in net.parameters())


@pytest.mark.parametrize("layer,bidx,cin,planes,stride,H", [
  (1, 0, 64, 64, 1, 17), (2, 0, 64, 128, 2, 17), (3, 1, 256, 256, 1, 5), (4, 0, 256, 512, 2, 13)])
def test_basic_block_teacher_forced(layer, bidx, cin, planes, stride, H):
  """One BasicBlock Function (forward + backward) against the bf16-emulating oracle on the
  SAME input / upstream gradient: no error accumulation across layers, so tolerances are
  tight (bf16 output rounding + accumulation order)."""
  import torch.nn.functional as F
  from iic_amd import ops
  from iic_amd.archs.cluster import BasicBlock
  from oracle import net_oracle
  import torch.nn as nn
  N = 8
  rng = np.random.default_rng(layer * 10 + bidx)
  pre = "trunk.layer%d.%d" % (layer, bidx)
  full = net_oracle.make_net5g_params(2, 10, 2, True, seed=3, randomize_bn=True)
  params = {k: v.clone() for k, v in full.items() if k.startswith(pre + ".")}
  ds = None
  if (pre + ".downsample.0.weight") in params:
    ds = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
  blk = BasicBlock(cin, planes, stride, ds, track_running_stats=True)
  blk.load_state_dict({k[len(pre) + 1:]: v for k, v in params.items()}, strict=True)
  blk.to(dev()).train()
  x = torch.from_numpy(rng.standard_normal((N, cin, H, H)).astype(np.float32)).relu()
  x = x.to(torch.bfloat16).float()
  Ho = (H + 2 - 3) // stride + 1
  dout = torch.from_numpy(rng.standard_normal((N, planes, Ho, Ho)).astype(np.float32)).to(torch.bfloat16).float()
  # oracle
  for k, v in params.items():
    if v.dtype.is_floating_point and "running" not in k:
      v.requires_grad_(True)
  xe = x.clone().requires_grad_(True)
  oe = net_oracle.block_bf16emu(params, pre, xe, stride, True)
  oe.backward(dout)
  # HIP
  xp = ops.pt_from_nchw(x.to(dev()), 1).requires_grad_(True)
  o = blk(xp)
  o.backward(ops.pt_from_nchw(dout.to(dev()), 1))
  torch.cuda.synchronize()
  got = ops.pt_to_nchw(o.detach(), 1).cpu()
  scale = float(oe.abs().max())
  assert float((got - oe.detach()).abs().max()) <= 2e-2 * scale
  assert float((got - oe.detach()).abs().mean()) <= 2e-3 * scale
  gx = ops.pt_to_nchw(xp.grad, 1).cpu()
  assert _cos(gx, xe.grad) >= 0.999 and abs(float(gx.norm() / xe.grad.norm()) - 1) < 2e-2
  for n, p in blk.named_parameters():
    ref = params[pre + "." + n].grad
    c = _cos(p.grad.cpu(), ref)
    r = float(p.grad.norm().cpu() / ref.norm())
    assert c >= 0.998 and abs(r - 1) < 2e-2, (n, c, r)
  assert torch.allclose(blk.bn1.running_mean.cpu(), params[pre + ".bn1.running_mean"], atol=1e-3)
  assert torch.allclose(blk.bn2.running_var.cpu(), params[pre + ".bn2.running_var"], rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("layer,nblk,cin,planes,stride,H", [
  (1, 3, 64, 64, 1, 17), (2, 4, 64, 128, 2, 17), (3, 6, 128, 256, 2, 9), (4, 3, 256, 512, 2, 9)])
def test_residual_layer_teacher_forced(layer, nblk, cin, planes, stride, H):
  """A whole residual LAYER (3-6 BasicBlocks through the trunk's own pre-masked gradient chain)
  against the bf16-emulating oracle on the same input / upstream gradient: the tight tier between
  the single-block test and the whole (chaotic) net -- nothing but accumulation order separates
  the two sides, over up to 12 convolutions and 13 BatchNorms."""
  from iic_amd import ops
  from iic_amd.archs import cluster as cl
  from oracle import net_oracle
  N = 16
  rng = np.random.default_rng(100 + layer)
  full = net_oracle.make_net5g_params(2, 10, 2, True, seed=3, randomize_bn=True)
  cfg = _cfg(input_sz=32, num_sub_heads=2, output_k=10)
  from iic_amd import archs
  net = archs.ClusterNet5g(cfg)
  net.load_state_dict(full, strict=True)
  net.to(dev()).train()
  blocks = list(getattr(net.trunk, "layer%d" % layer))
  assert len(blocks) == nblk
  pres = ["trunk.layer%d.%d" % (layer, i) for i in range(nblk)]
  params = {k: v.clone() for k, v in full.items() if any(k.startswith(p + ".") for p in pres)}
  for k, v in params.items():
    if v.dtype.is_floating_point and "running" not in k:
      v.requires_grad_(True)
  x = torch.from_numpy(rng.standard_normal((N, cin, H, H)).astype(np.float32)).relu().to(torch.bfloat16).float()
  Ho = (H + 2 - 3) // stride + 1
  dout = torch.from_numpy(rng.standard_normal((N, planes, Ho, Ho)).astype(np.float32)).to(torch.bfloat16).float()
  # oracle: blocks in sequence, each storing its output in bf16
  xe = x.clone().requires_grad_(True)
  cur = xe
  for i, pre in enumerate(pres):
    cur = net_oracle.block_bf16emu(params, pre, cur, stride if i == 0 else 1, True)
  oe = cur
  # the consumer of the layer output masks the gradient it hands back (PREMASK contract)
  oe.backward(dout * (oe.detach() > 0).float())
  # HIP: same blocks, pre-masked chain as the trunk's forward sets it up
  for i, b in enumerate(blocks):
    b._dout_premasked = True
    b._mask_dx = i > 0
  try:
    xp = ops.pt_from_nchw(x.to(dev()), 1).requires_grad_(True)
    cur = xp
    link = cl._Chain()        # as the trunk's forward: fused BatchNorm-backward reductions
    for b in blocks:
      cur = b(cur, link)
    got = ops.pt_to_nchw(cur.detach(), 1).cpu()
    gmask = (got > 0).float()
    cur.backward(ops.pt_from_nchw((dout * gmask).to(dev()), 1))
  finally:
    for b in blocks:
      b._dout_premasked = b._mask_dx = False
  torch.cuda.synchronize()
  scale = float(oe.abs().max())
  err = (got - oe.detach()).abs()
  # a flipped ReLU / rounding boundary a few blocks up moves isolated elements by a few bf16 ulps
  assert float(err.max()) <= 6e-2 * scale, float(err.max()) / scale
  assert float(err.mean()) <= 2e-3 * scale, float(err.mean()) / scale
  gx = ops.pt_to_nchw(xp.grad, 1).cpu()
  assert _cos(gx, xe.grad) >= 0.99 and abs(float(gx.norm() / xe.grad.norm()) - 1) < 3e-2, _cos(gx, xe.grad)
  worst = 1.0
  for i, (b, pre) in enumerate(zip(blocks, pres)):
    for n, p in b.named_parameters():
      ref = params[pre + "." + n].grad
      c = _cos(p.grad.cpu(), ref)
      r = float(p.grad.norm().cpu() / ref.norm())
      worst = min(worst, c)
      assert c >= 0.99 and abs(r - 1) < 4e-2, (pre, n, c, r)


def test_premasked_gradient_chain_matches_self_masking_blocks():
  """archs.cluster.PREMASK: blocks that receive their output gradient already multiplied by the
  ReLU mask (applied by the consumer's backward-data epilogue / the average-pool backward) against
  the same blocks masking for themselves -- a plain block, a stride-2 downsample block and another
  plain block, then the average pool.  The masked values are identical bit for bit; sums differ by
  fp32 accumulation order only."""
  from iic_amd import ops
  from iic_amd.archs import cluster as cl
  torch.manual_seed(3)
  d = dev()
  N, H = 6, 14
  ds = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, 2, bias=False),
                           torch.nn.BatchNorm2d(128, track_running_stats=True))
  blocks = [cl.BasicBlock(64, 64, track_running_stats=True),
            cl.BasicBlock(64, 128, 2, ds, track_running_stats=True),
            cl.BasicBlock(128, 128, track_running_stats=True)]
  for b in blocks:
    b.to(d).train()
    for m in b.modules():
      if isinstance(m, torch.nn.BatchNorm2d):
        m.weight.data.uniform_(0.5, 1.5)
        m.bias.data.normal_(0, 0.3)
  x0 = torch.relu(torch.randn(N, 64, H, H))
  dfe = torch.randn(N, 128)
  res = {}
  for chain in (False, True):
    for b in blocks:
      b.zero_grad()
    x = ops.pt_from_nchw(x0.to(d), 1).requires_grad_(True)
    for i, b in enumerate(blocks):
      b._dout_premasked, b._mask_dx = chain, chain and i > 0
    try:
      h = x
      for b in blocks:
        h = b(h)
      f = cl._AvgPoolFn.apply(h, chain)
    finally:
      for b in blocks:
        b._dout_premasked = b._mask_dx = False
    f.backward(dfe.to(d))
    torch.cuda.synchronize()
    res[chain] = (f.detach().clone(), x.grad.detach().float().clone(),
                  {n: p.grad.clone() for bi, b in enumerate(blocks) for n, p in
                   ((("%d.%s" % (bi, k)), v) for k, v in b.named_parameters())})
  f0, dx0, g0 = res[False]
  f1, dx1, g1 = res[True]
  assert torch.equal(f0, f1)
  # block 0 does not pre-mask its input gradient (stem convention): dx must agree to fp32-sum noise
  assert (dx0 - dx1).abs().max().item() <= 2e-2 * dx0.abs().max().item()
  assert _cos(dx0, dx1) > 0.9999
  for n in g0:
    assert _cos(g0[n], g1[n]) > 0.9999, n
    assert abs(float(g1[n].norm() / g0[n].norm()) - 1) < 2e-3, n


def test_net5g_fp32_mode_vs_reference_golden():
  """SURVEY.md §8c parity tier T2: the WHOLE ClusterNet5g train step (sobel -> two train-mode
  forwards -> IID_loss x sub-heads -> backward) with the library's host orchestration running on
  the exact-fp32 kernels (`ops.fp32_mode()`, csrc/f32_path.hip) against the fp32 golden produced by
  the reference itself (tests/golden/nets.npz): nothing but fp32 summation order separates the two,
  so outputs, loss, running statistics and every parameter gradient are held tightly -- which the
  bf16 production path (rounding amplified by 33 batch-statistics BatchNorm layers) cannot be."""
  from iic_amd import archs, ops
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle
  g = np.load(os.path.join(G, "nets.npz"))
  params = net_oracle.make_net5g_params(2, 10, 2, True, seed=3, randomize_bn=True, head_std=0.3)
  net = archs.ClusterNet5g(_cfg(input_sz=32, num_sub_heads=2, output_k=10))
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  imgs, imgs_tf = net_oracle.make_paired_batch(24, 32, 3, seed=5)
  with ops.fp32_mode():
    xo = net(sobel_process(imgs.to(dev()), False))
    xt = net(sobel_process(imgs_tf.to(dev()), False))
  tot = None
  for i in range(2):
    l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
    tot = l if tot is None else tot + l
  tot = tot / 2
  tot.backward()                       # (outside the context: the Functions remember their mode)
  torch.cuda.synchronize()
  out = np.stack([o.detach().cpu().numpy() for o in xo])
  out_tf = np.stack([o.detach().cpu().numpy() for o in xt])
  assert np.abs(out - g["net5g_out"]).max() <= 2e-4, np.abs(out - g["net5g_out"]).max()
  assert np.abs(out_tf - g["net5g_out_tf"]).max() <= 2e-4, np.abs(out_tf - g["net5g_out_tf"]).max()
  lref = float(g["net5g_loss"][0])
  assert abs(float(tot.detach()) - lref) <= 2e-4 * abs(lref), (float(tot.detach()), lref)
  worst = 0.0
  for n, p in net.named_parameters():
    gn, gs, g0 = g["net5g_grad/" + n]
    gd = p.grad.detach().double()
    assert abs(float(gd.norm()) - gn) <= 5e-3 * max(gn, 1e-6) + 1e-9, (n, float(gd.norm()), gn)
    # single elements: the loss here is ~-0.015 (MI ~ 0), its gradient is a difference of nearly
    # equal terms and the fp32 reference itself carries ~1e-2 of element noise against float64
    rms = gn / max(gd.numel() ** 0.5, 1)
    assert abs(float(gd.flatten()[0]) - g0) <= 3e-2 * max(rms, abs(g0)) + 1e-9, (n, float(gd.flatten()[0]), g0)
    assert abs(float(gd.sum()) - gs) <= 3e-2 * (abs(gs) + gn), (n, float(gd.sum()), gs)
    worst = max(worst, abs(float(gd.norm()) - gn) / max(gn, 1e-12))
  sd = net.state_dict()
  assert np.abs(sd["trunk.bn1.running_mean"].cpu().numpy() - g["net5g_rm_bn1"]).max() <= 1e-5
  assert np.allclose(sd["trunk.bn1.running_var"].cpu().numpy(), g["net5g_rv_bn1"], rtol=1e-4, atol=1e-7)
  assert np.allclose(sd["trunk.layer4.2.bn2.running_var"].cpu().numpy(), g["net5g_rv_l4"], rtol=1e-3, atol=1e-7)
  os.makedirs("gpurun_out", exist_ok=True)
  with open("gpurun_out/net5g_fp32_mode.txt", "w") as f:
    f.write("max|dprob| %.3e / %.3e, loss %.9f vs %.9f, worst grad-norm rel err %.3e\n"
            % (np.abs(out - g["net5g_out"]).max(), np.abs(out_tf - g["net5g_out_tf"]).max(),
               float(tot.detach()), lref, worst))


def test_net5g_five_input_channels_fp32_mode_vs_oracle():
  """The CIFAR configuration of ClusterNet5g (examples/commands.txt:24,27: 32 x 32, `--include_rgb` => Sobel + RGB = 5 input
  channels, cluster_sobel_twohead.py): 9 * 5 = 45 > 32 stem taps take the two-pass stem backward.  Whole train step on the
  exact-fp32 kernels against the CPU restatement on the same parameters and batch: outputs, loss, gradient norms."""
  from iic_amd import archs, ops
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle, iid_oracle
  params = net_oracle.make_net5g_params(5, 10, 2, True, seed=21, randomize_bn=True, head_std=0.3)
  g = torch.Generator().manual_seed(22)
  rgb = torch.rand(12, 3, 32, 32, generator=g)
  rgb_tf = (rgb.flip(3) * (0.6 + 0.8 * torch.rand(12, 1, 1, 1, generator=g))).clamp(0, 1)
  def with_grey(t):                       # sobel_process(include_rgb=True) input: [rgb, grey] -> [rgb, dx, dy]
    return torch.cat([t, t.mean(1, keepdim=True)], dim=1)
  a_c, b_c = net_oracle.sobel_process(with_grey(rgb), True), net_oracle.sobel_process(with_grey(rgb_tf), True)
  assert a_c.shape[1] == 5
  rp = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
        for k, v in params.items()}
  ro = net_oracle.net5g_forward(rp, a_c, True, 32, "head", 2)
  rt = net_oracle.net5g_forward(rp, b_c, True, 32, "head", 2)
  rtot = sum(iid_oracle.IID_loss(ro[i], rt[i], lamb=1.0)[0] for i in range(2)) / 2
  rtot.backward()
  net = archs.ClusterNet5g(_cfg(in_channels=5))
  net.load_state_dict(params, strict=True)
  net.to(dev()).train()
  a = sobel_process(with_grey(rgb).to(dev()), True)
  b = sobel_process(with_grey(rgb_tf).to(dev()), True)
  assert (a.cpu() - a_c).abs().max().item() <= 1e-6
  with ops.fp32_mode():
    xo, xt = net(a), net(b)
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
  # bf16 path: runs, finite, outputs inside the bf16 tier
  net.zero_grad()
  bo, bt = net(a), net(b)
  (sum(IID_loss(bo[i], bt[i], lamb=1.0)[0] for i in range(2)) / 2).backward()
  torch.cuda.synchronize()
  for i in range(2):       # (a 33-BatchNorm net on 12 images amplifies bf16 rounding: robust aggregates, as for the 24-image fixture)
    d = (bo[i].detach().cpu() - ro[i].detach()).abs()
    assert d.mean().item() <= 3e-2, d.mean().item()
    assert (bo[i].detach().cpu().argmax(1) == ro[i].detach().argmax(1)).float().mean().item() >= 0.75
  assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_net5g_feature_flags_vs_oracle():
  """The three feature taps of ClusterNet5g.forward (net5g.py:95-103, used by the reference's k-means / semi-supervised
  tooling): `trunk_features` (512-d pooled features), `penultimate_features` (layer 3's output flattened in (c, h, w) order,
  layer 4 and the pool skipped), `kmeans_use_features` (the features repeated once per sub-head) -- on the exact-fp32
  kernels against the CPU restatement, train and eval mode, and the gradient through the penultimate tap."""
  from iic_amd import archs, ops
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle
  params = net_oracle.make_net5g_params(2, 10, 2, True, seed=31, randomize_bn=True, head_std=0.3)
  imgs, _ = net_oracle.make_paired_batch(12, 32, 3, seed=32)
  xc = net_oracle.sobel_process(imgs, False)
  net = archs.ClusterNet5g(_cfg())
  net.load_state_dict(params, strict=True)
  net.to(dev())
  x = sobel_process(imgs.to(dev()), False)
  for training in (True, False):
    net.train(training)
    want_f = net_oracle.net5g_trunk({k: v.clone() for k, v in params.items()}, xc, training, 32)
    want_p = net_oracle.net5g_trunk({k: v.clone() for k, v in params.items()}, xc, training, 32, penultimate_features=True)
    with torch.no_grad(), ops.fp32_mode():
      # (a training-mode forward updates the running statistics: reload so that both taps see the same state)
      net.load_state_dict(params, strict=True)
      f = net(x, trunk_features=True)
      net.load_state_dict(params, strict=True)
      p = net(x, trunk_features=True, penultimate_features=True)
      net.load_state_dict(params, strict=True)
      km = net(x, kmeans_use_features=True)
    assert f.shape == want_f.shape == (12, 512) and p.shape == want_p.shape
    assert (f.cpu() - want_f).abs().max().item() <= 2e-4 * max(1.0, want_f.abs().max().item())
    assert (p.cpu() - want_p).abs().max().item() <= 2e-4 * max(1.0, want_p.abs().max().item())
    assert len(km) == 2 and all(torch.equal(t, km[0]) for t in km) and (km[0].cpu() - want_f).abs().max().item() <= 2e-4 * max(1.0, want_f.abs().max().item())
  # gradient through the penultimate tap (training mode)
  net.train()
  net.load_state_dict(params, strict=True)
  rp = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
        for k, v in params.items()}
  wp = net_oracle.net5g_trunk(rp, xc, True, 32, penultimate_features=True)
  g = torch.Generator().manual_seed(33)
  up = torch.randn(wp.shape, generator=g)
  (wp * up).sum().backward()
  with ops.fp32_mode():
    p = net(x, trunk_features=True, penultimate_features=True)
  (p * up.to(dev())).sum().backward()
  torch.cuda.synchronize()
  for n, q in net.named_parameters():
    if n.startswith("trunk.layer4") or n.startswith("head"):
      assert q.grad is None or float(q.grad.abs().max()) == 0.0, n
      continue
    gn = float(rp[n].grad.double().norm())
    assert abs(float(q.grad.double().norm()) - gn) <= 1e-2 * max(gn, 1e-6) + 1e-9, (n, float(q.grad.double().norm()), gn)


@pytest.mark.parametrize("n_img,track,training", [(1, True, True), (2, True, False), (3, False, False), (5, False, True)])
def test_net5g_tiny_batches_and_batchnorm_modes_vs_oracle(n_img, track, training):
  """Corners of the BatchNorm semantics on whole-net forwards: one image per batch (statistics over its
  pixels only), eval mode on the running statistics, and `batchnorm_track` absent (track_running_stats=False: batch
  statistics even in eval()) -- exact-fp32 kernels against the CPU restatement on the same parameters."""
  from iic_amd import archs, ops
  from iic_amd.transforms import sobel_process
  from oracle import net_oracle
  params = net_oracle.make_net5g_params(2, 10, 2, track, seed=41, randomize_bn=True, head_std=0.3)
  g = torch.Generator().manual_seed(42 + n_img)
  imgs = torch.rand(n_img, 1, 32, 32, generator=g)
  xc = net_oracle.sobel_process(imgs, False)
  # the oracle's _batchnorm uses batch statistics whenever training or no running statistics exist
  want = net_oracle.net5g_forward({k: v.clone() for k, v in params.items()}, xc, training or not track, 32, "head", 2)
  net = archs.ClusterNet5g(_cfg(batchnorm_track=track))
  net.load_state_dict(params, strict=True)
  net.to(dev()).train(training)
  with torch.no_grad(), ops.fp32_mode():
    got = net(sobel_process(imgs.to(dev()), False))
  for a, b in zip(got, want):
    assert a.shape == b.shape == (n_img, 10)
    assert (a.cpu() - b).abs().max().item() <= 3e-4, (a.cpu() - b).abs().max().item()
