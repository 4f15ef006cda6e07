"""Pin the oracle restatement against the golden vectors produced by the reference
itself (oracle/gen_golden.py).  CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import iid_oracle, net_oracle
from oracle.gen_golden import IID_CASES, SEG_CASES, make_seg_inputs

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g_iid():
  return np.load(os.path.join(G, "iid_loss.npz"), allow_pickle=True)


@pytest.mark.parametrize("ci", range(len(IID_CASES)))
def test_iid_loss_oracle_matches_reference(g_iid, ci):
  bn, k, kind, lamb, seed = IID_CASES[ci]
  z, zt = iid_oracle.make_softmax_pair(bn, k, kind, seed)
  for tag, dt, tol in (("f32", torch.float32, 2e-6), ("f64", torch.float64, 1e-12)):
    a = torch.from_numpy(z).to(dt).requires_grad_(True)
    b = torch.from_numpy(zt).to(dt).requires_grad_(True)
    l, ln = iid_oracle.IID_loss(a, b, lamb=lamb)
    l.backward()
    ref = g_iid["c%d_loss_%s" % (ci, tag)]
    assert abs(float(l) - ref[0]) <= tol * max(1.0, abs(ref[0]))
    assert abs(float(ln) - ref[1]) <= tol * max(1.0, abs(ref[1]))
    gref = g_iid["c%d_dz_%s" % (ci, tag)]
    assert np.abs(a.grad.numpy() - gref).max() <= tol * max(1e-3, np.abs(gref).max()) * 10
    gref = g_iid["c%d_dzt_%s" % (ci, tag)]
    assert np.abs(b.grad.numpy() - gref).max() <= tol * max(1e-3, np.abs(gref).max()) * 10


@pytest.mark.parametrize("ci", range(len(IID_CASES)))
def test_iid_closed_form_matches_reference_f64(g_iid, ci):
  """The float64 closed form (what the HIP kernel implements) == reference autograd."""
  bn, k, kind, lamb, seed = IID_CASES[ci]
  z, zt = iid_oracle.make_softmax_pair(bn, k, kind, seed)
  loss, loss_nl, dz, dzt = iid_oracle.iid_loss_np(z, zt, lamb)
  ref = g_iid["c%d_loss_f64" % ci]
  assert abs(loss - ref[0]) <= 1e-11 * max(1.0, abs(ref[0]))
  assert abs(loss_nl - ref[1]) <= 1e-11 * max(1.0, abs(ref[1]))
  for mine, key in ((dz, "dz"), (dzt, "dzt")):
    gref = g_iid["c%d_%s_f64" % (ci, key)]
    assert np.abs(mine - gref).max() <= 1e-10 * max(1e-3, np.abs(gref).max())


def test_iid_analytic_pins():
  # z = z' = balanced one-hot => loss = -ln k ; uniform => 0  
  for k in (5, 10):
    z, zt = iid_oracle.make_softmax_pair(10 * k, k, "onehot", 0, np.float64)
    loss, loss_nl, _, _ = iid_oracle.iid_loss_np(z, zt, 1.0)
    assert abs(loss + math.log(k)) < 1e-9
    assert abs(loss - loss_nl) < 1e-15
    u = np.full((30, k), 1.0 / k)
    loss, _, _, _ = iid_oracle.iid_loss_np(u, u, 1.0)
    assert abs(loss) < 1e-12


@pytest.mark.parametrize("ci", range(len(SEG_CASES)))
def test_seg_loss_oracle_matches_reference(ci):
  g = np.load(os.path.join(G, "iid_seg_loss.npz"), allow_pickle=True)
  bn, k, h, w, T, lamb, ff, mp, seed = SEG_CASES[ci]
  x1, x2, aff, mask = make_seg_inputs(bn, k, h, w, ff, mp, seed)
  for name, fn in (("unc", iid_oracle.IID_segmentation_loss_uncollapsed),
                   ("col", iid_oracle.IID_segmentation_loss)):
    for tag, dt, tol in (("f32", torch.float32, 1e-5), ("f64", torch.float64, 1e-11)):
      a = torch.from_numpy(x1).to(dt).requires_grad_(True)
      b = torch.from_numpy(x2).to(dt).requires_grad_(True)
      l, ln = fn(a, b, all_affine2_to_1=torch.from_numpy(aff).to(dt),
                 all_mask_img1=torch.from_numpy(mask).to(dt), lamb=lamb, half_T_side_dense=T)
      l.backward()
      ref = g["c%d_%s_loss_%s" % (ci, name, tag)]
      assert abs(float(l) - ref[0]) <= tol * max(1.0, abs(ref[0]))
      assert abs(float(ln) - ref[1]) <= tol * max(1.0, abs(ref[1]))
      for t, key in ((a, "dx1"), (b, "dx2")):
        gref = g["c%d_%s_%s_%s" % (ci, name, key, tag)]
        assert np.abs(t.grad.numpy() - gref).max() <= 10 * tol * max(1e-6, np.abs(gref).max())


@pytest.fixture(scope="module")
def g_nets():
  return np.load(os.path.join(G, "nets.npz"))


def test_sobel_oracle(g_nets):
  o = net_oracle.sobel_process(torch.from_numpy(g_nets["sobel_in1"]), False)
  assert np.array_equal(o.numpy(), g_nets["sobel_out1"])
  o = net_oracle.sobel_process(torch.from_numpy(g_nets["sobel_in4"]), True)
  assert np.array_equal(o.numpy(), g_nets["sobel_out4"])


def _grads(params, loss):
  names = [n for n, v in params.items() if v.requires_grad]
  gs = torch.autograd.grad(loss, [params[n] for n in names])
  return dict(zip(names, gs))


def _req(params):
  for n, v in params.items():
    if v.dtype.is_floating_point and "running" not in n:
      v.requires_grad_(True)
  return params


def test_net5g_oracle_matches_reference(g_nets):
  params = _req(net_oracle.make_net5g_params(2, 10, 2, True, seed=3, randomize_bn=True, head_std=0.3))
  imgs, imgs_tf = net_oracle.make_paired_batch(24, 32, 3, seed=5)
  loss, _, xo, xt = net_oracle.net5g_train_step_loss(params, imgs, imgs_tf, 1.0, 32, 2)
  assert np.abs(np.stack([o.detach().numpy() for o in xo]) - g_nets["net5g_out"]).max() < 2e-6
  assert np.abs(np.stack([o.detach().numpy() for o in xt]) - g_nets["net5g_out_tf"]).max() < 2e-6
  assert abs(float(loss) - g_nets["net5g_loss"][0]) < 1e-6
  gs = _grads(params, loss)
  for n, gr in gs.items():
    ref = g_nets["net5g_grad/" + n]
    assert abs(float(gr.double().norm()) - ref[0]) <= 1e-3 * max(ref[0], 1e-6), n
  # the gradients themselves (oracle/gen_golden_grads.py: whole tensors up to 40960 elements, 8192 evenly spaced
  # elements of the larger ones)
  gg = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "net5g_grads.npz"))
  assert float(gg["loss"][0]) == float(g_nets["net5g_loss"][0])
  for n, gr in gs.items():
    ref = gg["grad/" + n].astype(np.float64)
    got = gr.detach().double().numpy().reshape(-1)
    if got.size > 40960:
      got = got[(np.arange(8192, dtype=np.int64) * got.size) // 8192]
    err = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    assert err <= 1e-3 or np.linalg.norm(got - ref) <= 1e-9, (n, err)
  assert np.allclose(params["trunk.bn1.running_mean"].numpy(), g_nets["net5g_rm_bn1"], atol=1e-6)
  assert np.allclose(params["trunk.bn1.running_var"].numpy(), g_nets["net5g_rv_bn1"], atol=1e-6)
  assert np.allclose(params["trunk.layer4.2.bn2.running_var"].numpy(), g_nets["net5g_rv_l4"],
                     rtol=1e-4)


def test_net6c_oracle_matches_reference(g_nets):
  params = _req(net_oracle.make_net6c_params(1, 24, 10, 2, True, seed=4, randomize_bn=True, head_std=0.05))
  x6, x6t = net_oracle.make_paired_batch(24, 24, 3, seed=6)
  xo = net_oracle.net6c_forward(params, x6, True, "head", 2)
  xt = net_oracle.net6c_forward(params, x6t, True, "head", 2)
  assert np.abs(np.stack([o.detach().numpy() for o in xo]) - g_nets["net6c_out"]).max() < 2e-6
  tot = sum(iid_oracle.IID_loss(xo[i], xt[i], 1.0)[0] for i in range(2)) / 2
  assert abs(float(tot) - g_nets["net6c_loss"][0]) < 1e-6
  gs = _grads(params, tot)
  for n, gr in gs.items():
    ref = g_nets["net6c_grad/" + n]
    assert abs(float(gr.double().norm()) - ref[0]) <= 1e-3 * max(ref[0], 1e-6), n


def test_net10a_oracle_matches_reference(g_nets):
  params = net_oracle.make_net10a_params(4, 3, 1, True, seed=5, randomize_bn=True)
  ys = net_oracle.net10a_forward(params, torch.from_numpy(g_nets["net10a_in"]), 24, True, "head", 1)
  assert np.abs(ys[0].detach().numpy() - g_nets["net10a_out"]).max() < 2e-6


def test_eval_metrics_oracle_matches_reference():
  """oracle/eval_oracle.py vs the reference's own _original_match / _hungarian_match / _acc
  (tests/golden/eval.npz, produced by oracle/gen_golden_eval.py)."""
  from oracle import eval_oracle
  g = np.load(os.path.join(G, "eval.npz"))
  n_cases = len([k for k in g.files if k.endswith("/k")])
  assert n_cases >= 6
  for i in range(n_cases):
    p, t = g["c%d/preds" % i], g["c%d/targets" % i]
    kp, kt = (int(v) for v in g["c%d/k" % i])
    assert eval_oracle.original_match(p, t, kp, kt) == [tuple(int(v) for v in r) for r in g["c%d/original_match" % i]]
    if kp == kt:
      c = eval_oracle.contingency(p, t, kp, kt)
      assert np.array_equal(c, g["c%d/num_correct" % i])
      hm = eval_oracle.hungarian_match(p, t, kp, kt)
      ref = [tuple(int(v) for v in r) for r in g["c%d/hungarian_match" % i]]
      assert sum(c[a, b] for a, b in hm) == sum(c[a, b] for a, b in ref)     # same optimum
      assert sorted(b for _, b in hm) == list(range(kt))                       # a permutation
      re = np.zeros_like(p)
      for a, b in hm:
        re[p == a] = b
      assert abs(eval_oracle.acc(re, t) - float(g["c%d/acc" % i][0])) < 1e-12


# ---- second segmentation fixture: BASELINE shapes, general affine, sparse shift ------------
from oracle.gen_golden_seg2 import SEG2_CASES, case_inputs  # noqa: E402


@pytest.mark.parametrize("case", SEG2_CASES, ids=[c[0] for c in SEG2_CASES])
def test_seg_loss_oracle_matches_reference_fixture2(case):
  g = np.load(os.path.join(G, "iid_seg_loss2.npz"))
  name, bn, k, h, w, T, lamb, ff, mp, seed, affine, smin, smax, np_seed = case
  x1, x2, aff, mask = case_inputs(case)
  for vname, fn in (("unc", iid_oracle.IID_segmentation_loss_uncollapsed),
                    ("col", iid_oracle.IID_segmentation_loss)):
    for tag, dt, tol in (("f32", torch.float32, 1e-5), ("f64", torch.float64, 1e-11)):
      a = torch.from_numpy(x1).to(dt).requires_grad_(True)
      b = torch.from_numpy(x2).to(dt).requires_grad_(True)
      np.random.seed(np_seed)
      l, ln = fn(a, b, all_affine2_to_1=torch.from_numpy(aff).to(dt),
                 all_mask_img1=torch.from_numpy(mask).to(dt), lamb=lamb, half_T_side_dense=T,
                 half_T_side_sparse_min=smin, half_T_side_sparse_max=smax)
      l.backward()
      ref = g["%s_%s_loss_%s" % (name, vname, tag)]
      assert abs(float(l) - ref[0]) <= tol * max(1.0, abs(ref[0]))
      assert abs(float(ln) - ref[1]) <= tol * max(1.0, abs(ref[1]))
      if tag == "f64":
        for t, key in ((a, "dx1"), (b, "dx2")):
          gref = g["%s_%s_%s" % (name, vname, key)]
          assert np.abs(t.grad.numpy() - gref).max() <= 1e-6 * max(1e-6, np.abs(gref).max())
