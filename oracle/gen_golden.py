"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (imported read-only
from /root/reference through oracle/ref_import.py) on seeded inputs.

Run in the build container only:   python -m oracle.gen_golden
The outputs are small and committed; tests/test_oracle_golden.py pins the oracle
restatement (oracle/iid_oracle.py, oracle/net_oracle.py) against them, and the
``-m gpu`` tests pin the HIP path against the same files.

The reference's IID_loss cannot back-propagate on torch>=1.x because it writes in
place into an ``expand``-ed view (IID_losses.py:12-19).  For the gradient fixtures
we run the *unmodified reference function* with ``torch.Tensor.expand`` patched to
return a materialised copy (values identical) for the duration of the call.
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import iid_oracle, net_oracle, ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


@contextlib.contextmanager
def expand_clones():
  orig = torch.Tensor.expand
  torch.Tensor.expand = lambda self, *a, **k: orig(self, *a, **k).clone()
  try:
    yield
  finally:
    torch.Tensor.expand = orig


IID_CASES = [  # (bn, k, kind, lamb, seed)
  (12, 5, "trained", 1.0, 0),
  (37, 10, "trained", 1.5, 1),
  (64, 10, "init", 1.0, 2),
  (100, 50, "trained", 1.0, 3),
  (660, 70, "trained", 1.0, 4),
  (660, 70, "init", 1.5, 5),
  (700, 10, "trained", 1.0, 6),
  (231, 140, "trained", 1.0, 7),
  (50, 10, "onehot", 1.0, 8),
]


def gen_iid():
  ref = ref_import.ref_cluster_losses()
  out = {"cases": np.array(IID_CASES, dtype=object)}
  for ci, (bn, k, kind, lamb, seed) in enumerate(IID_CASES):
    z, zt = iid_oracle.make_softmax_pair(bn, k, kind, seed)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
      a = torch.from_numpy(z).to(dt).requires_grad_(True)
      b = torch.from_numpy(zt).to(dt).requires_grad_(True)
      # forward: unmodified reference
      with torch.no_grad():
        l0, l0n = ref.IID_loss(a.detach(), b.detach(), lamb=lamb)
      with expand_clones():
        l, ln = ref.IID_loss(a, b, lamb=lamb)
        l.backward()
      assert abs(float(l) - float(l0)) <= 1e-6 * max(1.0, abs(float(l0)))
      out["c%d_loss_%s" % (ci, tag)] = np.array([float(l0), float(l0n)], dtype=np.float64)
      out["c%d_dz_%s" % (ci, tag)] = a.grad.numpy().copy()
      out["c%d_dzt_%s" % (ci, tag)] = b.grad.numpy().copy()
  np.savez_compressed(os.path.join(OUT, "iid_loss.npz"), **out)
  print("iid_loss.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:6]})


SEG_CASES = [  # (bn, k, h, w, T, lamb, flip_frac, mask_p, seed)
  (2, 3, 12, 12, 1, 1.0, 0.0, 1.0, 0),
  (3, 4, 16, 12, 2, 1.5, 0.5, 0.7, 1),
  (2, 6, 20, 20, 3, 1.0, 1.0, 0.8, 2),
]


def make_seg_inputs(bn, k, h, w, flip_frac, mask_p, seed):
  rng = np.random.default_rng(seed)
  x1 = rng.standard_normal((bn, k, h, w)) * 2.0
  x2 = x1[:, :, :, ::-1] * 0.7 + rng.standard_normal((bn, k, h, w))

  def sm(x):
    x = x - x.max(axis=1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=1, keepdims=True)
  x1, x2 = sm(x1).astype(np.float32), sm(x2).astype(np.float32)
  aff = np.zeros((bn, 2, 3), dtype=np.float32)
  aff[:, 0, 0] = 1.0
  aff[:, 1, 1] = 1.0
  nflip = int(round(flip_frac * bn))
  aff[:nflip, 0, 0] = -1.0  # x-flip (potsdam.py:189-202)
  mask = (rng.random((bn, h, w)) < mask_p).astype(np.float32)
  return x1, x2, aff, mask


def gen_seg():
  ref = ref_import.ref_seg_losses()
  out = {"cases": np.array(SEG_CASES, dtype=object)}
  for ci, (bn, k, h, w, T, lamb, ff, mp, seed) in enumerate(SEG_CASES):
    x1, x2, aff, mask = make_seg_inputs(bn, k, h, w, ff, mp, seed)
    for name, fn in (("unc", ref.IID_segmentation_loss_uncollapsed),
                     ("col", ref.IID_segmentation_loss)):
      for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        a = torch.from_numpy(x1).to(dt).requires_grad_(True)
        b = torch.from_numpy(x2).to(dt).requires_grad_(True)
        l, ln = fn(a, b, all_affine2_to_1=torch.from_numpy(aff).to(dt),
                   all_mask_img1=torch.from_numpy(mask).to(dt), lamb=lamb,
                   half_T_side_dense=T, half_T_side_sparse_min=0, half_T_side_sparse_max=0)
        l.backward()
        out["c%d_%s_loss_%s" % (ci, name, tag)] = np.array([float(l), float(ln)])
        out["c%d_%s_dx1_%s" % (ci, name, tag)] = a.grad.numpy().copy()
        out["c%d_%s_dx2_%s" % (ci, name, tag)] = b.grad.numpy().copy()
  np.savez_compressed(os.path.join(OUT, "iid_seg_loss.npz"), **out)
  print("iid_seg_loss.npz written")


def _load(module, params):
  sd = {k: v.clone() for k, v in params.items()}
  missing = module.load_state_dict(sd, strict=True)
  return missing


def _grad_summary(module):
  res = {}
  for n, p in module.named_parameters():
    g = p.grad.detach().double()
    res[n] = np.array([float(g.norm()), float(g.sum()), float(g.flatten()[0])])
  return res


def gen_nets():
  out = {}
  # ---- sobel
  sob = ref_import.ref_sobel_process()
  rng = np.random.default_rng(11)
  g1 = torch.from_numpy(rng.random((3, 1, 10, 12)).astype(np.float32))
  g4 = torch.from_numpy(rng.random((2, 4, 9, 9)).astype(np.float32))
  out["sobel_in1"], out["sobel_out1"] = g1.numpy(), sob(g1, False).numpy()
  out["sobel_in4"], out["sobel_out4"] = g4.numpy(), sob(g4, True).numpy()

  # ---- ClusterNet5g  (input 32, in_ch 2, k=10, 2 sub-heads, batch 6)
  archs = ref_import.ref_cluster_archs()
  cfg = types.SimpleNamespace(in_channels=2, input_sz=32, batchnorm_track=True,
                              num_sub_heads=2, output_k=10)
  params = net_oracle.make_net5g_params(2, 10, 2, True, seed=3, randomize_bn=True, head_std=0.3)
  net = archs["net5g"].ClusterNet5g(cfg)
  _load(net, params)
  net.train()
  imgs, imgs_tf = net_oracle.make_paired_batch(24, 32, 3, seed=5)
  a, b = sob(imgs, False), sob(imgs_tf, False)
  ref_loss = ref_import.ref_cluster_losses()
  xo, xt = net(a), net(b)
  with expand_clones():
    tot = None
    for i in range(2):
      l, _ = ref_loss.IID_loss(xo[i], xt[i], lamb=1.0)
      tot = l if tot is None else tot + l
    tot = tot / 2
    tot.backward()
  out["net5g_out"] = np.stack([o.detach().numpy() for o in xo])
  out["net5g_out_tf"] = np.stack([o.detach().numpy() for o in xt])
  out["net5g_loss"] = np.array([float(tot)])
  for n, v in _grad_summary(net).items():
    out["net5g_grad/" + n] = v
  sd = net.state_dict()
  out["net5g_rm_bn1"] = sd["trunk.bn1.running_mean"].numpy()
  out["net5g_rv_bn1"] = sd["trunk.bn1.running_var"].numpy()
  out["net5g_rv_l4"] = sd["trunk.layer4.2.bn2.running_var"].numpy()

  # ---- ClusterNet6c  (input 24, in_ch 1, k=10, 2 sub-heads, batch 6)
  cfg = types.SimpleNamespace(in_channels=1, input_sz=24, batchnorm_track=True,
                              num_sub_heads=2, output_k=10)
  params = net_oracle.make_net6c_params(1, 24, 10, 2, True, seed=4, randomize_bn=True, head_std=0.05)
  net = archs["net6c"].ClusterNet6c(cfg)
  _load(net, params)
  net.train()
  x6, x6t = net_oracle.make_paired_batch(24, 24, 3, seed=6)
  xo, xt = net(x6), net(x6t)
  with expand_clones():
    tot = None
    for i in range(2):
      l, _ = ref_loss.IID_loss(xo[i], xt[i], lamb=1.0)
      tot = l if tot is None else tot + l
    tot = tot / 2
    tot.backward()
  out["net6c_out"] = np.stack([o.detach().numpy() for o in xo])
  out["net6c_loss"] = np.array([float(tot)])
  for n, v in _grad_summary(net).items():
    out["net6c_grad/" + n] = v

  # ---- SegmentationNet10a (input 24, in_ch 4, k=3, 1 sub-head, batch 2)
  sarchs = ref_import.ref_seg_archs()
  cfg = types.SimpleNamespace(in_channels=4, input_sz=24, batchnorm_track=True,
                              num_sub_heads=1, output_k=3)
  params = net_oracle.make_net10a_params(4, 3, 1, True, seed=5, randomize_bn=True)
  net = sarchs["net10a"].SegmentationNet10a(cfg)
  _load(net, params)
  net.train()
  rng = np.random.default_rng(12)
  xs = torch.from_numpy(rng.random((2, 4, 24, 24)).astype(np.float32))
  ys = net(xs)
  out["net10a_in"] = xs.numpy()
  out["net10a_out"] = ys[0].detach().numpy()
  np.savez_compressed(os.path.join(OUT, "nets.npz"), **out)
  print("nets.npz written:", len(out), "arrays")


if __name__ == "__main__":
  assert ref_import.available(), "reference tree not mounted"
  os.makedirs(OUT, exist_ok=True)
  torch.manual_seed(0)
  gen_iid()
  gen_seg()
  gen_nets()
