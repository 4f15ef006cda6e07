"""tests/golden/net5g_large.npz: one train step of the REFERENCE's own ClusterNet5g + IID_loss (imported
read-only from /root/reference, fp32, CPU) on a batch large enough to leave the chaotic small-batch
regime of batch-statistics BatchNorm: 96 images (32 base images x 3 replicas, cluster_sobel.py:215-226),
64 x 64, 2 sub-heads, k = 10 -- the whole-net fixture for the bf16 PRODUCTION kernels.

    python -m oracle.gen_golden_large          (build container only)

The trunk parameters come from a seed (net_oracle.make_net5g_params); a randomly initialised trunk maps
all images to nearly the same features, so random head weights give MI ~ 0 and a gradient that is pure
cancellation noise (loss 7e-8 with the small fixture's recipe).  The HEAD weights are therefore built from
the reference trunk's own features -- the ten leading principal directions of the two views' features,
whitened, rotated per sub-head, scaled by 3, bias = -W mean -- which gives peaked, view-correlated
soft-max outputs (loss ~ -0.4); they are small (2 x 10 x 513 floats) and stored in the fixture.  The second
view is a mild transform of the first (net_oracle.make_mild_pair): a random trunk is not flip-invariant.

Stored: the head parameters, both views' softmax outputs, the loss, per-parameter gradient norms, and the gradients themselves
-- complete for parameters up to 16 384 elements, a fixed strided sample of 16 384 elements for the large
convolution weights (stride = ceil(numel / 16384)) -- so that per-parameter cosines can be formed.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import net_oracle, ref_import  # noqa: E402
from oracle.gen_golden import expand_clones  # noqa: E402

N_PAIRS, INPUT_SZ, HEADS, K, SAMPLE = 96, 64, 2, 10, 16384


def sample_stride(numel):
  return max(1, -(-numel // SAMPLE))


def main():
  assert ref_import.available(), "reference tree not mounted"
  torch.manual_seed(0)
  sob = ref_import.ref_sobel_process()
  archs = ref_import.ref_cluster_archs()
  ref_loss = ref_import.ref_cluster_losses()
  cfg = types.SimpleNamespace(in_channels=2, input_sz=INPUT_SZ, batchnorm_track=True, num_sub_heads=HEADS, output_k=K)
  params = net_oracle.make_net5g_params(2, K, HEADS, True, seed=13, randomize_bn=True, head_std=0.03)
  net = archs["net5g"].ClusterNet5g(cfg)
  net.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
  net.train()
  imgs, imgs_tf = net_oracle.make_mild_pair(N_PAIRS, INPUT_SZ, 3, seed=21)
  a, b = sob(imgs, False), sob(imgs_tf, False)
  head_params = {}
  with torch.no_grad():
    momenta = [m.momentum for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in net.modules():                         # the probing forwards must not move the running statistics
      if isinstance(m, torch.nn.BatchNorm2d):
        m.momentum = 0.0
    fa, fb = net.trunk(a), net.trunk(b)
    for m, mo in zip([m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)], momenta):
      m.momentum = mo
      m.num_batches_tracked.zero_()
    both = torch.cat([fa, fb])
    mu = both.mean(0)
    _, S, Vh = torch.linalg.svd(both - mu, full_matrices=False)
    for h in range(HEADS):
      R = torch.from_numpy(np.linalg.qr(np.random.default_rng(100 + h).standard_normal((K, K)))[0].astype(np.float32))
      W = 3.0 * (R @ (Vh[:K] / (S[:K, None] / np.sqrt(both.shape[0]))))
      head_params["head.heads.%d.0.weight" % h] = W.contiguous()
      head_params["head.heads.%d.0.bias" % h] = (-(W @ mu)).contiguous()
  sd0 = net.state_dict()
  for k_, v_ in head_params.items():
    assert sd0[k_].shape == v_.shape, (k_, sd0[k_].shape, v_.shape)
  net.load_state_dict(head_params, strict=False)
  xo, xt = net(a), net(b)
  with expand_clones():
    tot = None
    for i in range(HEADS):
      l, _ = ref_loss.IID_loss(xo[i], xt[i], lamb=1.0)
      tot = l if tot is None else tot + l
    tot = tot / HEADS
    tot.backward()
  out = {"out": np.stack([o.detach().numpy() for o in xo]), "out_tf": np.stack([o.detach().numpy() for o in xt]),
         "loss": np.array([float(tot)])}
  for k_, v_ in head_params.items():
    out["param/" + k_] = v_.numpy()
  for n, p in net.named_parameters():
    g = p.grad.detach().flatten()
    out["gnorm/" + n] = np.array([float(g.double().norm())])
    out["grad/" + n] = g[::sample_stride(g.numel())].numpy().astype(np.float32)
  sd = net.state_dict()
  for k in ("trunk.bn1.running_mean", "trunk.bn1.running_var", "trunk.layer4.2.bn2.running_var"):
    out["state/" + k] = sd[k].numpy()
  path = os.path.join(ROOT, "tests", "golden", "net5g_large.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, os.path.getsize(path), "bytes; loss", float(tot))


if __name__ == "__main__":
  main()
