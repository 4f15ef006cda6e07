"""Golden PARAMETER GRADIENTS of the reference's own ClusterNet5g + IID_loss on the replicated 24-image batch of
tests/golden/nets.npz (nets.npz keeps norm / sum / first element per parameter only).

Same network, inputs and code path as oracle/gen_golden.py::gen_nets (the unmodified reference modules read from
/root/reference; IID_loss run with Tensor.expand materialised, see there).  Stored per parameter: the whole gradient
when it has <= 40960 elements (stem, every BatchNorm, heads, the 64-channel convolutions), otherwise the 8192 elements
at flat indices (arange(8192) * numel) // 8192 -- 116 parameters, 21.3 M gradient values sampled down to 0.5 M so that
the fixture stays a small file.  Run in the build container only:   python oracle/gen_golden_grads.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import net_oracle, ref_import  # noqa: E402
from oracle.gen_golden import _load, expand_clones  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
FULL_BELOW, SAMPLE = 40960, 8192


def sample_index(numel):
  return None if numel <= FULL_BELOW else (np.arange(SAMPLE, dtype=np.int64) * numel) // SAMPLE


def main():
  torch.manual_seed(0)
  sob = ref_import.ref_sobel_process()
  archs = ref_import.ref_cluster_archs()
  cfg = types.SimpleNamespace(in_channels=2, input_sz=32, batchnorm_track=True, num_sub_heads=2, output_k=10)
  params = net_oracle.make_net5g_params(2, 10, 2, True, seed=3, randomize_bn=True, head_std=0.3)
  net = archs["net5g"].ClusterNet5g(cfg)
  _load(net, params)
  net.train()
  imgs, imgs_tf = net_oracle.make_paired_batch(24, 32, 3, seed=5)
  ref_loss = ref_import.ref_cluster_losses()
  xo, xt = net(sob(imgs, False)), net(sob(imgs_tf, False))
  with expand_clones():
    tot = None
    for i in range(2):
      l, _ = ref_loss.IID_loss(xo[i], xt[i], lamb=1.0)
      tot = l if tot is None else tot + l
    tot = tot / 2
    tot.backward()
  old = np.load(os.path.join(OUT, "nets.npz"))
  assert float(tot) == float(old["net5g_loss"][0]), "not the computation of nets.npz"
  out = {"loss": np.array([float(tot)])}
  n_vals = 0
  for n, p in net.named_parameters():
    g = p.grad.detach().numpy().reshape(-1)
    idx = sample_index(g.size)
    out["grad/" + n] = g.copy() if idx is None else g[idx].copy()
    n_vals += out["grad/" + n].size
  np.savez_compressed(os.path.join(OUT, "net5g_grads.npz"), **out)
  print("net5g_grads.npz written: %d parameters, %d gradient values" % (len(out) - 1, n_vals))


if __name__ == "__main__":
  main()
