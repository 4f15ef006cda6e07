"""tests/golden/traj_{net5g,net6c}.json: loss TRAJECTORIES of the REFERENCE's own modules (imported read-only from
/root/reference, fp32, CPU) over identical synthetic steps -- SURVEY.md section 8c tier T3, end-to-end clause:
"loss trajectory over N identical synthetic steps [bf16] tracks the fp32 reference (same trend, final gap stated)".

    python -m oracle.gen_golden_traj          (build container only)

  * net5g: the 96-image fixture of tests/golden/net5g_large.npz (ClusterNet5g, 64 x 64, 2 sub-heads, k = 10, the head
    weights stored there, net_oracle.make_mild_pair seed 21; loss -0.403 at step 0), 30 steps of
    cluster_sobel.py:235-272 on the SAME batch: sobel x2 -> net(all_imgs), net(all_imgs_tf) -> IID_loss per sub-head
    -> mean -> backward -> torch.optim.Adam (general.py:5-9; lr 1e-5: at the scripts' 1e-4 this fixture's
    aligned head weights take the loss from -0.40 to -2.2 within five steps and the rest of the run is a plateau).
  * net6c: ClusterNet6c, 24 x 24 x 1, 120 pairs (net_oracle.make_mild_pair seed 5), 5 sub-heads, k = 10, the same loop,
    Adam(lr 1e-3), 30 steps.

Written at 1, 2 and 8 BLAS threads: a 30-step fp32 training run does not reproduce itself across summation orders
(tests/golden/script_cluster_sobel.json, LAB.md section R5.4), so the fixture carries the reference's OWN band and the
tests gate the HIP paths against it instead of pretending there is one trajectory.
"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import net_oracle, ref_import  # noqa: E402
from oracle.gen_golden import expand_clones  # noqa: E402

STEPS = 30
NET5G = dict(n_pairs=96, input_sz=64, heads=2, k=10, lr=1e-5, pair_seed=21)
NET6C = dict(n_pairs=120, input_sz=24, heads=5, k=10, lr=1e-3, pair_seed=5, param_seed=17)


def net5g_init():
  import numpy as np
  g = np.load(os.path.join(ROOT, "tests", "golden", "net5g_large.npz"))
  params = net_oracle.make_net5g_params(2, NET5G["k"], NET5G["heads"], True, seed=13, randomize_bn=True, head_std=0.03)
  for k in g.files:
    if k.startswith("param/"):
      params[k[6:]] = torch.from_numpy(g[k])
  return params


def net6c_init():
  return net_oracle.make_net6c_params(1, NET6C["input_sz"], output_k=NET6C["k"], num_sub_heads=NET6C["heads"],
                                      batchnorm_track=True, seed=NET6C["param_seed"])


def run(net, sob, loss_fn, imgs, imgs_tf, heads, lr):
  opt = torch.optim.Adam(net.parameters(), lr=lr)
  a, b = sob(imgs, False), sob(imgs_tf, False)
  losses = []
  for _ in range(STEPS):
    net.zero_grad()
    xo, xt = net(a), net(b)
    with expand_clones():
      tot = None
      for i in range(heads):
        l, _ = loss_fn(xo[i], xt[i], lamb=1.0)
        tot = l if tot is None else tot + l
      tot = tot / heads
      tot.backward()
    opt.step()
    losses.append(float(tot))
  return losses


def main():
  assert ref_import.available(), "reference tree not mounted"
  sob = ref_import.ref_sobel_process()
  archs = ref_import.ref_cluster_archs()
  loss_fn = ref_import.ref_cluster_losses().IID_loss
  out5, out6 = {"config": NET5G, "steps": STEPS, "threads": {}}, {"config": NET6C, "steps": STEPS, "threads": {}}
  for nt in (1, 2, 8):
    torch.set_num_threads(nt)
    cfg = types.SimpleNamespace(in_channels=2, input_sz=NET5G["input_sz"], batchnorm_track=True,
                                num_sub_heads=NET5G["heads"], output_k=NET5G["k"])
    net = archs["net5g"].ClusterNet5g(cfg)
    net.load_state_dict({k: v.clone() for k, v in net5g_init().items()}, strict=True)
    net.train()
    imgs, imgs_tf = net_oracle.make_mild_pair(NET5G["n_pairs"], NET5G["input_sz"], 3, seed=NET5G["pair_seed"])
    out5["threads"][str(nt)] = run(net, sob, loss_fn, imgs, imgs_tf, NET5G["heads"], NET5G["lr"])
    print("net5g", nt, ["%.4f" % v for v in out5["threads"][str(nt)][::5]], flush=True)
    cfg = types.SimpleNamespace(in_channels=1, input_sz=NET6C["input_sz"], batchnorm_track=True,
                                num_sub_heads=NET6C["heads"], output_k=NET6C["k"])
    net = archs["net6c"].ClusterNet6c(cfg)
    net.load_state_dict({k: v.clone() for k, v in net6c_init().items()}, strict=True)
    net.train()
    imgs, imgs_tf = net_oracle.make_mild_pair(NET6C["n_pairs"], NET6C["input_sz"], 3, seed=NET6C["pair_seed"])
    ident = lambda t, _rgb: t      # noqa: E731  (cluster_greyscale*.py feed the grey image itself: no sobel)
    out6["threads"][str(nt)] = run(net, ident, loss_fn, imgs, imgs_tf, NET6C["heads"], NET6C["lr"])
    print("net6c", nt, ["%.4f" % v for v in out6["threads"][str(nt)][::5]], flush=True)
  for name, o in (("traj_net5g.json", out5), ("traj_net6c.json", out6)):
    with open(os.path.join(ROOT, "tests", "golden", name), "w") as f:
      json.dump(o, f, indent=1)
    print("wrote", name)


if __name__ == "__main__":
  main()
