"""tests/golden/net5g_large_f64.npz: the gradients of the 96-image fixture (tests/golden/net5g_large.npz) evaluated in
FLOAT64 by the oracle's restatement of the reference network + loss (oracle/net_oracle.py: in float32 it reproduces the
reference's own gradients of that fixture BIT FOR BIT -- asserted below -- so its float64 run is the exact answer the
reference's float32 run approximates).

    python -m oracle.gen_golden_large_f64          (needs tests/golden/net5g_large.npz; CPU, about a minute)

Why: 33 batch-statistics BatchNorm layers amplify float32 rounding -- the reference's OWN float32 gradients differ from
the float64 ones by 4e-3 (median relative L2 per parameter; worst 7.5e-3) although loss and probabilities agree to 2e-6 /
1e-5.  An element-wise gate against the float32 golden therefore cannot go below that noise for ANY float32
implementation with another summation order (VERDICT r4 next #9 asked for 2e-4: not attainable, measured here).  The
tight gate that exists: the HIP fp32-mode path must be as close to the float64 gradients as the reference's float32 run is.

Stored per parameter: grad64/<name> -- every 4th element of net5g_large.npz's sample (whole tensors up to 4 096 elements),
float32 copies of the float64 values; ref_err/<name> -- relative L2 error of the reference's float32 gradient against
float64 on those elements.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import net_oracle  # noqa: E402
from oracle.gen_golden_large import HEADS, INPUT_SZ, K, N_PAIRS, sample_stride  # noqa: E402

SUB = 4      # every SUB-th element of the float32 fixture's sample


def _grads(params, imgs, imgs_tf, dtype):
  p = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in params.items()}
  for v in p.values():
    if v.dtype.is_floating_point:
      v.requires_grad_(True)
  for k in list(p):
    if "running" in k:
      p[k] = p[k].detach()
  loss = net_oracle.net5g_train_step_loss(p, imgs.to(dtype), imgs_tf.to(dtype), 1.0, INPUT_SZ, HEADS)[0]
  loss.backward()
  return float(loss.detach()), {k: v.grad.detach().double() for k, v in p.items() if v.requires_grad and v.grad is not None}


def main():
  g = np.load(os.path.join(ROOT, "tests", "golden", "net5g_large.npz"))
  params = net_oracle.make_net5g_params(2, K, HEADS, True, seed=13, randomize_bn=True, head_std=0.03)
  for k in g.files:
    if k.startswith("param/"):
      params[k[6:]] = torch.from_numpy(g[k])
  imgs, imgs_tf = net_oracle.make_mild_pair(N_PAIRS, INPUT_SZ, 3, seed=21)
  l32, g32 = _grads(params, imgs, imgs_tf, torch.float32)
  l64, g64 = _grads(params, imgs, imgs_tf, torch.float64)
  out = {"loss64": np.array([l64])}
  errs = []
  for n in g64:
    st = sample_stride(g64[n].numel())
    a32 = g32[n].flatten()[::st].numpy()
    gold = g["grad/" + n].astype(np.float64)
    # the oracle in float32 IS the reference's computation: same gradients, bit for bit
    assert np.array_equal(a32.astype(np.float32), g["grad/" + n]), n
    a64 = g64[n].flatten()[::st].numpy()[::SUB]
    out["grad64/" + n] = a64.astype(np.float32)
    e = float(np.linalg.norm(gold[::SUB] - a64) / max(np.linalg.norm(a64), 1e-30))
    out["ref_err/" + n] = np.array([e])
    errs.append(e)
  path = os.path.join(ROOT, "tests", "golden", "net5g_large_f64.npz")
  np.savez_compressed(path, **out)
  print("loss float32 %.9f float64 %.9f golden %.9f; reference float32 gradients vs float64: median %.3e worst %.3e "
        "(%d parameters) -> %s (%d bytes)" % (l32, l64, float(g["loss"][0]), float(np.median(errs)), max(errs), len(errs),
                                              path, os.path.getsize(path)))


if __name__ == "__main__":
  main()
