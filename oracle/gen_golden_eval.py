"""Golden vectors for the evaluation matching functions, produced by the reference's own
eval_metrics.py (run in THIS container, see oracle/ref_import.py::ref_eval_metrics).
  python -m oracle.gen_golden_eval   ->  tests/golden/eval.npz"""
import os

import numpy as np
import torch

from . import ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = [  # n, preds_k, targets_k, seed, correlated
  (1000, 10, 10, 0, True),
  (5000, 70, 10, 1, True),      # STL10 overclustering head vs 10 classes
  (3000, 140, 10, 2, True),
  (4000, 10, 10, 3, False),     # uncorrelated labels: near-ties
  (257, 3, 3, 4, True),
  (20000, 50, 10, 5, True),     # MNIST head A
]


def make_case(n, kp, kt, seed, correlated):
  rng = np.random.default_rng(seed)
  t = rng.integers(0, kt, n)
  if correlated:
    # every class owns a few clusters; 20 % label noise
    owner = rng.permutation(kp) % kt
    clusters_of = [np.nonzero(owner == c)[0] for c in range(kt)]
    p = np.array([rng.choice(clusters_of[c]) if len(clusters_of[c]) else rng.integers(0, kp) for c in t])
    noise = rng.random(n) < 0.2
    p[noise] = rng.integers(0, kp, int(noise.sum()))
  else:
    p = rng.integers(0, kp, n)
  return p.astype(np.int64), t.astype(np.int64)


def main():
  assert ref_import.available()
  ref = ref_import.ref_eval_metrics()
  out = {}
  for i, (n, kp, kt, seed, corr) in enumerate(CASES):
    p, t = make_case(n, kp, kt, seed, corr)
    tp, tt = torch.from_numpy(p), torch.from_numpy(t)
    om = ref._original_match(tp, tt, kp, kt)
    out["c%d/preds" % i], out["c%d/targets" % i] = p, t
    out["c%d/k" % i] = np.array([kp, kt])
    out["c%d/original_match" % i] = np.array(om, dtype=np.int64)
    if kp == kt:
      hm = ref._hungarian_match(tp, tt, kp, kt)
      out["c%d/num_correct" % i] = (n - ref._recorded["cost"]).astype(np.int64)
      out["c%d/hungarian_match" % i] = np.array(hm, dtype=np.int64)
      # reorder predictions like cluster_eval.py:213-227 and score them with the reference's _acc
      re = torch.zeros_like(tp)
      for o, g in hm:
        re[tp == int(o)] = int(g)
      out["c%d/acc" % i] = np.array([ref._acc(re, tt, kt)])
  np.savez_compressed(os.path.join(OUT, "eval.npz"), **out)
  print("wrote eval.npz with %d arrays" % len(out))


if __name__ == "__main__":
  main()
