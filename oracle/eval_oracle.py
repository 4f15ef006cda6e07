"""CPU restatement (TEST INFRASTRUCTURE, never imported by iic_amd/) of the reference's
evaluation matching functions, /root/reference/code/utils/cluster/eval_metrics.py.

  contingency  : the counts both matchers build pair by pair (:18-24, :42-46)
  original_match  (:9-26)  many-to-one: every output cluster -> the ground-truth class it
                           overlaps most; first maximum in class order wins (strict '>')
  hungarian_match (:29-57) one-to-one assignment maximising the matched count; the reference
                           calls sklearn 0.19.1's linear_assignment (Hungarian algorithm, a
                           dependency absent here) on num_samples - num_correct; any optimal
                           assignment has the same total, ties may be broken differently
  acc             (:60-71)

Pinned by tests/golden/eval.npz (oracle/gen_golden_eval.py runs the reference's own functions)."""
import numpy as np
from scipy.optimize import linear_sum_assignment


def contingency(flat_preds, flat_targets, preds_k, targets_k):
  p = np.asarray(flat_preds).astype(np.int64)
  t = np.asarray(flat_targets).astype(np.int64)
  ok = (p >= 0) & (p < preds_k) & (t >= 0) & (t < targets_k)
  out = np.zeros((preds_k, targets_k), dtype=np.int64)
  np.add.at(out, (p[ok], t[ok]), 1)
  return out


def original_match(flat_preds, flat_targets, preds_k, targets_k):
  c = contingency(flat_preds, flat_targets, preds_k, targets_k)
  return [(int(o), int(np.argmax(c[o]))) for o in range(preds_k)]   # np.argmax: first maximum


def hungarian_match(flat_preds, flat_targets, preds_k, targets_k):
  assert preds_k == targets_k
  c = contingency(flat_preds, flat_targets, preds_k, targets_k)
  r, col = linear_sum_assignment(len(np.asarray(flat_targets)) - c)
  return [(int(a), int(b)) for a, b in zip(r, col)]


def acc(preds, targets):
  p, t = np.asarray(preds), np.asarray(targets)
  return int((p == t).sum()) / float(p.shape[0])
