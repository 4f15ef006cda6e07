"""Drop-in for /root/reference/code/utils/cluster/eval_metrics.py: ``_original_match``,
``_hungarian_match``, ``_acc`` with the reference's signatures, asserts and return formats.

The reference builds the cluster-vs-class counts with preds_k x targets_k masked sums, each a tiny
kernel followed by ``int(...)`` (a host sync): 1 400 syncs per sub-head at k = 140, gt_k = 10
(eval_metrics.py:18-24, 42-46).  Here ONE contingency kernel (csrc/eval_metrics.hip) produces the
whole matrix; the tiny k x k assignment stays on the host, as in the reference
(``linear_assignment``; scipy's solver here -- sklearn.utils.linear_assignment_ no longer exists).
"""
import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr


def _counts(flat_preds, flat_targets, preds_k, targets_k):
  assert (isinstance(flat_preds, torch.Tensor) and
          isinstance(flat_targets, torch.Tensor) and
          flat_preds.is_cuda and flat_targets.is_cuda)
  p = flat_preds.reshape(-1).long().contiguous()
  t = flat_targets.reshape(-1).long().contiguous()
  assert p.numel() == t.numel()
  counts = torch.empty((preds_k, targets_k), dtype=torch.long, device=p.device)
  check(lib().iic_contingency(ptr(p), ptr(t), p.numel(), int(preds_k), int(targets_k), ptr(counts),
                              stream_ptr()), "iic_contingency")
  return counts.cpu().numpy()      # the one device -> host transfer of the match


def _original_match(flat_preds, flat_targets, preds_k, targets_k):
  # map each output channel to the best matching ground truth (many to one); first maximum in
  # class order wins, as the reference's strict '>' update (eval_metrics.py:22)
  c = _counts(flat_preds, flat_targets, preds_k, targets_k)
  return [(out_c, int(np.argmax(c[out_c]))) for out_c in range(preds_k)]


def _hungarian_match(flat_preds, flat_targets, preds_k, targets_k):
  from scipy.optimize import linear_sum_assignment
  num_samples = flat_targets.shape[0]
  assert (preds_k == targets_k)  # one to one
  num_correct = _counts(flat_preds, flat_targets, preds_k, targets_k)
  rows, cols = linear_sum_assignment(num_samples - num_correct)
  # return as list of tuples, out_c to gt_c
  return [(int(out_c), int(gt_c)) for out_c, gt_c in zip(rows, cols)]


def _acc(preds, targets, num_k, verbose=0):
  assert (isinstance(preds, torch.Tensor) and
          isinstance(targets, torch.Tensor) and
          preds.is_cuda and targets.is_cuda)
  if verbose >= 2:
    print("calling acc...")
  assert (preds.shape == targets.shape)
  assert (preds.max() < num_k and targets.max() < num_k)
  p = preds.reshape(-1).long().contiguous()
  t = targets.reshape(-1).long().contiguous()
  cnt = torch.empty((), dtype=torch.long, device=p.device)
  check(lib().iic_count_equal(ptr(p), ptr(t), p.numel(), ptr(cnt), stream_ptr()), "iic_count_equal")
  return int(cnt) / float(preds.shape[0])
