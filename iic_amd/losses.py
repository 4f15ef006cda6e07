"""Drop-in IID_loss on the fused HIP kernels (iic_amd/csrc/iid_loss.hip).

Mirrors /root/reference/code/utils/cluster/IID_losses.py:6-33:
    IID_loss(x_out, x_tf_out, lamb=1.0, EPS=sys.float_info.epsilon) -> (loss, loss_no_lamb)
both 0-d tensors on the inputs' device, differentiable w.r.t. both inputs, usable under
torch.no_grad() (cluster_eval.py:281-288).  Inputs: post-softmax [bn, k] fp32.

Data-parallel (iic_amd.dist enabled): the raw joint R = sum_n z_n z'_n^T is all-reduced
(SUM) between phase 1 and phase 2, so every rank evaluates the loss of the GLOBAL batch and
back-propagates dz = z' dR^T for its own rows.
"""
import sys

import torch

from . import dist as iic_dist
from . import ops
from ._lib import check, lib, ptr, stream_ptr

F32 = torch.float32


def _dense(t):
  """True if t ([H, bn, k], unit inner stride) is head-major or sample-major without gaps."""
  H, bn, k = t.shape
  st = t.stride()
  return st[2] == 1 and (st == (bn * k, k, 1) or st == (k, H * k, 1))


class _IIDLossFn(torch.autograd.Function):
  """z, zt: [H, bn, k] fp32 contiguous (H packed sub-heads). Returns loss[H], loss_no_lamb[H]."""

  @staticmethod
  def forward(ctx, z, zt, lamb, eps):
    assert z.is_cuda and zt.is_cuda, "IID_loss: HIP path needs device tensors (no CPU fallback)"
    ops.join()      # (no-op unless a view was forked onto a side stream: iic_amd.ops.branch)
    assert z.dtype == F32 and zt.dtype == F32 and z.shape == zt.shape and z.dim() == 3
    H, bn, k = z.shape
    # logical [H, bn, k]; physical either head-major [H][bn][k] or sample-major [bn][H][k]
    # (what the fused head kernel emits).  Anything else is densified (tiny copy).
    if not (_dense(z) and z.stride() == zt.stride()):
      z, zt = z.contiguous(), zt.contiguous()
    hs, ld = z.stride(0), z.stride(1)
    L = lib()
    s = stream_ptr()
    ns = L.iic_iid_nsplit(bn)
    part = torch.empty((ns, H, k, k), dtype=F32, device=z.device)
    check(L.iic_iid_joint_raw(ptr(z), ptr(zt), ptr(part), H, bn, k, hs, ld, ns, s), "iic_iid_joint_raw")
    nparts = ns
    if iic_dist.enabled():
      R = torch.empty((1, H, k, k), dtype=F32, device=z.device)
      check(L.iic_colsum_f32(ptr(part), ptr(R), ns, H * k * k, 0, s), "iic_colsum_f32")
      iic_dist.all_reduce_sum_(R)
      part, nparts = R, 1
    ws = torch.empty(L.iic_iid_workspace_bytes(H, k) // 8, dtype=torch.float64, device=z.device)
    loss = torch.empty(H, dtype=F32, device=z.device)
    loss_nl = torch.empty(H, dtype=F32, device=z.device)
    dR1 = torch.empty((H, k, k), dtype=F32, device=z.device)
    dR2 = torch.empty((H, k, k), dtype=F32, device=z.device)
    check(L.iic_iid_loss_from_joint(ptr(part), nparts, H, k, float(lamb), float(eps), ptr(ws),
                                    ptr(loss), ptr(loss_nl), ptr(dR1), ptr(dR2), s),
          "iic_iid_loss_from_joint")
    ctx.save_for_backward(z, zt, dR1, dR2)
    return loss, loss_nl

  @staticmethod
  def backward(ctx, g_loss, g_nl):
    z, zt, dR1, dR2 = ctx.saved_tensors
    H, bn, k = z.shape
    g_loss = g_loss.contiguous().to(F32)
    g_nl = g_nl.contiguous().to(F32)
    dz = torch.empty_strided(z.shape, z.stride(), dtype=F32, device=z.device)
    dzt = torch.empty_strided(z.shape, z.stride(), dtype=F32, device=z.device)
    check(lib().iic_iid_grad(ptr(z), ptr(zt), ptr(dR1), ptr(dR2), ptr(g_loss), ptr(g_nl), ptr(dz),
                             ptr(dzt), H, bn, k, z.stride(0), z.stride(1), stream_ptr()),
          "iic_iid_grad")
    return dz, dzt, None, None


BATCH_SUB_HEADS = [True]      # IID_loss: all sub-head pairs of two tagged output lists in one set of launches


def _packed_pair(x_out, x_tf_out, lamb, EPS):
  """(loss[H], loss_no_lamb[H]) of every sub-head pair (i, i) of the two forwards these tensors came from, computed
  once (at the first of the script's per-sub-head calls, cluster_sobel.py:241-253) and cached on the pack; None if
  the tensors are not same-index members of two tagged lists of equal shape."""
  pa, pb = getattr(x_out, "_iic_pack", None), getattr(x_tf_out, "_iic_pack", None)
  if pa is None or pb is None or pa[1] != pb[1]:
    return None
  A, B = pa[0], pb[0]
  ta, tb = A.alive(), B.alive()
  if ta is None or tb is None or ta[pa[1]] is not x_out or tb[pb[1]] is not x_tf_out:
    return None
  if len(ta) != len(tb) or any(a.shape != b.shape or a.shape != x_out.shape or not a.is_cuda or not b.is_cuda
                               for a, b in zip(ta, tb)):
    return None
  key = (id(B), float(lamb), float(EPS), torch.is_grad_enabled())
  hit = A.cache.get(key)
  if hit is None or hit[0]() is not B:
    import weakref
    res = _IIDLossFn.apply(torch.stack(ta, dim=0), torch.stack(tb, dim=0), lamb, EPS)
    hit = A.cache[key] = (weakref.ref(B), res)
  return hit[1]


def IID_loss(x_out, x_tf_out, lamb=1.0, EPS=sys.float_info.epsilon):
  """Reference signature (IID_losses.py:6)."""
  # The join comes FIRST: a view forked onto a side stream (ops.auto_branch) is only ordered before the caller's stream
  # from here on, and the batched evaluation below already reads both views' outputs on the caller's stream
  # (torch.stack) before _IIDLossFn.forward runs.  Round 5 (tools/race_hunt.py): with the stack in front of the join the
  # drop-in path -- graph replay, two streams -- computed its loss from partly stale outputs of the side view in about
  # half of all 4-step runs at 48 images (and the eager two-stream mode "in 1 of 13 runs" in round 4: the same read).
  ops.join()
  _, k = x_out.size()
  assert x_tf_out.size(0) == x_out.size(0) and x_tf_out.size(1) == k
  if BATCH_SUB_HEADS[0]:
    both = _packed_pair(x_out, x_tf_out, lamb, EPS)
    if both is not None:
      i = x_out._iic_pack[1]
      return both[0][i], both[1][i]
  loss, loss_nl = _IIDLossFn.apply(x_out.unsqueeze(0), x_tf_out.unsqueeze(0), lamb, EPS)
  return loss[0], loss_nl[0]


def IID_loss_heads(x_outs, x_tf_outs, lamb=1.0, EPS=sys.float_info.epsilon):
  """All sub-heads in 3 launches.  x_outs / x_tf_outs: the packed sample-major [bn, H, k]
  tensor the HIP head module produces (``net.forward_packed``), or lists of [bn, k].
  Returns (loss[H], loss_no_lamb[H])."""
  ops.join()          # before anything reads a forked view's outputs on this stream (see IID_loss)
  if isinstance(x_outs, (list, tuple)):
    x_outs = torch.stack(list(x_outs), dim=1)
    x_tf_outs = torch.stack(list(x_tf_outs), dim=1)
  return _IIDLossFn.apply(x_outs.permute(1, 0, 2), x_tf_outs.permute(1, 0, 2), lamb, EPS)
