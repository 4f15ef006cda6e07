"""Drop-in IID_segmentation_loss / IID_segmentation_loss_uncollapsed on the HIP kernels
(iic_amd/csrc/seg_loss.hip + the k x k stage of iid_loss.hip).

Mirrors /root/reference/code/utils/segmentation/IID_losses.py:14-159 -- same keyword
signature, same asserts (inputs require grad; affine / mask do not), returns
(loss, loss_no_lamb) 0-d tensors.  Identity and axis-flip affine2_to_1 matrices (what
potsdam.py:189-202 / cocostuff.py produce unless --use_random_affine, which no published run
sets) are folded into the kernels' index arithmetic; general matrices (perform_affine_tf,
transforms.py:131-143) and the sparse random translation (IID_losses.py:101-104,
transforms.py:145-165) go through an explicit warp of the second view (csrc/warp.hip).

Data-parallel: the raw per-shift joints are all-reduced (SUM) like the clustering loss.
"""
from sys import float_info

import torch

from . import dist as iic_dist
from . import ops
from ._lib import check, lib, ptr, stream_ptr

EPS = float_info.epsilon
F32 = torch.float32


# F.affine_grid / F.grid_sample convention of perform_affine_tf.  False = what the reference's code
# does on the torch of this image (>= 1.3 default; the goldens were produced that way); True = the
# behaviour of the reference's pinned torch 0.4.1.  Identity / flip matrices are exact under both.
ALIGN_CORNERS = [False]


def _flips_from_affine(aff):
  """[bn, 2, 3] (host) -> int32 [bn, 2] (flip x, flip y), or None for anything but identity /
  axis flips (the general warp kernel handles those)."""
  a = aff.detach().float().cpu()
  lin, tr = a[:, :, :2], a[:, :, 2]
  ok = (tr.abs() < 1e-6).all() and (lin[:, 0, 1].abs() < 1e-6).all() and \
       (lin[:, 1, 0].abs() < 1e-6).all() and ((lin[:, 0, 0].abs() - 1).abs() < 1e-6).all() and \
       ((lin[:, 1, 1].abs() - 1).abs() < 1e-6).all()
  if not bool(ok):
    return None
  return torch.stack([(lin[:, 0, 0] < 0), (lin[:, 1, 1] < 0)], dim=1).to(torch.int32)


def _pixel_matrices(aff, H, W):
  """theta [bn, 2, 3] in normalised coordinates (F.affine_grid) -> [bn, 6] pixel-space rows:
  source pixel (ix, iy) = M (ox, oy, 1), for the grid_sample un-normalisation in use."""
  t = aff.detach().double().cpu()
  M = torch.empty((t.size(0), 6), dtype=torch.float64)
  if ALIGN_CORNERS[0]:
    # xn = 2 ox / (W-1) - 1 ; ix = (xs + 1) (W-1) / 2
    ax, ay = (W - 1) / 2.0, (H - 1) / 2.0
    M[:, 0] = t[:, 0, 0]
    M[:, 1] = t[:, 0, 1] * ax / ay if H > 1 else 0.0
    M[:, 2] = ax * (-t[:, 0, 0] - t[:, 0, 1] + t[:, 0, 2] + 1.0)
    M[:, 3] = t[:, 1, 0] * ay / ax if W > 1 else 0.0
    M[:, 4] = t[:, 1, 1]
    M[:, 5] = ay * (-t[:, 1, 0] - t[:, 1, 1] + t[:, 1, 2] + 1.0)
  else:
    # xn = (2 ox + 1) / W - 1 ; ix = ((xs + 1) W - 1) / 2
    M[:, 0] = t[:, 0, 0]
    M[:, 1] = t[:, 0, 1] * W / H
    M[:, 2] = (W / 2.0) * (t[:, 0, 0] / W - t[:, 0, 0] + t[:, 0, 1] / H - t[:, 0, 1] + t[:, 0, 2] + 1.0) - 0.5
    M[:, 3] = t[:, 1, 0] * H / W
    M[:, 4] = t[:, 1, 1]
    M[:, 5] = (H / 2.0) * (t[:, 1, 0] / W - t[:, 1, 0] + t[:, 1, 1] / H - t[:, 1, 1] + t[:, 1, 2] + 1.0) - 0.5
  return M.float()


class _AffineWarpFn(torch.autograd.Function):
  """perform_affine_tf + random_translation_multiple of the second view (csrc/warp.hip)."""

  @staticmethod
  def forward(ctx, x, mats, sx, sy):
    assert x.is_cuda and x.dtype == F32 and x.dim() == 4
    x = x.contiguous()
    n, k, h, w = x.shape
    out = torch.empty_like(x)
    check(lib().iic_affine_warp_fwd(ptr(x), ptr(mats), ptr(out), n, k, h, w, int(sx), int(sy),
                                    stream_ptr()), "iic_affine_warp_fwd")
    ctx.save_for_backward(mats)
    ctx.meta = (n, k, h, w, int(sx), int(sy))
    return out

  @staticmethod
  def backward(ctx, dout):
    mats, = ctx.saved_tensors
    n, k, h, w, sx, sy = ctx.meta
    dout = dout.contiguous()
    dx = torch.empty_like(dout)
    check(lib().iic_affine_warp_bwd(ptr(dout), ptr(mats), ptr(dx), n, k, h, w, sx, sy, stream_ptr()),
          "iic_affine_warp_bwd")
    return dx, None, None, None


def _draw_sparse_shift(half_side_min, half_side_max):
  """The displacement random_translation_multiple draws (transforms.py:155-161), from numpy's
  global RNG with the reference's own sequence of calls; returns (shift_x, shift_y) such that
  out[y][x] = in[y + shift_y][x + shift_x]."""
  import numpy as np
  t = np.random.randint(half_side_min, half_side_max + 1, size=(2,))
  polarities = np.random.choice([-1, 1], size=(2,), replace=True)
  t *= polarities
  return int(t[0]), int(t[1])


class _SegLossFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x1, x2, flips, mask, lamb, T, collapsed):
    assert x1.is_cuda and x1.dtype == F32 and x1.shape == x2.shape and x1.dim() == 4
    ops.join()      # (no-op unless a view was forked onto a side stream: iic_amd.ops.branch)
    x1, x2, mask = x1.contiguous(), x2.contiguous(), mask.contiguous().float()
    bn, k, h, w = x1.shape
    L, s = lib(), stream_ptr()
    nq = 2 * T + 1
    ns = L.iic_seg_joint_nsplit(bn, h, k, T)
    part = torch.empty((ns, nq * nq, k, k), dtype=F32, device=x1.device)
    check(L.iic_seg_joint_raw(ptr(x1), ptr(x2), ptr(mask), ptr(flips), ptr(part), bn, k, h, w, T,
                              ns, s), "iic_seg_joint_raw")
    if iic_dist.enabled():
      R = torch.empty((1, nq * nq, k, k), dtype=F32, device=x1.device)
      check(L.iic_colsum_f32(ptr(part), ptr(R), ns, nq * nq * k * k, 0, s), "iic_colsum_f32")
      iic_dist.all_reduce_sum_(R)
      part, ns = R, 1
    H = 1 if collapsed else nq * nq
    nparts = ns * nq * nq if collapsed else ns
    ws = torch.empty(L.iic_iid_workspace_bytes(H, k) // 8, dtype=torch.float64, device=x1.device)
    loss = torch.empty(H, dtype=F32, device=x1.device)
    loss_nl = torch.empty(H, dtype=F32, device=x1.device)
    dR1 = torch.empty((H, k, k), dtype=F32, device=x1.device)
    dR2 = torch.empty((H, k, k), dtype=F32, device=x1.device)
    check(L.iic_seg_loss_from_joint(ptr(part), nparts, H, k, float(lamb), float(EPS), ptr(ws),
                                    ptr(loss), ptr(loss_nl), ptr(dR1), ptr(dR2),
                                    1 if collapsed else 0, s), "iic_seg_loss_from_joint")
    ctx.save_for_backward(x1, x2, flips, mask, dR1, dR2)
    ctx.meta = (T, collapsed)
    return loss, loss_nl

  @staticmethod
  def backward(ctx, g_loss, g_nl):
    x1, x2, flips, mask, dR1, dR2 = ctx.saved_tensors
    T, collapsed = ctx.meta
    bn, k, h, w = x1.shape
    g_loss, g_nl = g_loss.contiguous().to(F32), g_nl.contiguous().to(F32)
    dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
    L, s = lib(), stream_ptr()
    # scratch of the two launches (stream-ordered, so they share it): per-shift gradient matrices
    ws = torch.empty(L.iic_seg_grad_workspace_bytes(k, T) // 4, dtype=F32, device=x1.device)
    check(L.iic_seg_grad(ptr(x2), ptr(mask), ptr(flips), ptr(dR1), ptr(dR2), ptr(g_loss), ptr(g_nl),
                         ptr(dx1), bn, k, h, w, T, 0, 1 if collapsed else 0, ptr(ws), s), "iic_seg_grad")
    check(L.iic_seg_grad(ptr(x1), ptr(mask), ptr(flips), ptr(dR1), ptr(dR2), ptr(g_loss), ptr(g_nl),
                         ptr(dx2), bn, k, h, w, T, 1, 1 if collapsed else 0, ptr(ws), s), "iic_seg_grad")
    return dx1, dx2, None, None, None, None, None


def _seg_loss(x1_outs, x2_outs, all_affine2_to_1, all_mask_img1, lamb, half_T_side_dense,
              half_T_side_sparse_min, half_T_side_sparse_max, collapsed):
  ops.join()          # before anything below reads a forked view's outputs on this stream (iic_amd.losses.IID_loss)
  assert x1_outs.requires_grad
  assert x2_outs.requires_grad
  assert not all_affine2_to_1.requires_grad
  assert not all_mask_img1.requires_grad
  assert x1_outs.shape == x2_outs.shape
  # unchanged scripts under torchrun hand over full-batch masks / affines with sharded outputs
  all_affine2_to_1 = iic_dist.shard_like(all_affine2_to_1, x1_outs.size(0))
  all_mask_img1 = iic_dist.shard_like(all_mask_img1, x1_outs.size(0))
  sparse = (half_T_side_sparse_min != 0) or (half_T_side_sparse_max != 0)
  flips = _flips_from_affine(all_affine2_to_1)
  if flips is None or sparse:
    # general matrices (--use_random_affine) and / or the sparse random translation
    # (IID_losses.py:101-104): warp the second view explicitly, then the flip-free loss kernels
    bn, _, h, w = x2_outs.shape
    sx, sy = _draw_sparse_shift(half_T_side_sparse_min, half_T_side_sparse_max) if sparse else (0, 0)
    mats = _pixel_matrices(all_affine2_to_1, h, w).to(x2_outs.device)
    x2_outs = _AffineWarpFn.apply(x2_outs, mats, sx, sy)
    flips = torch.zeros((bn, 2), dtype=torch.int32)
  flips = flips.to(x1_outs.device)
  T = int(half_T_side_dense)
  loss, loss_nl = _SegLossFn.apply(x1_outs, x2_outs, flips, all_mask_img1, lamb, T, collapsed)
  if collapsed:
    return loss[0], loss_nl[0]
  n = float((2 * T + 1) ** 2)
  return loss.sum() / n, loss_nl.sum() / n


def IID_segmentation_loss(x1_outs, x2_outs, all_affine2_to_1=None, all_mask_img1=None, lamb=1.0,
                          half_T_side_dense=None, half_T_side_sparse_min=None,
                          half_T_side_sparse_max=None):
  """segmentation/IID_losses.py:14-83."""
  return _seg_loss(x1_outs, x2_outs, all_affine2_to_1, all_mask_img1, lamb, half_T_side_dense,
                   half_T_side_sparse_min or 0, half_T_side_sparse_max or 0, True)


def IID_segmentation_loss_uncollapsed(x1_outs, x2_outs, all_affine2_to_1=None, all_mask_img1=None,
                                      lamb=1.0, half_T_side_dense=None,
                                      half_T_side_sparse_min=None, half_T_side_sparse_max=None):
  """segmentation/IID_losses.py:86-159."""
  return _seg_loss(x1_outs, x2_outs, all_affine2_to_1, all_mask_img1, lamb, half_T_side_dense,
                   half_T_side_sparse_min or 0, half_T_side_sparse_max or 0, False)
