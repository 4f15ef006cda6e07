"""CPU restatement of the reference's IID losses (TEST INFRASTRUCTURE ONLY).

Follows, line by line:
  * IID_loss / compute_joint      /root/reference/code/utils/cluster/IID_losses.py:6-47
  * IID_segmentation_loss         /root/reference/code/utils/segmentation/IID_losses.py:14-83
  * IID_segmentation_loss_uncollapsed                                       ...:86-159
  * perform_affine_tf             /root/reference/code/utils/segmentation/transforms.py:131-143

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which
oracle/gen_golden.py produced by running the reference's own functions
(imported read-only) in the build container.

Only difference from the reference text: the reference clamps by in-place masked
assignment into an ``expand``-ed view (IID_losses.py:12-19), which modern torch
refuses to back-propagate through; we materialise the expansion first
(``.clone()``) -- values and gradients are unchanged.
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

EPS = sys.float_info.epsilon


# ----------------------------------------------------------------------------
# clustering loss
# ----------------------------------------------------------------------------

def compute_joint(x_out, x_tf_out):
  """IID_losses.py:36-47."""
  bn, k = x_out.size()
  assert x_tf_out.size(0) == bn and x_tf_out.size(1) == k
  p_i_j = x_out.unsqueeze(2) * x_tf_out.unsqueeze(1)  # bn, k, k
  p_i_j = p_i_j.sum(dim=0)  # k, k
  p_i_j = (p_i_j + p_i_j.t()) / 2.  # symmetrise
  p_i_j = p_i_j / p_i_j.sum()  # normalise
  return p_i_j


def IID_loss(x_out, x_tf_out, lamb=1.0, EPS=EPS):
  """IID_losses.py:6-33 (differentiable on modern torch)."""
  _, k = x_out.size()
  p_i_j = compute_joint(x_out, x_tf_out)
  assert p_i_j.size() == (k, k)

  p_i = p_i_j.sum(dim=1).view(k, 1).expand(k, k).clone()
  p_j = p_i_j.sum(dim=0).view(1, k).expand(k, k).clone()

  p_i_j = p_i_j.clone()
  p_i_j[(p_i_j < EPS).data] = EPS
  p_j[(p_j < EPS).data] = EPS
  p_i[(p_i < EPS).data] = EPS

  loss = - p_i_j * (torch.log(p_i_j) - lamb * torch.log(p_j) - lamb * torch.log(p_i))
  loss = loss.sum()
  loss_no_lamb = - p_i_j * (torch.log(p_i_j) - torch.log(p_j) - torch.log(p_i))
  loss_no_lamb = loss_no_lamb.sum()
  return loss, loss_no_lamb


def raw_joint_np(z, zt):
  """R = sum_n z_n z'_n^T  (the additive, all-reducible quantity), float64."""
  z = np.asarray(z, dtype=np.float64)
  zt = np.asarray(zt, dtype=np.float64)
  return z.T @ zt


def loss_and_grad_from_raw_np(R, lamb=1.0, g_loss=1.0, g_loss_no_lamb=0.0, EPS=EPS):
  """float64 closed form of IID_losses.py:6-33 starting from the raw joint R.

  Returns (loss, loss_no_lamb, dR) where dR = d(g_loss*loss + g_nl*loss_no_lamb)/dR,
  honouring the reference's clamp semantics (zero gradient through clamped
  entries; marginals computed before clamping).
  """
  R = np.asarray(R, dtype=np.float64)
  k = R.shape[0]
  Ps = (R + R.T) / 2.0
  S = Ps.sum()
  P = Ps / S
  pi = P.sum(axis=1)  # row marginal  (p_i in the reference: expand along j)
  pj = P.sum(axis=0)
  mP = P >= EPS
  mi = pi >= EPS
  mj = pj >= EPS
  Pc = np.where(mP, P, EPS)
  pic = np.where(mi, pi, EPS)
  pjc = np.where(mj, pj, EPS)
  lP, li, lj = np.log(Pc), np.log(pic)[:, None], np.log(pjc)[None, :]

  def one(l):
    val = -(Pc * (lP - l * lj - l * li)).sum()
    dP = np.where(mP, -(lP - l * lj - l * li) - 1.0, 0.0)
    row = np.where(mi, l * Pc.sum(axis=1) / pic, 0.0)  # via p_i (expanded over j)
    col = np.where(mj, l * Pc.sum(axis=0) / pjc, 0.0)
    dP = dP + row[:, None] + col[None, :]
    return val, dP

  loss, dP1 = one(lamb)
  loss_nl, dP2 = one(1.0)
  dP = g_loss * dP1 + g_loss_no_lamb * dP2
  dPs = (dP - (dP * P).sum()) / S
  dR = (dPs + dPs.T) / 2.0
  return loss, loss_nl, dR


def iid_loss_np(z, zt, lamb=1.0, g_loss=1.0, g_loss_no_lamb=0.0):
  """float64 loss + input gradients (dz, dz')."""
  z = np.asarray(z, dtype=np.float64)
  zt = np.asarray(zt, dtype=np.float64)
  R = raw_joint_np(z, zt)
  loss, loss_nl, dR = loss_and_grad_from_raw_np(R, lamb, g_loss, g_loss_no_lamb)
  dz = zt @ dR.T
  dzt = z @ dR
  return loss, loss_nl, dz, dzt


# ----------------------------------------------------------------------------
# segmentation losses
# ----------------------------------------------------------------------------

def perform_affine_tf(data, tf_matrices):
  """transforms.py:131-143 (torch>=1.3 default align_corners=False; exact for
  identity / flip matrices, the only ones the published runs use)."""
  n_i, k, h, w = data.shape
  n_i2, r, c = tf_matrices.shape
  assert n_i == n_i2 and r == 2 and c == 3
  grid = F.affine_grid(tf_matrices, data.shape, align_corners=False)
  return F.grid_sample(data, grid, padding_mode="zeros", align_corners=False)


def random_translation_multiple(data, half_side_min, half_side_max):
  """segmentation/transforms.py:145-165: ONE random (x, y) displacement for the whole batch, drawn
  from numpy's global RNG exactly as the reference draws it (magnitude in [min, max] per axis,
  random sign per axis), applied by zero padding + cropping."""
  n, c, h, w = data.shape
  data = F.pad(data, (half_side_max, half_side_max, half_side_max, half_side_max), "constant", 0)
  t = np.random.randint(half_side_min, half_side_max + 1, size=(2,))
  polarities = np.random.choice([-1, 1], size=(2,), replace=True)
  t *= polarities
  t += half_side_max
  return data[:, :, t[1]:(t[1] + h), t[0]:(t[0] + w)]


def _seg_joint(x1_outs, x2_outs, all_affine2_to_1, all_mask_img1, half_T_side_dense,
               half_T_side_sparse_min=0, half_T_side_sparse_max=0):
  """segmentation/IID_losses.py:27-56 / 99-126: returns p_i_j [k, k, 2T+1, 2T+1]."""
  x2_outs_inv = perform_affine_tf(x2_outs, all_affine2_to_1)
  if (half_T_side_sparse_min != 0) or (half_T_side_sparse_max != 0):      # :101-104
    x2_outs_inv = random_translation_multiple(x2_outs_inv, half_side_min=half_T_side_sparse_min,
                                              half_side_max=half_T_side_sparse_max)
  bn, k, h, w = x1_outs.shape
  m = all_mask_img1.view(bn, 1, h, w)
  x1 = (x1_outs * m).permute(1, 0, 2, 3).contiguous()
  x2 = (x2_outs_inv * m).permute(1, 0, 2, 3).contiguous()
  return F.conv2d(x1, weight=x2, padding=(half_T_side_dense, half_T_side_dense))


def IID_segmentation_loss(x1_outs, x2_outs, all_affine2_to_1=None, all_mask_img1=None,
                          lamb=1.0, half_T_side_dense=None,
                          half_T_side_sparse_min=0, half_T_side_sparse_max=0):
  """segmentation/IID_losses.py:14-83 (collapsed; normaliser detached :60)."""
  k = x1_outs.shape[1]
  p_i_j = _seg_joint(x1_outs, x2_outs, all_affine2_to_1, all_mask_img1, half_T_side_dense,
                     half_T_side_sparse_min or 0, half_T_side_sparse_max or 0)
  p_i_j = p_i_j.sum(dim=2, keepdim=False).sum(dim=2, keepdim=False)  # k, k
  current_norm = float(p_i_j.sum())
  p_i_j = p_i_j / current_norm
  p_i_j = (p_i_j + p_i_j.t()) / 2.
  p_i_mat = p_i_j.sum(dim=1).unsqueeze(1).clone()
  p_j_mat = p_i_j.sum(dim=0).unsqueeze(0).clone()
  p_i_j = p_i_j.clone()
  p_i_j[(p_i_j < EPS).data] = EPS
  p_i_mat[(p_i_mat < EPS).data] = EPS
  p_j_mat[(p_j_mat < EPS).data] = EPS
  loss = (-p_i_j * (torch.log(p_i_j) - lamb * torch.log(p_i_mat) -
                    lamb * torch.log(p_j_mat))).sum()
  loss_no_lamb = (-p_i_j * (torch.log(p_i_j) - torch.log(p_i_mat) -
                            torch.log(p_j_mat))).sum()
  return loss, loss_no_lamb


def IID_segmentation_loss_uncollapsed(x1_outs, x2_outs, all_affine2_to_1=None,
                                      all_mask_img1=None, lamb=1.0, half_T_side_dense=None,
                                      half_T_side_sparse_min=0, half_T_side_sparse_max=0):
  """segmentation/IID_losses.py:86-159."""
  k = x1_outs.shape[1]
  p_i_j = _seg_joint(x1_outs, x2_outs, all_affine2_to_1, all_mask_img1, half_T_side_dense,
                     half_T_side_sparse_min or 0, half_T_side_sparse_max or 0)
  T_side_dense = half_T_side_dense * 2 + 1
  p_i_j = p_i_j.permute(2, 3, 0, 1)
  p_i_j = p_i_j / p_i_j.sum(dim=3, keepdim=True).sum(dim=2, keepdim=True)
  p_i_j = (p_i_j + p_i_j.permute(0, 1, 3, 2)) / 2.0
  p_i_mat = p_i_j.sum(dim=2, keepdim=True).repeat(1, 1, k, 1)
  p_j_mat = p_i_j.sum(dim=3, keepdim=True).repeat(1, 1, 1, k)
  p_i_j = p_i_j.clone()
  p_i_j[(p_i_j < EPS).data] = EPS
  p_i_mat[(p_i_mat < EPS).data] = EPS
  p_j_mat[(p_j_mat < EPS).data] = EPS
  loss = (-p_i_j * (torch.log(p_i_j) - lamb * torch.log(p_i_mat) -
                    lamb * torch.log(p_j_mat))).sum() / (T_side_dense * T_side_dense)
  loss_no_lamb = (-p_i_j * (torch.log(p_i_j) - torch.log(p_i_mat) -
                            torch.log(p_j_mat))).sum() / (T_side_dense * T_side_dense)
  return loss, loss_no_lamb


# ----------------------------------------------------------------------------
# seeded loss-input generators
# ----------------------------------------------------------------------------

def make_softmax_pair(bn, k, kind="trained", seed=0, dtype=np.float32):
  """Post-softmax (z, z') rows. 'trained': shared peaky logits + private noise;
  'init': near-uniform; 'onehot': balanced one-hot, z == z'."""
  rng = np.random.default_rng(seed)
  if kind == "trained":
    shared = rng.standard_normal((bn, k)) * 4.0
    a = shared + rng.standard_normal((bn, k))
    b = shared + rng.standard_normal((bn, k))
  elif kind == "init":
    a = rng.standard_normal((bn, k)) * 0.05
    b = rng.standard_normal((bn, k)) * 0.05
  elif kind == "onehot":
    idx = np.arange(bn) % k
    a = np.full((bn, k), -1e4)
    a[np.arange(bn), idx] = 0.0
    b = a.copy()
  else:
    raise ValueError(kind)

  def sm(x):
    x = x - x.max(axis=1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=1, keepdims=True)
  return sm(a).astype(dtype), sm(b).astype(dtype)
