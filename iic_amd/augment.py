"""Paired augmentation on the GPU.

Host side of csrc/augment.hip.  Mirrors the three transform pipelines the reference builds in
/root/reference/code/utils/cluster/transforms.py:107-217 (`sobel_make_transforms`, default
branch -- crop_orig, no fluid_warp / cutout / random affine / demean):

  tf1  RandomCrop(rand_crop_sz) -> Resize(input_sz) -> custom_greyscale_to_tensor(include_rgb)
  tf2  RandomCrop -> Resize -> RandomHorizontalFlip -> ColorJitter(0.4, 0.4, 0.4, 0.125)
       -> custom_greyscale_to_tensor
  tf3  CenterCrop -> Resize -> custom_greyscale_to_tensor

and the way code/utils/cluster/data.py:259-335 (`_create_dataloaders`) pairs them: one loader
with tf1 and `num_dataloaders` loaders with tf2 over the SAME sample order, so a training step
sees imgs = tf1(x_i) and imgs_tf = tf2(x_i) for d = 0..num_dataloaders-1.  Here the uint8 dataset
lives in HBM (STL10 train+unlabeled is 105k x 96 x 96 x 3 = 2.9 GB of 288 GB) and a batch of
both views is one kernel launch; the pixels are bit-identical to PIL's for the same random draws
(tests/test_gpu_augment.py against oracle/augment_oracle.py).

The random parameters are drawn on the host with the distributions of torchvision 0.2.1's
RandomCrop.get_params / RandomHorizontalFlip / ColorJitter.get_params (uniform factors, shuffled
op order); they are a few dozen bytes per image.
"""
import math

import numpy as np
import torch

from . import _lib

OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE = 0, 1, 2, 3
_PRECISION_BITS = 32 - 8 - 2
IPARAMS, FPARAMS = 12, 4


def bilinear_tables(in_size, out_size):
  """Pillow's resampling coefficients for the triangle (BILINEAR) filter, quantised to 22
  fractional bits: bounds int32 [out][2] = (first source index, tap count), kk int32
  [out][ksize].  When out < in the filter support grows with the scale (antialiasing), as in
  Pillow."""
  scale = in_size / float(out_size)
  fscale = max(scale, 1.0)
  support = fscale                      # bilinear support 1.0 * filterscale
  ksize = int(math.ceil(support)) * 2 + 1
  bounds = np.zeros((out_size, 2), np.int32)
  kk = np.zeros((out_size, ksize), np.int32)
  one = float(1 << _PRECISION_BITS)
  for o in range(out_size):
    center = (o + 0.5) * scale
    lo = max(int(center - support + 0.5), 0)
    n = min(int(center + support + 0.5), in_size) - lo
    w = [max(0.0, 1.0 - abs((lo + j - center + 0.5) / fscale)) for j in range(n)]
    tot = sum(w)
    for j in range(n):
      v = w[j] / tot if tot != 0.0 else w[j]
      kk[o, j] = int(v * one + (0.5 if v >= 0 else -0.5))
    bounds[o] = (lo, n)
  return bounds, kk


def hue_shift(hue_factor):
  """np.uint8(hue_factor * 255) of torchvision's adjust_hue as a wrap-around uint8 increment."""
  return int(hue_factor * 255) % 256


class PairedAugmenter(object):
  """images_u8: uint8 [B, H, W, 3] on the GPU (HWC, the layout torchvision datasets hold).

  plain(idx) / jittered(idx) / center(idx) return float32 [len(idx), C, input_sz, input_sz]
  (C = 4 with include_rgb, else 1) for tf1 / tf2 / tf3; `draw` exposes the parameter draws so
  that callers (and the tests) can replay them.
  """

  def __init__(self, images_u8, rand_crop_sz, input_sz, include_rgb, jitter=(0.4, 0.4, 0.4, 0.125),
               seed=0):
    assert images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 \
        and images_u8.shape[3] == 3 and images_u8.is_contiguous()
    self.images = images_u8
    self.B, self.H, self.W = (int(v) for v in images_u8.shape[:3])
    self.crop, self.S, self.include_rgb = int(rand_crop_sz), int(input_sz), bool(include_rgb)
    assert self.crop <= self.H and self.crop <= self.W
    self.jitter = tuple(float(j) for j in jitter)
    self.rng = np.random.RandomState(seed)
    dev = images_u8.device
    bounds, kk = bilinear_tables(self.crop, self.S)
    self.ksize = int(kk.shape[1])
    self.bounds = torch.from_numpy(bounds).to(dev)
    self.kk = torch.from_numpy(kk).to(dev)
    self.lut = (torch.arange(256, dtype=torch.float32) / 255).to(dev)   # to_tensor's .div(255)

  # ---- parameter draws (host) ------------------------------------------------------------
  def draw(self, idx, mode):
    """mode 'plain' (tf1), 'jittered' (tf2) or 'center' (tf3).  Returns (iparams int32 [n, 12],
    fparams float32 [n, 4]) as iic_augment reads them."""
    idx = np.asarray(idx, dtype=np.int64).reshape(-1)
    n = idx.shape[0]
    ip = np.zeros((n, IPARAMS), np.int32)
    fp = np.zeros((n, FPARAMS), np.float32)
    ip[:, 0] = idx
    if mode == "center":
      # torchvision F.center_crop: i = int(round((h - th) / 2.)), likewise j
      ip[:, 1] = int(round((self.W - self.crop) / 2.))
      ip[:, 2] = int(round((self.H - self.crop) / 2.))
      return ip, fp
    r = self.rng
    ip[:, 1] = r.randint(0, self.W - self.crop + 1, size=n)
    ip[:, 2] = r.randint(0, self.H - self.crop + 1, size=n)
    if mode == "plain":
      return ip, fp
    assert mode == "jittered"
    b, c, s, h = self.jitter
    ip[:, 3] = r.random_sample(n) < 0.5
    ip[:, 4] = 4
    ip[:, 5:9] = np.argsort(r.random_sample((n, 4)), axis=1)   # uniform random op orders
    fp[:, OP_BRIGHTNESS] = r.uniform(max(0.0, 1 - b), 1 + b, size=n)
    fp[:, OP_CONTRAST] = r.uniform(max(0.0, 1 - c), 1 + c, size=n)
    fp[:, OP_SATURATION] = r.uniform(max(0.0, 1 - s), 1 + s, size=n)
    fp[:, OP_HUE] = r.uniform(-h, h, size=n)                 # kept for the record; the kernel reads
    ip[:, 9] = [hue_shift(float(v)) for v in fp[:, OP_HUE]]  # the uint8 increment derived from it
    return ip, fp

  # ---- device ----------------------------------------------------------------------------
  def apply(self, iparams, fparams):
    ip = torch.from_numpy(np.ascontiguousarray(iparams, dtype=np.int32)).to(self.images.device, non_blocking=True)
    fp = torch.from_numpy(np.ascontiguousarray(fparams, dtype=np.float32)).to(self.images.device, non_blocking=True)
    n = int(ip.shape[0])
    assert ip.shape == (n, IPARAMS) and fp.shape == (n, FPARAMS)
    src = iparams[:, 0]
    assert n > 0 and src.min() >= 0 and src.max() < self.B, "source index out of range"
    assert (iparams[:, 1] >= 0).all() and (iparams[:, 1] + self.crop <= self.W).all()
    assert (iparams[:, 2] >= 0).all() and (iparams[:, 2] + self.crop <= self.H).all()
    C = 4 if self.include_rgb else 1
    out = torch.empty(n, C, self.S, self.S, device=self.images.device, dtype=torch.float32)
    _lib.check(_lib.lib().iic_augment(
      self.images.data_ptr(), self.B, self.H, self.W, ip.data_ptr(), fp.data_ptr(), n,
      self.bounds.data_ptr(), self.kk.data_ptr(), self.ksize, self.crop, self.S,
      self.lut.data_ptr(), out.data_ptr(), int(self.include_rgb), _lib.stream_ptr()), "iic_augment")
    return out

  def plain(self, idx):
    return self.apply(*self.draw(idx, "plain"))

  def jittered(self, idx):
    return self.apply(*self.draw(idx, "jittered"))

  def center(self, idx):
    return self.apply(*self.draw(idx, "center"))

  def paired_batch(self, idx, num_dataloaders=1):
    """What one iteration of the reference's zipped dataloaders yields
    (code/scripts/cluster/cluster_sobel.py:205-232): imgs (tf1) and num_dataloaders
    independently re-drawn imgs_tf (tf2) of the same samples."""
    return self.plain(idx), [self.jittered(idx) for _ in range(num_dataloaders)]


class _PairedLoader(object):
  """One element of the list `_create_dataloaders` returns (code/utils/cluster/data.py:259-335):
  iterating yields (images, targets) batches in SEQUENTIAL sample order (the reference builds its
  training loaders with shuffle=False, drop_last=False), every loader of the list over the same
  indices, the first with tf1 and the others with independently drawn tf2."""

  def __init__(self, augmenter, targets, batch_sz, jittered):
    self.aug, self.targets, self.batch_sz, self.jittered = augmenter, targets, int(batch_sz), jittered
    self.n = int(targets.shape[0])

  def __len__(self):
    return (self.n + self.batch_sz - 1) // self.batch_sz

  def __iter__(self):
    for lo in range(0, self.n, self.batch_sz):
      idx = np.arange(lo, min(self.n, lo + self.batch_sz))
      imgs = self.aug.jittered(idx) if self.jittered else self.aug.plain(idx)
      yield imgs, self.targets[lo:lo + self.batch_sz]


def paired_dataloaders(augmenter, targets, dataloader_batch_sz, num_dataloaders):
  """Drop-in for the list of DataLoaders the training scripts zip
  (code/scripts/cluster/cluster_sobel.py:204-232): [tf1 loader] + num_dataloaders x [tf2 loader].
  Batches are already on the GPU (the scripts' `.cuda()` is then a no-op)."""
  assert int(targets.shape[0]) == augmenter.B
  return [_PairedLoader(augmenter, targets, dataloader_batch_sz, False)] + \
         [_PairedLoader(augmenter, targets, dataloader_batch_sz, True) for _ in range(num_dataloaders)]
