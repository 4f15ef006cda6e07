"""Paired augmentation on the GPU.

Host side of csrc/augment.hip.  `PairedAugmenter` mirrors the three transform pipelines the reference builds in
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

`GreyscaleAugmenter` does the same for `greyscale_make_transforms` (transforms.py:220-330, the
MNIST scripts: mode-L images, optional RandomRotation, a crop size chosen per sample, ToTensor).

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
IPARAMS, FPARAMS = 20, 4


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


def rotation_fixed_point(angle, w, h):
  """PIL Image.rotate(angle, NEAREST, expand=False, center=None) as the six 16.16 fixed-point
  coefficients of ImagingTransformAffine's inverse mapping: source x = (a2 + a1*y + a0*x) >> 16,
  source y = (a5 + a4*y + a3*x) >> 16 (python doubles, cos/sin rounded to 15 digits, like PIL)."""
  a = -math.radians(angle % 360.0)
  m0, m1 = round(math.cos(a), 15), round(math.sin(a), 15)
  m3, m4 = round(-math.sin(a), 15), round(math.cos(a), 15)
  cx, cy = w / 2, h / 2
  m2 = m0 * (-cx) + m1 * (-cy) + 0.0 + cx
  m5 = m3 * (-cx) + m4 * (-cy) + 0.0 + cy

  def fix(v):
    return int(math.floor(v * 65536.0 + 0.5))
  return (fix(m0), fix(m1), fix(m2 + m0 * 0.5 + m1 * 0.5), fix(m3), fix(m4), fix(m5 + m3 * 0.5 + m4 * 0.5))


def _center_xy(w, h, crop):
  """torchvision 0.2.1 F.center_crop: i = int(round((h - th) / 2.)), j = int(round((w - tw) / 2.))."""
  return int(round((w - crop) / 2.)), int(round((h - crop) / 2.))


class _AugmenterBase(object):
  """Dataset + resampling tables on the GPU, and the launch (`apply`)."""

  def __init__(self, images_u8, crop_szs, input_sz, include_rgb, jitter, seed, norm=None):
    assert images_u8.dtype == torch.uint8 and images_u8.is_contiguous()
    assert images_u8.dim() == 3 or (images_u8.dim() == 4 and images_u8.shape[3] == 3), \
        "[B, H, W] (mode L) or [B, H, W, 3] (RGB) uint8"
    self.images = images_u8
    self.channels = 1 if images_u8.dim() == 3 else 3
    self.B, self.H, self.W = (int(v) for v in images_u8.shape[:3])
    self.crop_szs = [int(c) for c in crop_szs]
    assert 1 <= len(self.crop_szs) <= 8 and all(0 < c <= min(self.H, self.W) for c in self.crop_szs)
    self.S, self.include_rgb = int(input_sz), bool(include_rgb)
    self.jitter = tuple(float(j) for j in jitter)
    self.rng = np.random.RandomState(seed)
    dev = images_u8.device
    tabs, bl, kl, brow, kint = [], [], [], 0, 0
    for c in self.crop_szs:
      bounds, kk = bilinear_tables(c, self.S)
      tabs.append((c, kk.shape[1], brow, kint))
      bl.append(bounds)
      kl.append(kk.reshape(-1))
      brow += bounds.shape[0]
      kint += kk.size
    self.tables_host = np.ascontiguousarray(np.asarray(tabs, dtype=np.int32))
    self.bounds = torch.from_numpy(np.concatenate(bl, 0)).to(dev)
    self.kk = torch.from_numpy(np.concatenate(kl, 0)).to(dev)
    self.lut = (torch.arange(256, dtype=torch.float32) / 255).to(dev)   # to_tensor's .div(255)
    # --demean: torchvision Normalize(mean=config.data_mean, std=config.data_std) on the tensor
    self.norm = None
    if norm is not None:
      mean, std = (np.asarray(v, dtype=np.float32).reshape(-1) for v in norm)
      C = 1 if self.channels == 1 else (4 if self.include_rgb else 1)
      assert mean.size == C and std.size == C, "data_mean / data_std need one entry per output channel"
      self.norm = torch.from_numpy(np.concatenate([mean, std])).to(dev)

  @property
  def out_channels(self):
    return 1 if self.channels == 1 else (4 if self.include_rgb else 1)

  def _jitter_draws(self, ip, fp):
    r, n = self.rng, ip.shape[0]
    b, c, s, h = self.jitter
    ip[:, 4] = 4
    ip[:, 5:9] = np.argsort(r.random_sample((n, 4)), axis=1)   # uniform random op orders
    fp[:, OP_BRIGHTNESS] = r.uniform(max(0.0, 1 - b), 1 + b, size=n)
    fp[:, OP_CONTRAST] = r.uniform(max(0.0, 1 - c), 1 + c, size=n)
    fp[:, OP_SATURATION] = r.uniform(max(0.0, 1 - s), 1 + s, size=n)
    fp[:, OP_HUE] = r.uniform(-h, h, size=n)                 # kept for the record; the kernel reads
    ip[:, 9] = [hue_shift(float(v)) for v in fp[:, OP_HUE]]  # the uint8 increment derived from it

  def _random_crops(self, ip, rows=None):
    """RandomCrop.get_params for the rows' table entries: uniform integer offsets."""
    rows = np.arange(ip.shape[0]) if rows is None else rows
    crop = np.asarray(self.crop_szs)[ip[rows, 10]]
    ip[rows, 1] = (self.rng.random_sample(len(rows)) * (self.W - crop + 1)).astype(np.int64)
    ip[rows, 2] = (self.rng.random_sample(len(rows)) * (self.H - crop + 1)).astype(np.int64)

  def _center_crops(self, ip, rows=None):
    rows = np.arange(ip.shape[0]) if rows is None else rows
    for t, c in enumerate(self.crop_szs):
      sel = rows[ip[rows, 10] == t]
      ip[sel, 1], ip[sel, 2] = _center_xy(self.W, self.H, c)

  def apply(self, iparams, fparams):
    iparams = np.ascontiguousarray(iparams, dtype=np.int32)
    fparams = np.ascontiguousarray(fparams, dtype=np.float32)
    n = int(iparams.shape[0])
    assert iparams.shape == (n, IPARAMS) and fparams.shape == (n, FPARAMS)
    assert n > 0 and iparams[:, 0].min() >= 0 and iparams[:, 0].max() < self.B, "source index out of range"
    assert iparams[:, 10].min() >= 0 and iparams[:, 10].max() < len(self.crop_szs)
    crop = np.asarray(self.crop_szs)[iparams[:, 10]]
    assert (iparams[:, 1] >= 0).all() and (iparams[:, 1] + crop <= self.W).all()
    assert (iparams[:, 2] >= 0).all() and (iparams[:, 2] + crop <= self.H).all()
    assert (iparams[:, 4] >= 0).all() and (iparams[:, 4] <= 4).all()
    dev = self.images.device
    assert self.images.is_cuda, "the dataset must be resident on the GPU (there is no CPU path)"
    ip = torch.from_numpy(iparams).to(dev, non_blocking=True)
    fp = torch.from_numpy(fparams).to(dev, non_blocking=True)
    out = torch.empty(n, self.out_channels, self.S, self.S, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().iic_augment(
      self.images.data_ptr(), self.B, self.H, self.W, self.channels, ip.data_ptr(), fp.data_ptr(), n,
      self.tables_host.ctypes.data, len(self.crop_szs), self.bounds.data_ptr(), self.kk.data_ptr(),
      self.S, self.lut.data_ptr(), out.data_ptr(), int(self.include_rgb),
      None if self.norm is None else self.norm.data_ptr(), _lib.stream_ptr()),
      "iic_augment")
    return out

  def _new_params(self, idx):
    idx = np.asarray(idx, dtype=np.int64).reshape(-1)
    ip = np.zeros((idx.shape[0], IPARAMS), np.int32)
    ip[:, 0] = idx
    return ip, np.zeros((idx.shape[0], FPARAMS), np.float32)

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


class PairedAugmenter(_AugmenterBase):
  """`sobel_make_transforms` (code/utils/cluster/transforms.py:107-217).  images_u8: uint8
  [B, H, W, 3] on the GPU (HWC, the layout torchvision datasets hold).

  plain(idx) / jittered(idx) / center(idx) return float32 [len(idx), C, input_sz, input_sz]
  (C = 4 with include_rgb, else 1) for tf1 / tf2 / tf3; `draw` exposes the parameter draws so
  that callers (and the tests) can replay them.

  Branches of the reference's tf2 (all optional, the defaults are the default branch):
    cutout (--cutout, transforms.py:170-186): with probability cutout_p a black box of side
      2*floor(b/2), b uniform in [int(0.2 crop), int(cutout_max_box crop)], is pasted on the crop;
    fluid_warp (--fluid_warp, :142-152): with probability 0.5 a RandomRotation(rot_val) of the
      whole image (PIL NEAREST) first, then a RandomCrop whose size is chosen uniformly from
      rand_crop_szs_tf;
    demean (--demean, :196-204): Normalize(data_mean, data_std) at the end of tf1 / tf2 / tf3.
  per_img_demean (:98-104, :206-212) asserts a 3-channel tensor and therefore cannot run behind
  custom_greyscale_to_tensor (1 or 4 channels) in the reference either: not built.  random_affine
  is never switched on by the reference's data layer (data.py calls sobel_make_transforms(config)).
  """

  def __init__(self, images_u8, rand_crop_sz, input_sz, include_rgb, jitter=(0.4, 0.4, 0.4, 0.125),
               seed=0, cutout=False, cutout_p=0.5, cutout_max_box=0.5, fluid_warp=False, rot_val=0.0,
               rand_crop_szs_tf=(), demean=False, data_mean=(), data_std=()):
    assert images_u8.dim() == 4, "RGB dataset [B, H, W, 3]"
    assert not (cutout and fluid_warp), "transforms.py:172"
    szs = [int(rand_crop_sz)]
    self.tf2_tables = [0]
    if fluid_warp:
      assert len(rand_crop_szs_tf) > 0
      self.tf2_tables = []
      for c in rand_crop_szs_tf:
        if int(c) not in szs:
          szs.append(int(c))
        self.tf2_tables.append(szs.index(int(c)))
    super(PairedAugmenter, self).__init__(images_u8, szs, input_sz, include_rgb, jitter, seed,
                                          norm=(data_mean, data_std) if demean else None)
    self.crop = int(rand_crop_sz)
    self.cutout, self.cutout_p = bool(cutout), float(cutout_p)
    self.cut_min, self.cut_max = int(self.crop * 0.2), int(self.crop * float(cutout_max_box))
    self.fluid_warp, self.rot_val = bool(fluid_warp), float(rot_val)

  def _cutout_draws(self, ip):
    """custom_cutout (transforms.py:28-44) under RandomApply(p): box side, centre."""
    r, n = self.rng, ip.shape[0]
    do = r.random_sample(n) < self.cutout_p
    self.last_cutout = np.zeros((n, 4), np.int64)
    for i in np.nonzero(do)[0]:
      box_sz = r.randint(self.cut_min, self.cut_max + 1)
      half = int(np.floor(box_sz / 2.))
      crop = self.crop_szs[ip[i, 10]]
      x_c = r.randint(half, crop - half)
      y_c = r.randint(half, crop - half)
      box = (x_c - half, y_c - half, x_c + half, y_c + half)
      self.last_cutout[i] = box
      if box[2] > box[0] and box[3] > box[1]:
        ip[i, 18] = box[0] | (box[1] << 16)
        ip[i, 19] = box[2] | (box[3] << 16)

  def draw(self, idx, mode):
    """mode 'plain' (tf1), 'jittered' (tf2) or 'center' (tf3).  Returns (iparams int32 [n, 20],
    fparams float32 [n, 4]) as iic_augment reads them."""
    ip, fp = self._new_params(idx)
    n = ip.shape[0]
    if mode == "center":
      self._center_crops(ip)
      return ip, fp
    if mode == "plain":
      self._random_crops(ip)
      return ip, fp
    assert mode == "jittered"
    self.last_angles = np.full(n, np.nan)
    if self.fluid_warp:
      do = self.rng.random_sample(n) < 0.5
      ang = self.rng.uniform(-self.rot_val, self.rot_val, size=n)
      for i in np.nonzero(do)[0]:
        if ang[i] % 360.0 != 0:                     # PIL's angle-0 fast path is a plain copy
          ip[i, 11] = 1
          ip[i, 12:18] = rotation_fixed_point(float(ang[i]), self.W, self.H)
          self.last_angles[i] = ang[i]
      ip[:, 10] = np.asarray(self.tf2_tables)[self.rng.randint(0, len(self.tf2_tables), size=n)]
    self._random_crops(ip)
    if self.cutout:
      self._cutout_draws(ip)
    ip[:, 3] = self.rng.random_sample(n) < 0.5
    self._jitter_draws(ip, fp)
    return ip, fp


class GreyscaleAugmenter(_AugmenterBase):
  """`greyscale_make_transforms` (code/utils/cluster/transforms.py:220-330; the MNIST scripts).
  images_u8: uint8 [B, H, W] (mode L) on the GPU; `config` carries the reference's flags:
  crop_orig, tf1_crop ('random' | 'centre_half' | 'centre'), tf1_crop_sz, tf3_crop_diff,
  tf3_crop_sz, rot_val, always_rot, crop_other, tf2_crop, tf2_crop_szs, input_sz, no_flip,
  no_jitter, demean + data_mean / data_std (Normalize).  per_img_demean asserts a 3-channel tensor
  (transforms.py:99) and cannot run on these 1-channel images in the reference either: refused."""

  def __init__(self, images_u8, config, jitter=(0.4, 0.4, 0.4, 0.125), seed=0):
    assert images_u8.dim() == 3, "mode-L dataset [B, H, W]"
    if getattr(config, "per_img_demean", False):
      raise NotImplementedError("per_img_demean asserts a 3-channel tensor (transforms.py:99): it cannot "
                                "run on the greyscale pipelines in the reference either")
    norm = (config.data_mean, config.data_std) if getattr(config, "demean", False) else None
    H, W = int(images_u8.shape[1]), int(images_u8.shape[2])
    full = min(H, W)
    szs = []

    def table(sz):
      sz = int(sz)
      if sz not in szs:
        szs.append(sz)
      return szs.index(sz)
    if config.crop_orig:
      assert config.tf1_crop in ("random", "centre_half", "centre")
      self.t1 = table(config.tf1_crop_sz)
      self.t3 = table(config.tf3_crop_sz if config.tf3_crop_diff else config.tf1_crop_sz)
    else:               # no crop: Resize(input_sz) of the whole (square) image
      assert H == W, "Resize(int) of a non-square image keeps the aspect ratio: not built"
      self.t1 = self.t3 = table(full)
    if config.crop_other:
      assert config.tf2_crop in ("random", "centre_half", "centre")
      self.t2 = [table(c) for c in config.tf2_crop_szs]
    else:
      assert H == W
      self.t2 = [table(full)]
    self.cfg = config
    super(GreyscaleAugmenter, self).__init__(images_u8, szs, config.input_sz, False, jitter, seed, norm=norm)

  def _crop_kind(self, ip, rows, kind):
    if kind == "random":
      self._random_crops(ip, rows)
    elif kind == "centre":
      self._center_crops(ip, rows)
    else:               # RandomChoice([RandomCrop, CenterCrop])
      pick = self.rng.random_sample(len(rows)) < 0.5
      self._random_crops(ip, rows[pick])
      self._center_crops(ip, rows[~pick])

  def draw(self, idx, mode):
    cfg = self.cfg
    ip, fp = self._new_params(idx)
    n = ip.shape[0]
    rows = np.arange(n)
    if mode == "center":
      ip[:, 10] = self.t3
      self._center_crops(ip)
      return ip, fp
    if mode == "plain":
      ip[:, 10] = self.t1
      if cfg.crop_orig:
        self._crop_kind(ip, rows, cfg.tf1_crop)
      return ip, fp
    assert mode == "jittered"
    self.last_angles = np.full(n, np.nan)          # the rotation draws behind ip[:, 11:18] (replay / tests)
    if cfg.rot_val > 0:
      do = np.ones(n, bool) if cfg.always_rot else self.rng.random_sample(n) < 0.5
      ang = self.rng.uniform(-cfg.rot_val, cfg.rot_val, size=n)
      for i in np.nonzero(do)[0]:
        if ang[i] % 360.0 != 0:                     # PIL's angle-0 fast path is a plain copy
          ip[i, 11] = 1
          ip[i, 12:18] = rotation_fixed_point(float(ang[i]), self.W, self.H)
          self.last_angles[i] = ang[i]
    ip[:, 10] = np.asarray(self.t2)[self.rng.randint(0, len(self.t2), size=n)]
    if cfg.crop_other:
      self._crop_kind(ip, rows, cfg.tf2_crop)
    if not cfg.no_flip:
      ip[:, 3] = self.rng.random_sample(n) < 0.5
    if not cfg.no_jitter:
      self._jitter_draws(ip, fp)
    return ip, fp


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
