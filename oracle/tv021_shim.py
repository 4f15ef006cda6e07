"""torchvision 0.2.1 `transforms` / `transforms.functional` restated over the PIL installed here.
TEST INFRASTRUCTURE (never imported by iic_amd/): it exists so that the reference's OWN augmentation
code -- /root/reference/code/utils/cluster/transforms.py:12-44,107-334 (`custom_greyscale_to_tensor`,
`custom_cutout`, `sobel_make_transforms`, `greyscale_make_transforms`) -- can be imported and executed in
this container (oracle/gen_golden_augment.py), which pins the op composition, parameters and draw order
of the fixtures in tests/golden/augment.npz to the reference instead of to a restatement of it.

Why a shim: the reference pins torchvision 0.2.1 + Pillow 5.2.0 (package_versions.txt:73,115); neither
is installable offline.  torchvision's transforms are thin wrappers -- every pixel operation below is a
call into PIL (crop / resize / rotate / transpose / ImageEnhance / convert), restated from the published
0.2.1 sources (torchvision/transforms/transforms.py, functional.py, tag v0.2.1): class by class the
same control flow, the same random draws from the same generators (python's `random` for RandomCrop /
RandomApply / RandomChoice / RandomRotation / RandomHorizontalFlip, `numpy.random` for ColorJitter,
as 0.2.1 did), nothing added.  What stays unpinned is the Pillow version itself (12.2.0 here vs 5.2.0).

Every random draw is appended to `LOG` as (kind, value) in call order, so that the fixture can store
the parameters next to the pixels: the oracle and the HIP kernel are then checked on IDENTICAL draws.
"""
import numbers
import random
import sys
import types

import numpy as np
import torch
from PIL import Image, ImageEnhance

LOG = []


def _log(kind, value):
  LOG.append((kind, value))


# ------------------------------------------------------------------------------------------
# functional.py (0.2.1)
# ------------------------------------------------------------------------------------------
def _is_pil_image(img):
  return isinstance(img, Image.Image)


def to_tensor(pic):
  """PIL image -> float tensor CHW in [0, 1] (uint8 modes: .float().div(255))."""
  assert _is_pil_image(pic)
  if pic.mode == "I":
    img = torch.from_numpy(np.array(pic, np.int32, copy=False))
  elif pic.mode == "I;16":
    img = torch.from_numpy(np.array(pic, np.int16, copy=False))
  elif pic.mode == "F":
    img = torch.from_numpy(np.array(pic, np.float32, copy=False))
  else:
    img = torch.frombuffer(bytearray(pic.tobytes()), dtype=torch.uint8)
  if pic.mode == "YCbCr":
    nchannel = 3
  elif pic.mode == "I;16":
    nchannel = 1
  else:
    nchannel = len(pic.mode)
  img = img.view(pic.size[1], pic.size[0], nchannel)
  img = img.transpose(0, 1).transpose(0, 2).contiguous()
  if isinstance(img, torch.ByteTensor) or img.dtype == torch.uint8:
    return img.float().div(255)
  return img


def normalize(tensor, mean, std):
  for t, m, s in zip(tensor, mean, std):
    t.sub_(m).div_(s)
  return tensor


def resize(img, size, interpolation=Image.BILINEAR):
  if isinstance(size, int):
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
      return img
    if w < h:
      ow = size
      oh = int(size * h / w)
      return img.resize((ow, oh), interpolation)
    oh = size
    ow = int(size * w / h)
    return img.resize((ow, oh), interpolation)
  return img.resize(tuple(int(v) for v in size[::-1]), interpolation)


def crop(img, i, j, h, w):
  return img.crop((j, i, j + w, i + h))


def center_crop(img, output_size):
  if isinstance(output_size, numbers.Number):
    output_size = (int(output_size), int(output_size))
  w, h = img.size
  th, tw = output_size
  i = int(round((h - th) / 2.))
  j = int(round((w - tw) / 2.))
  return crop(img, i, j, th, tw)


def hflip(img):
  return img.transpose(Image.FLIP_LEFT_RIGHT)


def adjust_brightness(img, brightness_factor):
  return ImageEnhance.Brightness(img).enhance(brightness_factor)


def adjust_contrast(img, contrast_factor):
  return ImageEnhance.Contrast(img).enhance(contrast_factor)


def adjust_saturation(img, saturation_factor):
  return ImageEnhance.Color(img).enhance(saturation_factor)


def adjust_hue(img, hue_factor):
  if not (-0.5 <= hue_factor <= 0.5):
    raise ValueError("hue_factor is not in [-0.5, 0.5].")
  input_mode = img.mode
  if input_mode in {"L", "1", "I", "F"}:
    return img
  h, s, v = img.convert("HSV").split()
  np_h = np.array(h, dtype=np.uint8)
  with np.errstate(over="ignore"):
    # 0.2.1: np_h += np.uint8(hue_factor * 255) -- a C cast (truncation) then a wrap-around uint8 add;
    # numpy 2 refuses the out-of-range scalar conversion, so the same arithmetic is spelt out
    np_h = (np_h.astype(np.int64) + (int(hue_factor * 255) % 256)).astype(np.uint8)
  h = Image.fromarray(np_h, "L")
  return Image.merge("HSV", (h, s, v)).convert(input_mode)


def rotate(img, angle, resample=False, expand=False, center=None):
  return img.rotate(angle, resample, expand, center)


def to_grayscale(img, num_output_channels=1):
  if num_output_channels == 1:
    return img.convert("L")
  if num_output_channels == 3:
    img = img.convert("L")
    np_img = np.array(img, dtype=np.uint8)
    return Image.fromarray(np.dstack([np_img, np_img, np_img]), "RGB")
  raise ValueError("num_output_channels should be either 1 or 3")


# ------------------------------------------------------------------------------------------
# transforms.py (0.2.1)
# ------------------------------------------------------------------------------------------
class Compose(object):
  def __init__(self, transforms):
    self.transforms = transforms

  def __call__(self, img):
    for t in self.transforms:
      img = t(img)
    return img


class ToTensor(object):
  def __call__(self, pic):
    return to_tensor(pic)


class Normalize(object):
  def __init__(self, mean, std):
    self.mean, self.std = mean, std

  def __call__(self, tensor):
    return normalize(tensor, self.mean, self.std)


class Resize(object):
  def __init__(self, size, interpolation=Image.BILINEAR):
    self.size, self.interpolation = size, interpolation

  def __call__(self, img):
    return resize(img, self.size, self.interpolation)


class CenterCrop(object):
  def __init__(self, size):
    self.size = (int(size), int(size)) if isinstance(size, numbers.Number) else size

  def __call__(self, img):
    _log("center_crop", int(self.size[0]))
    return center_crop(img, self.size)


class Lambda(object):
  def __init__(self, lambd):
    self.lambd = lambd

  def __call__(self, img):
    return self.lambd(img)


class RandomApply(object):
  def __init__(self, transforms, p=0.5):
    self.transforms, self.p = transforms, p

  def __call__(self, img):
    skip = self.p < random.random()
    _log("apply", 0 if skip else 1)
    if skip:
      return img
    for t in self.transforms:
      img = t(img)
    return img


class RandomChoice(object):
  def __init__(self, transforms):
    self.transforms = transforms

  def __call__(self, img):
    t = random.choice(self.transforms)
    _log("choice", self.transforms.index(t))
    return t(img)


class RandomCrop(object):
  def __init__(self, size, padding=0):
    self.size = (int(size), int(size)) if isinstance(size, numbers.Number) else size
    self.padding = padding

  @staticmethod
  def get_params(img, output_size):
    w, h = img.size
    th, tw = output_size
    if w == tw and h == th:
      return 0, 0, h, w
    i = random.randint(0, h - th)
    j = random.randint(0, w - tw)
    return i, j, th, tw

  def __call__(self, img):
    assert self.padding == 0
    i, j, h, w = self.get_params(img, self.size)
    _log("crop", (int(j), int(i), int(h)))          # (x0, y0, size)
    return crop(img, i, j, h, w)


class RandomHorizontalFlip(object):
  def __init__(self, p=0.5):
    self.p = p

  def __call__(self, img):
    flip = random.random() < self.p
    _log("flip", 1 if flip else 0)
    return hflip(img) if flip else img


class RandomRotation(object):
  def __init__(self, degrees, resample=False, expand=False, center=None):
    if isinstance(degrees, numbers.Number):
      if degrees < 0:
        raise ValueError("If degrees is a single number, it must be positive.")
      self.degrees = (-degrees, degrees)
    else:
      self.degrees = degrees
    self.resample, self.expand, self.center = resample, expand, center

  @staticmethod
  def get_params(degrees):
    return random.uniform(degrees[0], degrees[1])

  def __call__(self, img):
    angle = self.get_params(self.degrees)
    _log("angle", float(angle))
    return rotate(img, angle, self.resample, self.expand, self.center)


class ColorJitter(object):
  def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
    self.brightness, self.contrast, self.saturation, self.hue = brightness, contrast, saturation, hue

  @staticmethod
  def get_params(brightness, contrast, saturation, hue):
    transforms = []
    if brightness > 0:
      brightness_factor = np.random.uniform(max(0, 1 - brightness), 1 + brightness)
      transforms.append((0, brightness_factor, Lambda(lambda img: adjust_brightness(img, brightness_factor))))
    if contrast > 0:
      contrast_factor = np.random.uniform(max(0, 1 - contrast), 1 + contrast)
      transforms.append((1, contrast_factor, Lambda(lambda img: adjust_contrast(img, contrast_factor))))
    if saturation > 0:
      saturation_factor = np.random.uniform(max(0, 1 - saturation), 1 + saturation)
      transforms.append((2, saturation_factor, Lambda(lambda img: adjust_saturation(img, saturation_factor))))
    if hue > 0:
      hue_factor = np.random.uniform(-hue, hue)
      transforms.append((3, hue_factor, Lambda(lambda img: adjust_hue(img, hue_factor))))
    np.random.shuffle(transforms)
    _log("jitter", [(int(op), float(f)) for op, f, _ in transforms])     # in application order
    return Compose([t for _, _, t in transforms])

  def __call__(self, img):
    return self.get_params(self.brightness, self.contrast, self.saturation, self.hue)(img)


class RandomAffine(object):
  """Only constructed (never called) by the data layer's configurations: sobel_make_transforms builds it
  under random_affine=True, which code/utils/cluster/data.py never passes."""

  def __init__(self, *a, **k):
    pass

  def __call__(self, img):
    raise NotImplementedError("RandomAffine: not reached by the reference's clustering data layer")


def install():
  """Register the shim as `torchvision`, `torchvision.transforms`, `torchvision.transforms.functional`
  and an empty `torchvision.datasets` in sys.modules (idempotent).  Returns the functional module."""
  me = sys.modules[__name__]
  tv = types.ModuleType("torchvision")
  tr = types.ModuleType("torchvision.transforms")
  fn = types.ModuleType("torchvision.transforms.functional")
  ds = types.ModuleType("torchvision.datasets")
  for n in ("Compose", "ToTensor", "Normalize", "Resize", "CenterCrop", "Lambda", "RandomApply", "RandomChoice",
            "RandomCrop", "RandomHorizontalFlip", "RandomRotation", "ColorJitter", "RandomAffine"):
    setattr(tr, n, getattr(me, n))
  for n in ("to_tensor", "normalize", "resize", "crop", "center_crop", "hflip", "adjust_brightness",
            "adjust_contrast", "adjust_saturation", "adjust_hue", "rotate", "to_grayscale"):
    setattr(fn, n, getattr(me, n))

  class _NoDataset(object):
    pass
  for n in ("STL10", "CIFAR10", "CIFAR100", "MNIST"):
    setattr(ds, n, type(n, (_NoDataset,), {}))
  tr.functional = fn
  tv.transforms, tv.datasets = tr, ds
  tv.__version__ = "0.2.1-shim"
  sys.modules["torchvision"] = tv
  sys.modules["torchvision.transforms"] = tr
  sys.modules["torchvision.transforms.functional"] = fn
  sys.modules["torchvision.datasets"] = ds
  return fn
