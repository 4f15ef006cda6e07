"""CPU restatement (TEST INFRASTRUCTURE, never imported by iic_amd/) of the reference's paired
augmentation for the clustering scripts, /root/reference/code/utils/cluster/transforms.py:107-217
(`sobel_make_transforms`, default branch: no fluid_warp / cutout / random affine / demean):

  tf1: RandomCrop(rand_crop_sz) -> Resize(input_sz) -> custom_greyscale_to_tensor(include_rgb)
  tf2: RandomCrop(rand_crop_sz) -> Resize(input_sz) -> RandomHorizontalFlip ->
       ColorJitter(0.4, 0.4, 0.4, 0.125) -> custom_greyscale_to_tensor(include_rgb)
  tf3: CenterCrop(rand_crop_sz) -> Resize(input_sz) -> custom_greyscale_to_tensor(include_rgb)

and of `greyscale_make_transforms` (transforms.py:220-330; the MNIST scripts, mode "L" images):

  tf1: {RandomCrop | CenterCrop | RandomChoice of both}(tf1_crop_sz) -> Resize(input_sz) -> ToTensor
  tf2: [RandomApply(RandomRotation(rot_val), 0.5)] -> RandomChoice(RandomCrop(sz) for sz in
       tf2_crop_szs) -> Resize -> [RandomHorizontalFlip] -> [ColorJitter] -> ToTensor
  tf3: CenterCrop -> Resize -> ToTensor

The reference composes torchvision 0.2.1 transforms (package_versions.txt), which are thin
wrappers over PIL.  torchvision is absent here; its functional ops are restated below on PIL
itself (present: the arithmetic that matters -- bilinear resampling, ImageEnhance blends, the
HSV round trip, the L conversion -- is PIL's own code), with the RANDOM PARAMETERS made explicit
so that a device implementation can be compared on identical parameters:

PINNING.  Pinned to the reference's OWN transform code: oracle/gen_golden_augment.py imports
code/utils/cluster/transforms.py through the Python-2 hook and executes its sobel_make_transforms /
greyscale_make_transforms / custom_greyscale_to_tensor / custom_cutout on oracle/tv021_shim.py (torchvision
0.2.1's transforms restated over PIL, every random draw logged); tests/golden/augment.npz holds the
tensors its Compose objects returned for 8 flag sets x 6 images with the draw logs, and
tests/test_augment_golden_cpu.py replays those draws through pil_pipeline / np_pipeline below
(`params_from_log`) and demands bit-equal pixels -- including the --cutout, --fluid_warp and --demean
branches -- and compares the draw distributions field by field.  What is left unpinned is the Pillow
version underneath: the shim and this oracle call the Pillow installed here (12.2.0) for resize / rotate /
ImageEnhance / the HSV conversion, the reference's environment pins Pillow 5.2.0
(package_versions.txt:73,115; not available offline).  Drift between those two Pillow versions in the
four operations is NOT checked: with respect to the reference's pinned Pillow this part stays
"parity unpinned".

  pil_pipeline(...)   the reference's op sequence on PIL images            (the oracle)
  np_pipeline(...)    the same arithmetic in numpy integer / float32 / float64 steps, i.e. the
                      algorithm specification of csrc/augment.hip; tests check it against
                      pil_pipeline bit for bit on CPU.
"""
import math

import numpy as np
from PIL import Image, ImageEnhance

OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE = 0, 1, 2, 3
PRECISION_BITS = 32 - 8 - 2


# ------------------------------------------------------------------------------------------
# torchvision.transforms.functional (0.2.1) on PIL
# ------------------------------------------------------------------------------------------
def tv_adjust_hue(img, hue_factor):
  assert -0.5 <= hue_factor <= 0.5
  h, s, v = img.convert("HSV").split()
  np_h = np.array(h, dtype=np.uint8)
  np_h = (np_h.astype(np.int64) + hue_delta(hue_factor)).astype(np.uint8)   # uint8 wrap-around add
  h = Image.fromarray(np_h, "L")
  return Image.merge("HSV", (h, s, v)).convert("RGB")


def hue_delta(hue_factor):
  """np.uint8(hue_factor * 255) of torchvision's adjust_hue: C cast = truncation toward zero,
  then modulo 256."""
  return int(hue_factor * 255) % 256


def _normalize(t, norm):
  """torchvision 0.2.1 Normalize: t.sub_(m).div_(s) per channel, float32."""
  if norm is None:
    return t
  mean, std = (np.asarray(v, dtype=np.float32).reshape(-1, 1, 1) for v in norm)
  return ((t - mean) / std).astype(np.float32)


def pil_pipeline(img_u8, crop_xy, crop_sz, out_sz, include_rgb, flip=False, order=(), factors=None,
                 angle=None, cutout_box=None, norm=None):
  """img_u8: HWC uint8 RGB (sobel pipelines) or HW uint8 (mode "L", greyscale pipelines).
  order: sequence of OP_* (the shuffled ColorJitter order); factors: dict op -> factor
  (brightness / contrast / saturation factors, hue_factor); angle: RandomRotation's draw in degrees
  (F.rotate(img, angle, resample=False, expand=False, center=None)) or None; cutout_box:
  (left, upper, right, lower) of custom_cutout's img.paste(0, box) on the crop
  (transforms.py:28-44) or None; norm: (data_mean, data_std) of the trailing Normalize or None.
  Returns float32 [C, out_sz, out_sz]: custom_greyscale_to_tensor for RGB input, ToTensor for L."""
  img = Image.fromarray(img_u8)
  if angle is not None:
    img = img.rotate(angle, Image.NEAREST, False, None)                     # F.rotate
  x0, y0 = crop_xy
  img = img.crop((x0, y0, x0 + crop_sz, y0 + crop_sz))                     # F.crop
  if cutout_box is not None:
    img.paste(0, box=tuple(int(v) for v in cutout_box))                    # custom_cutout
  img = img.resize((out_sz, out_sz), Image.BILINEAR)                       # F.resize
  if flip:
    img = img.transpose(Image.FLIP_LEFT_RIGHT)                             # F.hflip
  for op in order:
    f = factors[op]
    if op == OP_BRIGHTNESS:
      img = ImageEnhance.Brightness(img).enhance(f)
    elif op == OP_CONTRAST:
      img = ImageEnhance.Contrast(img).enhance(f)
    elif op == OP_SATURATION:
      img = ImageEnhance.Color(img).enhance(f)
    elif img.mode != "L":                  # adjust_hue returns mode L / 1 / I / F images unchanged
      img = tv_adjust_hue(img, f)
  if img.mode == "L":                                                       # ToTensor
    return _normalize((np.asarray(img).astype(np.float32) / np.float32(255))[None], norm)
  grey = np.asarray(img.convert("L")).astype(np.float32) / np.float32(255)  # to_tensor: .float().div(255)
  if not include_rgb:
    return _normalize(grey[None], norm)
  rgb = np.transpose(np.asarray(img).astype(np.float32) / np.float32(255), (2, 0, 1))
  return _normalize(np.concatenate([rgb, grey[None]], 0), norm)


def center_crop_xy(w, h, crop_sz):
  """torchvision 0.2.1 F.center_crop: i = int(round((h - th) / 2.)), j = int(round((w - tw) / 2.))."""
  return int(round((w - crop_sz) / 2.)), int(round((h - crop_sz) / 2.))


# ------------------------------------------------------------------------------------------
# the same arithmetic, step by step (specification of the HIP kernel)
# ------------------------------------------------------------------------------------------
def resample_coeffs(in_size, out_size):
  """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1):
  per output index (xmin, count) and `count` fixed-point weights (22 fractional bits)."""
  scale = in_size / out_size
  fscale = max(scale, 1.0)
  support = 1.0 * fscale
  ksize = int(math.ceil(support)) * 2 + 1
  bounds = np.zeros((out_size, 2), dtype=np.int32)
  kk = np.zeros((out_size, ksize), dtype=np.int32)
  for xx in range(out_size):
    center = (xx + 0.5) * scale
    ss = 1.0 / fscale
    xmin = max(int(center - support + 0.5), 0)
    xmax = min(int(center + support + 0.5), in_size) - xmin
    w = []
    for x in range(xmax):
      t = abs((x + xmin - center + 0.5) * ss)
      w.append(1.0 - t if t < 1.0 else 0.0)
    ww = sum(w)
    for x in range(xmax):
      wn = w[x] / ww if ww != 0.0 else w[x]
      kk[xx, x] = int(-0.5 + wn * (1 << PRECISION_BITS)) if wn < 0 else int(0.5 + wn * (1 << PRECISION_BITS))
    bounds[xx] = (xmin, xmax)
  return bounds, kk


def _resize1d(a, out_size, axis):
  a = np.moveaxis(a, axis, 0).astype(np.int64)
  bounds, kk = resample_coeffs(a.shape[0], out_size)
  out = np.zeros((out_size,) + a.shape[1:], dtype=np.int64)
  for xx, (xmin, cnt) in enumerate(bounds):
    ss = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
    for x in range(cnt):
      ss += a[xmin + x] * int(kk[xx, x])
    out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255)
  return np.moveaxis(out, 0, axis)


def np_resize(a, out_sz):
  return _resize1d(_resize1d(a, out_sz, 1), out_sz, 0)      # horizontal pass, then vertical


def np_luma(a):
  r, g, b = (a[..., i].astype(np.int64) for i in range(3))
  return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16


def np_blend(in1, in2, alpha):
  """Pillow ImagingBlend: (float)((int)in1 + alpha * ((int)in2 - (int)in1)), float32, truncated,
  clipped to [0, 255]."""
  a = np.float32(alpha)
  d = (in2.astype(np.int64) - in1.astype(np.int64)).astype(np.float32)
  t = in1.astype(np.float32) + a * d
  return np.clip(t, 0, 255).astype(np.int64)               # astype truncates toward zero (t >= 0 here)


def np_rgb2hsv(a):
  r, g, b = (a[..., i].astype(np.int32) for i in range(3))
  maxc = np.maximum(r, np.maximum(g, b))
  minc = np.minimum(r, np.minimum(g, b))
  f32, f64 = np.float32, np.float64
  with np.errstate(divide="ignore", invalid="ignore"):
    cr = (maxc - minc).astype(f32)
    s = cr / maxc.astype(f32)
    rc = (maxc - r).astype(f32) / cr
    gc = (maxc - g).astype(f32) / cr
    bc = (maxc - b).astype(f32) / cr
    h = np.where(r == maxc, bc.astype(f64) - gc.astype(f64),
                 np.where(g == maxc, 2.0 + rc.astype(f64) - bc.astype(f64),
                          4.0 + gc.astype(f64) - rc.astype(f64))).astype(f32)
    h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(f64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(f64) * 255.0).astype(np.int64), 0, 255)
  grey = maxc == minc
  return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc.astype(np.int64)], -1)


def np_hsv2rgb(a):
  h, s, v = (a[..., i].astype(np.float64) for i in range(3))
  hh = h * 6.0 / 255.0
  i = np.floor(hh).astype(np.int64)
  f = (hh - i).astype(np.float32).astype(np.float64)
  fs = (s / 255.0).astype(np.float32).astype(np.float64)

  def rnd(x):
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)
  p = np.clip(rnd(v * (1.0 - fs)), 0, 255)
  q = np.clip(rnd(v * (1.0 - fs * f)), 0, 255)
  t = np.clip(rnd(v * (1.0 - fs * (1.0 - f))), 0, 255)
  V = v.astype(np.int64)
  i6 = i % 6
  R = np.choose(i6, [V, q, p, p, t, V])
  G = np.choose(i6, [t, V, V, q, p, p])
  B = np.choose(i6, [p, p, t, V, V, q])
  s0 = a[..., 1] == 0
  return np.stack([np.where(s0, V, R), np.where(s0, V, G), np.where(s0, V, B)], -1)


def rotation_coeffs(angle, w, h):
  """PIL Image.rotate's inverse affine matrix (python doubles, cos / sin rounded to 15 digits,
  centre (w/2, h/2)) and ImagingTransformAffine's 16.16 fixed-point form of it (nearest filter):
  source x = (a2 + a1*y + a0*x) >> 16, source y = (a5 + a4*y + a3*x) >> 16."""
  a = -math.radians(angle % 360.0)
  m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
  cx, cy = w / 2, h / 2
  m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2] + cx
  m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5] + cy

  def fix(v):
    return int(math.floor(v * 65536.0 + 0.5))
  return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5),
          fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def np_rotate(a, angle):
  if angle % 360.0 == 0:
    return a                                                 # PIL fast path: a copy
  h, w = a.shape[:2]
  a0, a1, a2, a3, a4, a5 = rotation_coeffs(angle, w, h)
  y, x = np.mgrid[0:h, 0:w]
  xin, yin = (a2 + a1 * y + a0 * x) >> 16, (a5 + a4 * y + a3 * x) >> 16
  ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
  out = np.zeros_like(a)
  out[ok] = a[yin[ok], xin[ok]]
  return out


def np_pipeline(img_u8, crop_xy, crop_sz, out_sz, include_rgb, flip=False, order=(), factors=None,
                angle=None, cutout_box=None, norm=None):
  x0, y0 = crop_xy
  grey_in = img_u8.ndim == 2
  a = img_u8[..., None] if grey_in else img_u8
  if angle is not None:
    a = np_rotate(a, angle)
  a = a[y0:y0 + crop_sz, x0:x0 + crop_sz].astype(np.int64)
  if cutout_box is not None:
    l, u, r, lo = (int(v) for v in cutout_box)
    a = a.copy()
    a[max(u, 0):lo, max(l, 0):r] = 0
  a = np_resize(a, out_sz)
  if flip:
    a = a[:, ::-1]
  for op in order:
    f = factors[op]
    if op == OP_BRIGHTNESS:
      a = np_blend(np.zeros_like(a), a, f)
    elif op == OP_CONTRAST:
      L = a[..., 0] if grey_in else np_luma(a)
      mean = int(float(L.sum()) / float(L.size) + 0.5)      # int(ImageStat.Stat(L).mean[0] + 0.5)
      a = np_blend(np.full_like(a, mean), a, f)
    elif grey_in:
      pass       # mode L: Color's degenerate is the image itself (blend = identity), hue is skipped
    elif op == OP_SATURATION:
      L = np_luma(a)
      a = np_blend(np.repeat(L[..., None], 3, -1), a, f)
    else:
      hsv = np_rgb2hsv(a)
      hsv[..., 0] = (hsv[..., 0] + hue_delta(f)) % 256
      a = np_hsv2rgb(hsv)
  lut = np.arange(256, dtype=np.float32) / np.float32(255)
  if grey_in:
    return _normalize(lut[a[..., 0]][None], norm)
  grey = lut[np_luma(a)]
  if not include_rgb:
    return _normalize(grey[None], norm)
  return _normalize(np.concatenate([np.transpose(lut[a], (2, 0, 1)), grey[None]], 0), norm)


def params_from_log(log, src_hw):
  """Translate the draw log of oracle/tv021_shim.py (the random draws the reference's own Compose objects
  made for one image, tests/golden/augment.npz) into pil_pipeline / np_pipeline keyword arguments."""
  H, W = src_hw
  kw = dict(flip=False, order=(), factors=None, angle=None, cutout_box=None)
  for kind, val in log:
    if kind == "angle":
      kw["angle"] = float(val)
    elif kind == "crop":
      kw["crop_xy"], kw["crop_sz"] = (int(val[0]), int(val[1])), int(val[2])
    elif kind == "center_crop":
      kw["crop_sz"] = int(val)
      kw["crop_xy"] = center_crop_xy(W, H, int(val))
    elif kind == "paste":
      kw["cutout_box"] = tuple(int(v) for v in val)
    elif kind == "flip":
      kw["flip"] = bool(val)
    elif kind == "jitter":
      kw["order"] = [int(op) for op, _ in val]
      kw["factors"] = {int(op): float(f) for op, f in val}
    else:
      assert kind in ("apply", "choice"), kind         # implied by the entries that follow them
  return kw


def random_params(rng, n, src_hw, crop_sz, jitter=(0.4, 0.4, 0.4, 0.125)):
  """Explicit parameter draws with the distributions of RandomCrop.get_params,
  RandomHorizontalFlip and ColorJitter.get_params (torchvision 0.2.1)."""
  H, W = src_hw
  out = []
  for _ in range(n):
    x0 = int(rng.integers(0, W - crop_sz + 1))
    y0 = int(rng.integers(0, H - crop_sz + 1))
    flip = bool(rng.random() < 0.5)
    b, c, s, h = jitter
    factors = {OP_BRIGHTNESS: float(rng.uniform(max(0, 1 - b), 1 + b)),
               OP_CONTRAST: float(rng.uniform(max(0, 1 - c), 1 + c)),
               OP_SATURATION: float(rng.uniform(max(0, 1 - s), 1 + s)),
               OP_HUE: float(rng.uniform(-h, h))}
    order = [int(o) for o in rng.permutation(4)]
    out.append(dict(crop_xy=(x0, y0), flip=flip, order=order, factors=factors))
  return out
