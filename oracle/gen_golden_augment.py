"""Generates tests/golden/augment.npz by EXECUTING the reference's own augmentation code
(/root/reference/code/utils/cluster/transforms.py:12-44,107-334: custom_greyscale_to_tensor,
custom_cutout, sobel_make_transforms, greyscale_make_transforms), imported read-only through the
Python-2 hook, on top of oracle/tv021_shim.py (torchvision 0.2.1 restated over the installed PIL;
torchvision itself is not installable here).  Run in the build container (where /root/reference exists):

    python oracle/gen_golden_augment.py

Per configuration (the flag sets of examples/commands.txt plus the --cutout / --fluid_warp / --demean
branches) and per sample the fixture stores the uint8 source image, the float32 tensors tf1 / tf2 / tf3
returned by the reference's Compose objects, and every random draw the shim made for them, in call
order -- so tests/test_augment_golden_cpu.py (oracle, numpy specification) and tests/test_gpu_augment.py
(HIP kernel) can replay IDENTICAL draws and demand bit-equal pixels.
Seeds: python `random` and numpy's global RNG are seeded per (config, sample); see `seed_for`.
"""
import os
import random
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("IIC_REFERENCE", "/root/reference")

from oracle import tv021_shim  # noqa: E402

tv021_shim.install()
from iic_amd import py2compat  # noqa: E402

py2compat.enable(REF)
import code.utils.cluster.transforms as ref_tf  # noqa: E402  (the reference's module, Python-2 source)

from PIL import Image  # noqa: E402

NS = types.SimpleNamespace

# name -> (kind, config, source size, channels)
CONFIGS = {
  # STL10 commands (commands.txt:18-22,50): --crop_orig --rand_crop_sz 64 --input_sz 64, no include_rgb
  "stl10": ("sobel", NS(crop_orig=True, rand_crop_sz=64, input_sz=64, include_rgb=False, fluid_warp=False,
                        cutout=False, demean=False, per_img_demean=False), 96, 3),
  # CIFAR100-20 command (commands.txt:41): --rand_crop_sz 20 --input_sz 24 --include_rgb
  "cifar_rgb": ("sobel", NS(crop_orig=True, rand_crop_sz=20, input_sz=24, include_rgb=True, fluid_warp=False,
                            cutout=False, demean=False, per_img_demean=False), 32, 3),
  # CIFAR10 commands (commands.txt:25,38): --rand_crop_sz 20 --input_sz 32 (an up-scaling resize)
  "cifar_up": ("sobel", NS(crop_orig=True, rand_crop_sz=20, input_sz=32, include_rgb=False, fluid_warp=False,
                           cutout=False, demean=False, per_img_demean=False), 32, 3),
  # --cutout branch (transforms.py:170-186) with the flags of commands.txt:56 (--cutout_p 0.5 --cutout_max_box 0.7)
  "stl10_cutout": ("sobel", NS(crop_orig=True, rand_crop_sz=64, input_sz=64, include_rgb=True, fluid_warp=False,
                               cutout=True, cutout_p=0.5, cutout_max_box=0.7, demean=False,
                               per_img_demean=False), 96, 3),
  # --fluid_warp branch (transforms.py:142-152)
  "stl10_fluid": ("sobel", NS(crop_orig=True, rand_crop_sz=64, input_sz=64, include_rgb=False, fluid_warp=True,
                              rot_val=30.0, rand_crop_szs_tf=[48, 64, 80], cutout=False, demean=False,
                              per_img_demean=False), 96, 3),
  # --demean branch (transforms.py:196-204)
  "stl10_demean": ("sobel", NS(crop_orig=True, rand_crop_sz=64, input_sz=64, include_rgb=True, fluid_warp=False,
                               cutout=False, demean=True, data_mean=[0.43, 0.42, 0.39, 0.41],
                               data_std=[0.27, 0.26, 0.27, 0.25], per_img_demean=False), 96, 3),
  # MNIST commands (commands.txt:30,44)
  "mnist": ("grey", NS(crop_orig=True, crop_other=True, tf1_crop="centre_half", tf2_crop="random", tf1_crop_sz=20,
                       tf2_crop_szs=[16, 20, 24], tf3_crop_diff=False, tf3_crop_sz=0, input_sz=24, rot_val=25.0,
                       always_rot=False, no_flip=True, no_jitter=False, demean=False, per_img_demean=False), 28, 1),
  # the other greyscale switches: always_rot, centre crops, flips on, a tf3 crop of its own
  "grey_alt": ("grey", NS(crop_orig=True, crop_other=True, tf1_crop="random", tf2_crop="centre_half", tf1_crop_sz=22,
                          tf2_crop_szs=[18, 26], tf3_crop_diff=True, tf3_crop_sz=24, input_sz=24, rot_val=40.0,
                          always_rot=True, no_flip=False, no_jitter=False, demean=True, data_mean=[0.13],
                          data_std=[0.31], per_img_demean=False), 28, 1),
}
SAMPLES = 6


def seed_for(ci, si, which):
  return 1000003 * (ci + 1) + 1009 * si + which


def make_image(rs, size, channels):
  """Natural-image-like content: low-frequency colour blobs + texture + a few saturated pixels."""
  yy, xx = np.mgrid[0:size, 0:size].astype(np.float64) / size
  img = np.zeros((size, size, 3))
  for c in range(3):
    a, b, ph = rs.uniform(1, 4), rs.uniform(1, 4), rs.uniform(0, 6.28)
    img[..., c] = 0.5 + 0.35 * np.sin(a * 6.28 * xx + ph) * np.cos(b * 6.28 * yy + 0.7 * c)
  img += rs.normal(0, 0.08, img.shape)
  img = np.clip(img, 0, 1)
  k = rs.randint(0, size, (8, 2))
  img[k[:, 0], k[:, 1]] = rs.randint(0, 2, (8, 3))
  u8 = (img * 255).round().astype(np.uint8)
  return u8 if channels == 3 else u8[..., 1]


def main():
  import io
  from contextlib import redirect_stdout
  # custom_cutout (transforms.py:28-44) draws its box from numpy's global RNG inside the reference's own
  # code; the box it pastes is recorded by observing PIL's paste (the reference code stays untouched)
  orig_paste = Image.Image.paste

  def logging_paste(self, im, box=None, mask=None):
    tv021_shim.LOG.append(("paste", tuple(int(v) for v in box)))
    return orig_paste(self, im, box, mask)
  Image.Image.paste = logging_paste
  out = {}
  names = sorted(CONFIGS)
  for ci, name in enumerate(names):
    kind, cfg, size, ch = CONFIGS[name]
    with redirect_stdout(io.StringIO()):          # the builders print their choices
      tfs = ref_tf.sobel_make_transforms(cfg) if kind == "sobel" else ref_tf.greyscale_make_transforms(cfg)
    imgs, outs, logs = [], [[], [], []], [[], [], []]
    for si in range(SAMPLES):
      u8 = make_image(np.random.RandomState(77 + 131 * ci + si), size, ch)
      imgs.append(u8)
      for which in range(3):
        random.seed(seed_for(ci, si, which))
        np.random.seed(seed_for(ci, si, which))
        del tv021_shim.LOG[:]
        t = tfs[which](Image.fromarray(u8))
        outs[which].append(t.numpy().astype(np.float32))
        logs[which].append(repr(list(tv021_shim.LOG)))
    import json
    out[name + "/meta"] = np.array(json.dumps(dict(
      kind=kind, include_rgb=bool(getattr(cfg, "include_rgb", False)), input_sz=int(cfg.input_sz),
      norm=[list(cfg.data_mean), list(cfg.data_std)] if cfg.demean else None, config=vars(cfg))))
    out[name + "/images"] = np.stack(imgs)
    for which in range(3):
      out[name + "/tf%d" % (which + 1)] = np.stack(outs[which])
      out[name + "/log%d" % (which + 1)] = np.array(logs[which])
  # draw DISTRIBUTIONS of tf2 (the product draws its parameters vectorised, from its own generator: it cannot
  # replay a torchvision draw sequence, so its distributions are compared with the reference's, field by
  # field, in tests/test_augment_golden_cpu.py): NDRAW independent runs of the reference's tf2 per config
  NDRAW = 3000
  for ci, name in enumerate(names):
    kind, cfg, size, ch = CONFIGS[name]
    with redirect_stdout(io.StringIO()):
      tfs = ref_tf.sobel_make_transforms(cfg) if kind == "sobel" else ref_tf.greyscale_make_transforms(cfg)
    pil = Image.fromarray(make_image(np.random.RandomState(5), size, ch))
    random.seed(4242 + ci)
    np.random.seed(4242 + ci)
    rows = np.full((NDRAW, 12), np.nan)
    for k in range(NDRAW):
      del tv021_shim.LOG[:]
      tfs[1](pil.copy())
      for kd, val in tv021_shim.LOG:
        if kd == "crop":
          rows[k, 0:3] = val
        elif kd == "center_crop":
          rows[k, 2] = val
          rows[k, 0:2] = -1                      # marks "centre crop chosen"
        elif kd == "flip":
          rows[k, 3] = val
        elif kd == "jitter":
          for op, f in val:
            rows[k, 4 + op] = f
          rows[k, 8] = val[0][0]
        elif kd == "angle":
          rows[k, 9] = val
        elif kd == "paste":
          rows[k, 10] = val[2] - val[0]
          rows[k, 11] = (val[0] + val[2]) / 2.0
    out[name + "/draws_tf2"] = rows.astype(np.float32)
  out["names"] = np.array(names)
  path = os.path.join(ROOT, "tests", "golden", "augment.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, os.path.getsize(path), "bytes;", len(names), "configurations x", SAMPLES, "samples")
  print("example draws:", out["stl10_cutout/log2"][0])


if __name__ == "__main__":
  main()
