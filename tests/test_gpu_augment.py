"""csrc/augment.hip through the C ABI against the PIL restatement of the reference's
sobel_make_transforms pipelines (oracle/augment_oracle.py) -- bit-exact float32 outputs for the
same random draws."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import augment_oracle as ao   # noqa: E402

pytestmark = pytest.mark.gpu


def _dataset(rng, B, H):
  imgs = rng.integers(0, 256, (B, H, H, 3), dtype=np.uint8)
  y, x = np.mgrid[0:H, 0:H]
  imgs[1] = np.clip(np.stack([127 + 120 * np.sin(x / 9.0 + y / 17.0), 127 + 120 * np.cos(x / 5.0),
                              (x * y) % 256], -1), 0, 255).astype(np.uint8)
  imgs[2] = np.repeat(imgs[2][:, :, :1], 3, 2)          # grey image: hue / saturation degenerate
  imgs[3] = 255
  imgs[4] = 0
  return imgs


def _oracle(imgs, ip, fp, crop, S, include_rgb, angles=None, norm=None):
  out = []
  for i in range(ip.shape[0]):
    ang = None if angles is None or np.isnan(angles[i]) else float(angles[i])
    c = crop[ip[i, 10]] if isinstance(crop, (list, tuple)) else crop
    order = [int(o) for o in ip[i, 5:5 + ip[i, 4]]]
    factors = {ao.OP_BRIGHTNESS: float(fp[i, 0]), ao.OP_CONTRAST: float(fp[i, 1]),
               ao.OP_SATURATION: float(fp[i, 2]), ao.OP_HUE: float(fp[i, 3])}
    assert ao.hue_delta(float(fp[i, 3])) == ip[i, 9]
    box = None
    if ip[i, 19] != 0:
      box = (ip[i, 18] & 0xffff, ip[i, 18] >> 16, ip[i, 19] & 0xffff, ip[i, 19] >> 16)
    a = ao.pil_pipeline(imgs[ip[i, 0]], (int(ip[i, 1]), int(ip[i, 2])), c, S, include_rgb,
                        bool(ip[i, 3]), order, factors, angle=ang, cutout_box=box, norm=norm)
    out.append(a)
  return np.stack(out)


@pytest.mark.parametrize("H,crop,S,include_rgb", [(96, 84, 96, True), (32, 20, 24, False),
                                                  (96, 64, 64, True), (40, 36, 24, True)])
def test_augment_bit_exact_vs_pil(H, crop, S, include_rgb):
  from iic_amd.augment import PairedAugmenter
  rng = np.random.default_rng(H + crop)
  imgs = _dataset(rng, 8, H)
  aug = PairedAugmenter(torch.from_numpy(imgs).cuda(), crop, S, include_rgb, seed=3)
  idx = rng.integers(0, 8, 48)
  for mode in ("jittered", "plain", "center"):
    ip, fp = aug.draw(idx, mode)
    got = aug.apply(ip, fp).cpu().numpy()
    want = _oracle(imgs, ip, fp, crop, S, include_rgb)
    assert got.shape == want.shape
    bad = [i for i in range(len(idx)) if not np.array_equal(got[i], want[i])]
    assert not bad, (mode, bad[:5], np.abs(got - want).max())
  if crop < H:
    ip, _ = aug.draw(idx, "plain")
    assert len(set(map(tuple, ip[:, 1:3]))) > 1           # crops are actually random


def test_augment_single_ops_and_extremes():
  from iic_amd.augment import PairedAugmenter, hue_shift
  rng = np.random.default_rng(0)
  imgs = _dataset(rng, 6, 48)
  aug = PairedAugmenter(torch.from_numpy(imgs).cuda(), 40, 48, True)
  rows_i, rows_f = [], []
  for src in range(6):
    for op, vals in ((0, (0.6, 1.0, 1.4)), (1, (0.6, 1.0, 1.4)), (2, (0.0, 0.6, 1.4)),
                     (3, (-0.5, -0.125, -0.004, 0.0, 0.004, 0.125, 0.5))):
      for v in vals:
        ip = np.zeros(20, np.int32)
        fp = np.zeros(4, np.float32)
        ip[[0, 1, 2, 3, 4, 5]] = (src, 3, 5, src & 1, 1, op)
        fp[op] = v
        if op == 3:
          ip[9] = hue_shift(float(fp[op]))
        rows_i.append(ip)
        rows_f.append(fp)
  ip, fp = np.stack(rows_i), np.stack(rows_f)
  got = aug.apply(ip, fp).cpu().numpy()
  want = _oracle(imgs, ip, fp, 40, 48, True)
  bad = [i for i in range(ip.shape[0]) if not np.array_equal(got[i], want[i])]
  assert not bad, (bad[:5], ip[bad[0]], fp[bad[0]])


def test_paired_batch_feeds_the_net_input_contract():
  """imgs / imgs_tf have the layout sobel_process expects (cluster_sobel.py:205-232):
  float32 [n, 4, S, S] in [0, 1], grey last; the two views differ, repeated tf2 draws differ."""
  from iic_amd.augment import PairedAugmenter
  rng = np.random.default_rng(1)
  imgs = _dataset(rng, 8, 96)
  aug = PairedAugmenter(torch.from_numpy(imgs).cuda(), 84, 96, True, seed=1)
  a, tfs = aug.paired_batch(np.arange(8), num_dataloaders=2)
  assert a.shape == (8, 4, 96, 96) and len(tfs) == 2 and tfs[0].shape == a.shape
  assert float(a.min()) >= 0 and float(a.max()) <= 1
  assert not torch.equal(tfs[0], tfs[1]) and not torch.equal(a, tfs[0])
  with pytest.raises(AssertionError):
    aug.plain([8])                                          # source index out of range


def test_augment_all_colours_exhaustive():
  """Every one of the 2^24 RGB values through hue / saturation / brightness / contrast (4096
  identity-resized 64x64 images): the float32 / float64 step order of PIL's blend and HSV round
  trip has no room for a fused multiply-add or a reordered operation."""
  from iic_amd.augment import PairedAugmenter, hue_shift
  v = np.arange(1 << 24, dtype=np.uint32)
  rgb = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8)
  rgb = rgb[np.random.default_rng(0).permutation(1 << 24)]        # mixed images: varied contrast means
  imgs = rgb.reshape(4096, 64, 64, 3)
  aug = PairedAugmenter(torch.from_numpy(imgs).cuda(), 64, 64, True)
  rng = np.random.default_rng(1)
  for op in (3, 2, 0, 1):
    ip = np.zeros((4096, 20), np.int32)
    fp = np.zeros((4096, 4), np.float32)
    ip[:, 0] = np.arange(4096)
    ip[:, 4] = 1
    ip[:, 5] = op
    fp[:, op] = rng.uniform(-0.5, 0.5, 4096) if op == 3 else rng.uniform(0.6, 1.4, 4096)
    if op == 3:
      ip[:, 9] = [hue_shift(float(f)) for f in fp[:, 3]]
    got = aug.apply(ip, fp).cpu().numpy()
    want = _oracle(imgs, ip, fp, 64, 64, True)
    bad = [i for i in range(4096) if not np.array_equal(got[i], want[i])]
    assert not bad, (op, len(bad), bad[:3])


def _mnist_cfg(**kw):
  import types
  cfg = dict(crop_orig=True, tf1_crop="centre_half", tf1_crop_sz=20, tf3_crop_diff=False, tf3_crop_sz=0,
             rot_val=25, always_rot=False, crop_other=True, tf2_crop="random", tf2_crop_szs=[16, 20, 24],
             input_sz=24, no_flip=True, no_jitter=False, demean=False, per_img_demean=False)
  cfg.update(kw)
  return types.SimpleNamespace(**cfg)


@pytest.mark.parametrize("variant", ["mnist685", "always_rot_flip", "no_crop_no_jitter"])
def test_greyscale_pipeline_bit_exact_vs_pil(variant):
  """greyscale_make_transforms (transforms.py:220-330) with the MNIST command's flags
  (examples/commands.txt:30) and two other flag sets: rotation (PIL NEAREST affine), per-sample crop
  size, resize, jitter on mode-L images, ToTensor."""
  from iic_amd.augment import GreyscaleAugmenter
  rng = np.random.default_rng(7)
  imgs = rng.integers(0, 256, (16, 28, 28), dtype=np.uint8)
  imgs[1] = (imgs[1] > 128) * 255                 # digit-like: saturated strokes on black
  imgs[2] = 0
  imgs[3] = 255
  cfg = {"mnist685": _mnist_cfg(),
         "always_rot_flip": _mnist_cfg(always_rot=True, no_flip=False, tf1_crop="random", tf2_crop="centre_half",
                                       tf3_crop_diff=True, tf3_crop_sz=24, rot_val=180),
         "no_crop_no_jitter": _mnist_cfg(crop_orig=False, crop_other=False, no_jitter=True, input_sz=32)}[variant]
  aug = GreyscaleAugmenter(torch.from_numpy(imgs).cuda(), cfg, seed=11)
  idx = rng.integers(0, 16, 96)
  for mode in ("jittered", "plain", "center"):
    ip, fp = aug.draw(idx, mode)
    got = aug.apply(ip, fp).cpu().numpy()
    want = _oracle(imgs, ip, fp, aug.crop_szs, cfg.input_sz, False,
                   aug.last_angles if mode == "jittered" else None)
    assert got.shape == want.shape == (96, 1, cfg.input_sz, cfg.input_sz)
    bad = [i for i in range(len(idx)) if not np.array_equal(got[i], want[i])]
    assert not bad, (variant, mode, bad[:5], ip[bad[0]])
    if mode == "jittered" and variant == "mnist685":
      assert 20 < int(ip[:, 11].sum()) < 76            # RandomApply(p=0.5)
      assert set(np.unique(ip[:, 10])) == {0, 1, 2}   # all three crop sizes drawn



@pytest.mark.parametrize("variant", ["cutout", "fluid_warp", "demean"])
def test_sobel_pipeline_variants_bit_exact_vs_pil(variant):
  """The non-default branches of sobel_make_transforms (transforms.py:142-204): --cutout (the one a
  published command uses, commands.txt), --fluid_warp (rotation + crop-size choice on RGB images)
  and --demean (Normalize), bit for bit against PIL with the same draws."""
  from iic_amd.augment import PairedAugmenter
  rng = np.random.default_rng(17)
  imgs = _dataset(rng, 8, 96)
  kw, norm, include_rgb = {}, None, True
  if variant == "cutout":
    kw = dict(cutout=True, cutout_p=0.7, cutout_max_box=0.5)
  elif variant == "fluid_warp":
    kw = dict(fluid_warp=True, rot_val=25.0, rand_crop_szs_tf=[64, 84, 72])
    include_rgb = False
  else:
    norm = ([0.43, 0.42, 0.39, 0.41], [0.27, 0.26, 0.28, 0.25])
    kw = dict(demean=True, data_mean=norm[0], data_std=norm[1])
  aug = PairedAugmenter(torch.from_numpy(imgs).cuda(), 84, 96, include_rgb, seed=4, **kw)
  idx = rng.integers(0, 8, 64)
  for mode in ("jittered", "plain", "center"):
    ip, fp = aug.draw(idx, mode)
    angles = aug.last_angles if (mode == "jittered" and variant == "fluid_warp") else None
    got = aug.apply(ip, fp).cpu().numpy()
    want = _oracle(imgs, ip, fp, aug.crop_szs, 96, include_rgb, angles=angles, norm=norm)
    bad = [i for i in range(len(idx)) if not np.array_equal(got[i], want[i])]
    assert not bad, (variant, mode, bad[:5], np.abs(got - want).max())
    if mode == "jittered" and variant == "cutout":
      assert (ip[:, 19] != 0).sum() > 20
    if mode == "jittered" and variant == "fluid_warp":
      assert (ip[:, 11] == 1).sum() > 10 and len(set(ip[:, 10])) == 3


def _iparams_from_log(aug, log, src_index, H, W):
  """The draws the reference's transforms made (oracle/tv021_shim.py log) as one iic_augment parameter row."""
  from iic_amd.augment import FPARAMS, IPARAMS, hue_shift, rotation_fixed_point
  kw = ao.params_from_log(log, (H, W))
  ip = np.zeros(IPARAMS, np.int32)
  fp = np.zeros(FPARAMS, np.float32)
  ip[0] = src_index
  ip[1], ip[2] = kw["crop_xy"]
  ip[3] = int(kw["flip"])
  ip[10] = aug.crop_szs.index(kw["crop_sz"])
  if kw["order"]:
    ip[4] = len(kw["order"])
    ip[5:5 + len(kw["order"])] = kw["order"]
    for op, f in kw["factors"].items():
      fp[op] = f
    if ao.OP_HUE in kw["factors"]:
      ip[9] = hue_shift(float(kw["factors"][ao.OP_HUE]))
  if kw["angle"] is not None and kw["angle"] % 360.0 != 0:
    ip[11] = 1
    ip[12:18] = rotation_fixed_point(float(kw["angle"]), W, H)
  if kw["cutout_box"] is not None:
    l, u, r, lo = kw["cutout_box"]
    if r > l and lo > u:
      ip[18] = l | (u << 16)
      ip[19] = r | (lo << 16)
  return ip, fp


def test_kernel_replays_reference_transform_fixtures():
  """csrc/augment.hip on the draws recorded while the reference's OWN sobel_make_transforms /
  greyscale_make_transforms (code/utils/cluster/transforms.py:107-334) ran in the build container
  (tests/golden/augment.npz, oracle/gen_golden_augment.py): bit-equal float32 tensors for every
  configuration (STL10 / CIFAR flag sets, --cutout, --fluid_warp, --demean, the MNIST flag set)."""
  import ast
  import json
  import types
  from iic_amd.augment import GreyscaleAugmenter, PairedAugmenter
  z = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment.npz"), allow_pickle=False)
  total = 0
  for name in [str(n) for n in z["names"]]:
    meta = json.loads(str(z[name + "/meta"]))
    cfg = types.SimpleNamespace(**meta["config"])
    imgs = z[name + "/images"]
    dev_imgs = torch.from_numpy(imgs).cuda()
    H, W = imgs.shape[1:3]
    if meta["kind"] == "sobel":
      aug = PairedAugmenter(dev_imgs, cfg.rand_crop_sz, cfg.input_sz, cfg.include_rgb, cutout=cfg.cutout,
                            cutout_p=getattr(cfg, "cutout_p", 0.5), cutout_max_box=getattr(cfg, "cutout_max_box", 0.5),
                            fluid_warp=cfg.fluid_warp, rot_val=getattr(cfg, "rot_val", 0.0),
                            rand_crop_szs_tf=getattr(cfg, "rand_crop_szs_tf", ()), demean=cfg.demean,
                            data_mean=getattr(cfg, "data_mean", ()), data_std=getattr(cfg, "data_std", ()))
    else:
      aug = GreyscaleAugmenter(dev_imgs, cfg)
    for which in (1, 2, 3):
      ref = z[name + "/tf%d" % which]
      rows = [_iparams_from_log(aug, ast.literal_eval(str(s)), i, H, W)
              for i, s in enumerate(z[name + "/log%d" % which])]
      got = aug.apply(np.stack([r[0] for r in rows]), np.stack([r[1] for r in rows])).cpu().numpy()
      assert got.shape == ref.shape, (name, which, got.shape, ref.shape)
      bad = [i for i in range(ref.shape[0]) if not np.array_equal(got[i], ref[i])]
      assert not bad, (name, which, bad, float(np.abs(got - ref).max()))
      total += ref.shape[0]
  assert total == 3 * 6 * len(z["names"])
