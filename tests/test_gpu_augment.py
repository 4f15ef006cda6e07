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


def _oracle(imgs, ip, fp, crop, S, include_rgb):
  out = []
  for i in range(ip.shape[0]):
    order = [int(o) for o in ip[i, 5:5 + ip[i, 4]]]
    factors = {ao.OP_BRIGHTNESS: float(fp[i, 0]), ao.OP_CONTRAST: float(fp[i, 1]),
               ao.OP_SATURATION: float(fp[i, 2]), ao.OP_HUE: float(fp[i, 3])}
    assert ao.hue_delta(float(fp[i, 3])) == ip[i, 9]
    a = ao.pil_pipeline(imgs[ip[i, 0]], (int(ip[i, 1]), int(ip[i, 2])), crop, S, include_rgb,
                        bool(ip[i, 3]), order, factors)
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
        ip = np.zeros(12, np.int32)
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
    ip = np.zeros((4096, 12), np.int32)
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
