"""Paired-augmentation oracle (oracle/augment_oracle.py): the numpy algorithm specification of
csrc/augment.hip against the PIL restatement of the reference pipeline
(/root/reference/code/utils/cluster/transforms.py:107-217), bit for bit, plus the host-side
tables of iic_amd/augment.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import augment_oracle as ao   # noqa: E402


def _image(rng, H, W, kind):
  if kind == "noise":
    return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
  if kind == "smooth":
    y, x = np.mgrid[0:H, 0:W]
    a = np.stack([127 + 120 * np.sin(x / 9.0 + y / 17.0), 127 + 120 * np.cos(x / 5.0), (x * y) % 256], -1)
    return np.clip(a + rng.normal(0, 4, a.shape), 0, 255).astype(np.uint8)
  if kind == "grey":
    g = rng.integers(0, 256, (H, W, 1), dtype=np.uint8)
    return np.repeat(g, 3, 2)
  return np.full((H, W, 3), int(rng.integers(0, 256)), np.uint8)


@pytest.mark.parametrize("H,crop,S", [(96, 84, 96), (32, 20, 24), (96, 64, 64), (40, 36, 24)])
def test_numpy_spec_matches_pil(H, crop, S):
  rng = np.random.default_rng(H * 1000 + crop)
  bad = 0
  for t, kind in enumerate(["noise", "smooth", "grey", "flat", "noise", "smooth"]):
    img = _image(rng, H, H, kind)
    for p in ao.random_params(rng, 3, (H, H), crop):
      for include_rgb in (True, False):
        a = ao.pil_pipeline(img, p["crop_xy"], crop, S, include_rgb, p["flip"], p["order"], p["factors"])
        b = ao.np_pipeline(img, p["crop_xy"], crop, S, include_rgb, p["flip"], p["order"], p["factors"])
        assert a.shape == b.shape == ((4 if include_rgb else 1), S, S) and a.dtype == b.dtype == np.float32
        bad += int(not np.array_equal(a, b))
    # tf1 / tf3: no flip, no jitter
    a = ao.pil_pipeline(img, (1, 2), crop, S, True)
    b = ao.np_pipeline(img, (1, 2), crop, S, True)
    bad += int(not np.array_equal(a, b))
  assert bad == 0


def test_jitter_extremes_match_pil():
  """Factor range ends, every single op alone, hue wrap-around both ways."""
  rng = np.random.default_rng(5)
  img = _image(rng, 48, 48, "noise")
  cases = []
  for op, vals in ((ao.OP_BRIGHTNESS, (0.6, 1.0, 1.4)), (ao.OP_CONTRAST, (0.6, 1.0, 1.4)),
                   (ao.OP_SATURATION, (0.6, 1.0, 1.4, 0.0)), (ao.OP_HUE, (-0.125, -0.004, 0.0, 0.004, 0.125, 0.5, -0.5))):
    for v in vals:
      cases.append(([op], {op: v}))
  for order, factors in cases:
    a = ao.pil_pipeline(img, (3, 5), 40, 48, True, True, order, factors)
    b = ao.np_pipeline(img, (3, 5), 40, 48, True, True, order, factors)
    assert np.array_equal(a, b), (order, factors)


def test_host_tables_match_oracle_and_identity():
  from iic_amd.augment import bilinear_tables, hue_shift
  for i, o in ((84, 96), (20, 24), (64, 64), (36, 24), (96, 32), (7, 31)):
    b0, k0 = ao.resample_coeffs(i, o)
    b1, k1 = bilinear_tables(i, o)
    assert np.array_equal(b0, b1) and np.array_equal(k0, k1)
    assert (k1.sum(1) - (1 << 22)).__abs__().max() <= k1.shape[1]       # weights sum to ~1.0
  b, k = bilinear_tables(64, 64)                                         # same size: identity taps
  assert (k[:, 0] == 1 << 22).all() and (k[:, 1:] == 0).all() and (b[:, 0] == np.arange(64)).all()
  for f in (-0.5, -0.125, -0.001, 0.0, 0.001, 0.125, 0.5):
    assert hue_shift(f) == ao.hue_delta(f)


def test_paired_dataloaders_order_and_lengths():
  """The loader list mirrors _create_dataloaders (data.py:259-335): same sequential indices in
  every loader, tf1 first then tf2 loaders, short last batch kept."""
  import torch
  from iic_amd.augment import paired_dataloaders

  class Stub(object):
    B = 10
    def __init__(self):
      self.calls = []
    def plain(self, idx):
      self.calls.append(("plain", tuple(idx)))
      return torch.tensor(idx, dtype=torch.float32)
    def jittered(self, idx):
      self.calls.append(("jit", tuple(idx)))
      return torch.tensor(idx, dtype=torch.float32) + 0.5

  aug = Stub()
  targets = torch.arange(10) * 3
  loaders = paired_dataloaders(aug, targets, 4, 2)
  assert len(loaders) == 3 and all(len(l) == 3 for l in loaders)
  seen = 0
  for tup in zip(*[iter(l) for l in loaders]):
    a, t = tup[0]
    n = a.shape[0]
    assert n == (4 if seen < 8 else 2)
    for d in (1, 2):
      b, t2 = tup[d]
      assert torch.equal(b, a + 0.5) and torch.equal(t, t2)
    assert torch.equal(t, targets[seen:seen + n])
    seen += n
  assert seen == 10
  assert [c[0] for c in aug.calls[:3]] == ["plain", "jit", "jit"]


def test_greyscale_spec_matches_pil_with_rotation():
  """Mode-L pipeline of greyscale_make_transforms (transforms.py:220-330): PIL rotate (NEAREST
  fixed-point affine) -> crop (16 / 20 / 24) -> resize 24 -> flip -> jitter -> ToTensor."""
  rng = np.random.default_rng(3)
  bad = 0
  for t in range(90):
    img = rng.integers(0, 256, (28, 28), dtype=np.uint8)
    if t % 3 == 0:
      img = ((img > 128) * 255).astype(np.uint8)
    crop = [16, 20, 24][t % 3]
    p = ao.random_params(rng, 1, (28, 28), crop)[0]
    ang = float(rng.uniform(-25, 25)) if t % 2 else (None if t % 4 else float(rng.uniform(-180, 180)))
    a = ao.pil_pipeline(img, p["crop_xy"], crop, 24, False, p["flip"], p["order"], p["factors"], angle=ang)
    b = ao.np_pipeline(img, p["crop_xy"], crop, 24, False, p["flip"], p["order"], p["factors"], angle=ang)
    assert a.shape == b.shape == (1, 24, 24)
    bad += int(not np.array_equal(a, b))
  assert bad == 0


def test_greyscale_draws_follow_the_reference_flags():
  """Host-side parameter draws of GreyscaleAugmenter for the MNIST command (commands.txt:30):
  RandomApply(rotation, 0.5), RandomChoice over crop sizes, centre_half tf1, no flip."""
  import types
  import torch
  from iic_amd.augment import GreyscaleAugmenter, rotation_fixed_point
  cfg = types.SimpleNamespace(crop_orig=True, tf1_crop="centre_half", tf1_crop_sz=20, tf3_crop_diff=False,
                              tf3_crop_sz=0, rot_val=25, always_rot=False, crop_other=True, tf2_crop="random",
                              tf2_crop_szs=[16, 20, 24], input_sz=24, no_flip=True, no_jitter=False,
                              demean=False, per_img_demean=False)
  aug = GreyscaleAugmenter(torch.zeros(10, 28, 28, dtype=torch.uint8), cfg, seed=0)
  assert aug.crop_szs == [20, 16, 24]
  idx = np.arange(2000) % 10
  ip, fp = aug.draw(idx, "jittered")
  crop = np.asarray(aug.crop_szs)[ip[:, 10]]
  assert (ip[:, 1] >= 0).all() and (ip[:, 1] + crop <= 28).all() and (ip[:, 2] + crop <= 28).all()
  assert 0.4 < ip[:, 11].mean() < 0.6 and (ip[:, 3] == 0).all() and (ip[:, 4] == 4).all()
  assert all(abs((ip[:, 10] == t).mean() - 1 / 3) < 0.05 for t in range(3))
  assert (np.sort(ip[:, 5:9], 1) == np.arange(4)).all()
  assert fp[:, :3].min() >= 0.6 - 1e-6 and fp[:, :3].max() <= 1.4 + 1e-6
  rot = np.nonzero(ip[:, 11])[0]
  assert np.abs(aug.last_angles[rot]).max() <= 25 and np.isnan(aug.last_angles[ip[:, 11] == 0]).all()
  i = int(rot[0])
  assert tuple(ip[i, 12:18]) == ao.rotation_coeffs(float(aug.last_angles[i]), 28, 28) \
      == rotation_fixed_point(float(aug.last_angles[i]), 28, 28)
  ip1, _ = aug.draw(idx, "plain")                    # centre_half: half the crops at the centre (4, 4)
  centred = ((ip1[:, 1] == 4) & (ip1[:, 2] == 4)).mean()
  assert 0.45 < centred < 0.6 and (ip1[:, 10] == 0).all() and (ip1[:, 4] == 0).all()
  ip3, _ = aug.draw(idx, "center")
  assert (ip3[:, 1] == 4).all() and (ip3[:, 2] == 4).all()
  with __import__("pytest").raises(AssertionError):
    aug.apply(ip3, np.zeros((2000, 4), np.float32))  # CPU dataset: there is no CPU path


def test_sobel_draws_and_unsupported_flags():
  """Host-side draws of PairedAugmenter (distributions of RandomCrop / RandomHorizontalFlip /
  ColorJitter.get_params) and the flags the GPU path refuses."""
  import types
  import pytest
  import torch
  from iic_amd.augment import GreyscaleAugmenter, PairedAugmenter, hue_shift, paired_dataloaders
  aug = PairedAugmenter(torch.zeros(7, 96, 96, 3, dtype=torch.uint8), 84, 96, True, seed=5)
  idx = np.arange(3000) % 7
  ip, fp = aug.draw(idx, "jittered")
  assert ip.shape == (3000, 20) and fp.shape == (3000, 4) and (ip[:, 0] == idx).all()
  assert ip[:, 1].min() == 0 and ip[:, 1].max() == 12 and ip[:, 2].min() == 0 and ip[:, 2].max() == 12
  assert 0.45 < ip[:, 3].mean() < 0.55 and (ip[:, 10] == 0).all() and (ip[:, 11] == 0).all()
  assert (np.sort(ip[:, 5:9], 1) == np.arange(4)).all()
  assert len({tuple(r) for r in ip[:, 5:9]}) == 24                       # every op order occurs
  assert 0.6 <= fp[:, :3].min() and fp[:, :3].max() <= 1.4 + 1e-6 and np.abs(fp[:, 3]).max() <= 0.125
  assert all(ip[i, 9] == hue_shift(float(fp[i, 3])) for i in range(0, 3000, 97))
  ip1, fp1 = aug.draw(idx, "plain")
  assert (ip1[:, 3:10] == 0).all() and (fp1 == 0).all() and ip1[:, 1].max() == 12
  ip3, _ = aug.draw(idx, "center")
  assert (ip3[:, 1] == 6).all() and (ip3[:, 2] == 6).all()
  with pytest.raises(AssertionError):
    PairedAugmenter(torch.zeros(2, 28, 28, dtype=torch.uint8), 20, 24, False)   # RGB dataset required
  with pytest.raises(AssertionError):
    paired_dataloaders(aug, torch.zeros(6), 4, 2)                               # one target per image
  cfg = types.SimpleNamespace(crop_orig=True, tf1_crop="random", tf1_crop_sz=20, tf3_crop_diff=False,
                              tf3_crop_sz=0, rot_val=0, always_rot=False, crop_other=False, tf2_crop="random",
                              tf2_crop_szs=[20], input_sz=24, no_flip=False, no_jitter=True, demean=False,
                              per_img_demean=True)
  with pytest.raises(NotImplementedError):
    GreyscaleAugmenter(torch.zeros(2, 28, 28, dtype=torch.uint8), cfg)
  cfg.per_img_demean = False
  g = GreyscaleAugmenter(torch.zeros(2, 28, 28, dtype=torch.uint8), cfg, seed=1)
  ipg, fpg = g.draw(np.arange(400) % 2, "jittered")
  assert g.crop_szs == [20, 28]                      # tf1/tf3 crop 20; tf2 without crop_other: whole image
  assert (ipg[:, 10] == 1).all() and (ipg[:, 1:3] == 0).all() and (ipg[:, 4] == 0).all() and (ipg[:, 11] == 0).all()
  assert 0.4 < ipg[:, 3].mean() < 0.6 and (fpg == 0).all()


def test_cutout_and_normalize_spec_matches_pil():
  """custom_cutout (transforms.py:28-44) and the trailing Normalize: the numpy specification of the
  kernel against PIL, and the draws of PairedAugmenter(cutout=True)."""
  import torch
  from iic_amd.augment import PairedAugmenter
  rng = np.random.default_rng(3)
  img = rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)
  for box in ((0, 0, 8, 8), (5, 7, 21, 23), (20, 2, 32, 14), (30, 30, 32, 32)):
    norm = ([0.4, 0.5, 0.45, 0.47], [0.2, 0.25, 0.3, 0.22])
    a = ao.pil_pipeline(img, (4, 3), 32, 36, True, cutout_box=box, norm=norm)
    b = ao.np_pipeline(img, (4, 3), 32, 36, True, cutout_box=box, norm=norm)
    assert np.array_equal(a, b)
    c = ao.pil_pipeline(img, (4, 3), 32, 36, True)
    assert not np.array_equal(ao.pil_pipeline(img, (4, 3), 32, 36, True, cutout_box=box), c)
  aug = PairedAugmenter(torch.zeros(5, 96, 96, 3, dtype=torch.uint8), 84, 96, False, seed=2, cutout=True,
                        cutout_p=0.5, cutout_max_box=0.5)
  ip, _ = aug.draw(np.arange(4000) % 5, "jittered")
  has = ip[:, 19] != 0
  assert 0.45 < has.mean() < 0.55
  l, u = ip[has, 18] & 0xffff, ip[has, 18] >> 16
  r, lo = ip[has, 19] & 0xffff, ip[has, 19] >> 16
  side = r - l
  assert (side == lo - u).all() and side.min() == 2 * (int(84 * 0.2) // 2) and side.max() == 2 * (42 // 2)
  assert l.min() >= 0 and u.min() >= 0 and r.max() <= 84 and lo.max() <= 84
  # fluid_warp: rotation half of the time, crop size chosen from the list
  fw = PairedAugmenter(torch.zeros(5, 96, 96, 3, dtype=torch.uint8), 84, 96, False, seed=2, fluid_warp=True,
                       rot_val=25.0, rand_crop_szs_tf=[64, 84])
  ipf, _ = fw.draw(np.arange(2000) % 5, "jittered")
  assert fw.crop_szs == [84, 64] and set(np.unique(ipf[:, 10])) == {0, 1}
  assert 0.45 < (ipf[:, 11] == 1).mean() < 0.55 and np.nanmax(np.abs(fw.last_angles)) <= 25.0
  ipp, _ = fw.draw(np.arange(100) % 5, "plain")
  assert (ipp[:, 10] == 0).all() and (ipp[:, 11] == 0).all()
