"""The augmentation oracle (oracle/augment_oracle.py: pil_pipeline, and np_pipeline -- the numpy
specification of csrc/augment.hip) against tests/golden/augment.npz, whose tensors were produced by the
reference's OWN transform builders (code/utils/cluster/transforms.py:107-334, executed through the
Python-2 hook on oracle/tv021_shim.py by oracle/gen_golden_augment.py) with every random draw recorded.
Replaying the recorded draws must give bit-equal float32 tensors: this pins the op composition,
parameters and draw order of the oracle to the reference's code (partial f1)."""
import ast
import json
import os

import numpy as np
import pytest

from oracle import augment_oracle as ao

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz")


def _cases():
  z = np.load(GOLDEN, allow_pickle=False)
  for name in [str(n) for n in z["names"]]:
    yield name, z


@pytest.mark.parametrize("name", [str(n) for n in np.load(GOLDEN, allow_pickle=False)["names"]])
def test_oracle_replays_reference_transforms(name):
  z = np.load(GOLDEN, allow_pickle=False)
  meta = json.loads(str(z[name + "/meta"]))
  imgs = z[name + "/images"]
  n_checked = 0
  for which in (1, 2, 3):
    ref = z[name + "/tf%d" % which]
    logs = z[name + "/log%d" % which]
    for i in range(imgs.shape[0]):
      kw = ao.params_from_log(ast.literal_eval(str(logs[i])), imgs[i].shape[:2])
      args = dict(out_sz=meta["input_sz"], include_rgb=meta["include_rgb"], norm=meta["norm"], **kw)
      got = ao.pil_pipeline(imgs[i], **args)
      assert got.dtype == np.float32 and got.shape == ref[i].shape
      assert np.array_equal(got, ref[i]), (name, which, i, float(np.abs(got - ref[i]).max()))
      spec = ao.np_pipeline(imgs[i], **args)
      assert np.array_equal(spec, ref[i]), (name, which, i, "numpy specification", float(np.abs(spec - ref[i]).max()))
      n_checked += 1
  assert n_checked == 3 * imgs.shape[0]


def test_fixture_covers_every_branch_of_the_reference_builders():
  """cutout applied and skipped, rotation applied and skipped, every crop-size choice, both crop kinds of
  centre_half, flips, all four jitter ops in several orders."""
  z = np.load(GOLDEN, allow_pickle=False)
  seen = {"paste": 0, "angle": 0, "flip1": 0, "flip0": 0, "center_crop": 0, "orders": set(), "crop_sizes": set(),
          "apply0": 0}
  for name in [str(n) for n in z["names"]]:
    for which in (1, 2, 3):
      for s in z[name + "/log%d" % which]:
        for kind, val in ast.literal_eval(str(s)):
          if kind == "paste": seen["paste"] += 1
          if kind == "angle": seen["angle"] += 1
          if kind == "flip": seen["flip%d" % val] += 1
          if kind == "center_crop": seen["center_crop"] += 1
          if kind == "apply" and val == 0: seen["apply0"] += 1
          if kind == "jitter": seen["orders"].add(tuple(op for op, _ in val))
          if kind == "crop": seen["crop_sizes"].add(val[2])
  assert seen["paste"] > 0 and seen["angle"] > 0 and seen["apply0"] > 0
  assert seen["flip1"] > 0 and seen["flip0"] > 0 and seen["center_crop"] > 0
  assert len(seen["orders"]) >= 6
  assert {16, 20, 24, 48, 64, 80} <= seen["crop_sizes"]


def _product_draws(name, meta, n):
  """The same 12 fields as `<name>/draws_tf2` of the fixture, from iic_amd.augment's host-side draws."""
  import types

  import torch
  from iic_amd.augment import GreyscaleAugmenter, PairedAugmenter
  cfg = types.SimpleNamespace(**meta["config"])
  z = np.load(GOLDEN, allow_pickle=False)
  imgs = torch.from_numpy(z[name + "/images"])           # host tensor: draw() never touches the device
  if meta["kind"] == "sobel":
    aug = PairedAugmenter(imgs, cfg.rand_crop_sz, cfg.input_sz, cfg.include_rgb, seed=11, cutout=cfg.cutout,
                          cutout_p=getattr(cfg, "cutout_p", 0.5), cutout_max_box=getattr(cfg, "cutout_max_box", 0.5),
                          fluid_warp=cfg.fluid_warp, rot_val=getattr(cfg, "rot_val", 0.0),
                          rand_crop_szs_tf=getattr(cfg, "rand_crop_szs_tf", ()), demean=cfg.demean,
                          data_mean=getattr(cfg, "data_mean", ()), data_std=getattr(cfg, "data_std", ()))
  else:
    aug = GreyscaleAugmenter(imgs, cfg, seed=11)
  ip, fp = aug.draw(np.zeros(n, np.int64), "jittered")
  rows = np.full((n, 12), np.nan)
  crop = np.asarray(aug.crop_szs)[ip[:, 10]]
  rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3] = ip[:, 1], ip[:, 2], crop, ip[:, 3]
  if meta["kind"] == "grey" and cfg.no_flip:
    rows[:, 3] = np.nan
  has_j = ip[:, 4] > 0
  rows[has_j, 4:8] = fp[has_j]
  rows[has_j, 8] = ip[has_j, 5]
  ang = getattr(aug, "last_angles", None)
  if ang is not None:
    rows[:, 9] = ang
  cut = getattr(aug, "last_cutout", None)
  if cut is not None and getattr(cfg, "cutout", False):
    on = cut[:, 2] > cut[:, 0]
    rows[on, 10] = (cut[on, 2] - cut[on, 0])
    rows[on, 11] = (cut[on, 0] + cut[on, 2]) / 2.0
  # centre crops of RandomChoice([RandomCrop, CenterCrop]) are marked -1 / -1 in the fixture
  if meta["kind"] == "grey" and cfg.crop_other and cfg.tf2_crop == "centre_half":
    H = imgs.shape[1]
    cx = np.array([int(round((H - c) / 2.)) for c in crop])
    centred = (ip[:, 1] == cx) & (ip[:, 2] == cx)
    # (a random crop can land on the centre too: only the RATE of the -1 marker is compared, below)
    rows[centred, 0] = rows[centred, 1] = -1
  return rows


@pytest.mark.parametrize("name", [str(n) for n in np.load(GOLDEN, allow_pickle=False)["names"]])
def test_product_draw_distributions_match_reference_transforms(name):
  """iic_amd.augment draws its parameters vectorised from its own generator, so it cannot replay a
  torchvision draw SEQUENCE; what must agree with the reference is every draw's DISTRIBUTION.  3000 runs
  of the reference's own tf2 (fixture) against 3000 draws of the product, field by field: two-sample
  Kolmogorov-Smirnov for the continuous fields, rate / frequency comparisons at 5 sigma for the discrete."""
  from scipy import stats
  z = np.load(GOLDEN, allow_pickle=False)
  meta = json.loads(str(z[name + "/meta"]))
  ref = z[name + "/draws_tf2"].astype(np.float64)
  got = _product_draws(name, meta, ref.shape[0])
  n = ref.shape[0]

  def rate_close(a, b, what):
    pa, pb = float(np.mean(a)), float(np.mean(b))
    sig = np.sqrt(max(pa * (1 - pa), 1e-4) * 2.0 / n)
    assert abs(pa - pb) <= 5 * sig, (name, what, pa, pb)

  fields = ["x0", "y0", "crop", "flip", "brightness", "contrast", "saturation", "hue", "first_op", "angle",
            "cut_size", "cut_centre"]
  for k, f in enumerate(fields):
    r, g = ref[:, k], got[:, k]
    rate_close(np.isnan(r), np.isnan(g), f + " presence")               # how often the draw happens at all
    r, g = r[~np.isnan(r)], g[~np.isnan(g)]
    if len(r) < 50:
      continue
    if f in ("x0", "y0"):
      rate_close(r < 0, g < 0, f + " centre-crop rate")
      r, g = r[r >= 0], g[g >= 0]
    if f in ("flip",):
      rate_close(r > 0.5, g > 0.5, f)
    elif f in ("crop", "first_op", "cut_size"):
      for v in np.unique(r):
        rate_close(r == v, g == v, "%s == %g" % (f, v))
      assert set(np.unique(g)) <= set(np.unique(r)), (name, f, np.unique(g), np.unique(r))
    else:
      assert g.min() >= r.min() - 1e-6 - 0.02 * (r.max() - r.min()) and g.max() <= r.max() + 1e-6 + 0.02 * (r.max() - r.min()), \
          (name, f, g.min(), g.max(), r.min(), r.max())
      p = stats.ks_2samp(r, g).pvalue
      assert p > 1e-4, (name, f, p)
