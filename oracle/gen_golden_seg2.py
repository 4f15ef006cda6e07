"""Second segmentation-loss fixture, again by running the REFERENCE ITSELF (read-only import):
  * BASELINE.json shapes at reduced batch / image size: Potsdam-3 head A
    (k = 24, T = 10, x-flips) and COCO-Stuff-3 head A (k = 15, T = 10, Bernoulli stuff mask);
  * general affine2_to_1 matrices (rotation + shear + scale, transforms.py:90-128 construction)
    through perform_affine_tf (transforms.py:131-143: affine_grid + grid_sample);
  * the sparse random translation (IID_losses.py:101-104, transforms.py:145-165), numpy's global
    RNG seeded right before each call.
Run in the build container only:  python -m oracle.gen_golden_seg2  -> tests/golden/iid_seg_loss2.npz
(gradients are stored from the float64 run, as float32, to keep the fixture small)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle.gen_golden import make_seg_inputs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (name, bn, k, h, w, T, lamb, flip_frac, mask_p, seed, affine, sparse_min, sparse_max, np_seed)
SEG2_CASES = [
  ("potsdam3_kA24_T10", 2, 24, 40, 40, 10, 1.0, 0.5, 1.0, 10, "flip", 0, 0, 0),
  ("coco3_kA15_T10_masked", 2, 15, 32, 32, 10, 1.0, 0.5, 0.6, 11, "flip", 0, 0, 0),
  ("potsdam3_kB3_T10", 3, 3, 40, 40, 10, 1.0, 0.34, 1.0, 12, "flip", 0, 0, 0),
  ("general_affine_T2", 3, 4, 16, 12, 2, 1.5, 0.0, 0.8, 13, "random", 0, 0, 0),
  ("general_affine_T1_wide", 2, 6, 20, 28, 1, 1.0, 0.0, 1.0, 14, "random", 0, 0, 0),
  ("sparse_shift_flip_T1", 3, 4, 16, 16, 1, 1.0, 0.67, 0.9, 15, "flip", 1, 3, 123),
  ("sparse_shift_affine_T2", 2, 5, 18, 14, 2, 1.0, 0.0, 1.0, 16, "random", 2, 2, 7),
]


def random_affines(bn, seed):
  """affine2_to_1 as transforms.py:90-128 builds it (inverse of rotation/shear/scale)."""
  rng = np.random.default_rng(seed)
  out = np.zeros((bn, 2, 3), dtype=np.float32)
  for i in range(bn):
    a = np.radians(rng.uniform(-30.0, 30.0))
    shear = np.radians(rng.uniform(-10.0, 10.0))
    scale = rng.uniform(0.8, 1.2)
    m = np.array([[np.cos(a) * scale, -np.sin(a + shear) * scale, 0.],
                  [np.sin(a) * scale, np.cos(a + shear) * scale, 0.],
                  [0., 0., 1.]], dtype=np.float32)
    out[i] = np.linalg.inv(m).astype(np.float32)[:2, :]
  return out


def case_inputs(case):
  name, bn, k, h, w, T, lamb, ff, mp, seed, affine, smin, smax, np_seed = case
  x1, x2, aff, mask = make_seg_inputs(bn, k, h, w, ff, mp, seed)
  if affine == "random":
    aff = random_affines(bn, seed + 100)
  return x1, x2, aff, mask


def main():
  ref = ref_import.ref_seg_losses()
  out = {}
  for case in SEG2_CASES:
    name, bn, k, h, w, T, lamb, ff, mp, seed, affine, smin, smax, np_seed = case
    x1, x2, aff, mask = case_inputs(case)
    for vname, fn in (("unc", ref.IID_segmentation_loss_uncollapsed), ("col", ref.IID_segmentation_loss)):
      for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        a = torch.from_numpy(x1).to(dt).requires_grad_(True)
        b = torch.from_numpy(x2).to(dt).requires_grad_(True)
        np.random.seed(np_seed)
        l, ln = fn(a, b, all_affine2_to_1=torch.from_numpy(aff).to(dt),
                   all_mask_img1=torch.from_numpy(mask).to(dt), lamb=lamb, half_T_side_dense=T,
                   half_T_side_sparse_min=smin, half_T_side_sparse_max=smax)
        l.backward()
        out["%s_%s_loss_%s" % (name, vname, tag)] = np.array([float(l), float(ln)])
        if tag == "f64":
          out["%s_%s_dx1" % (name, vname)] = a.grad.numpy().astype(np.float32)
          out["%s_%s_dx2" % (name, vname)] = b.grad.numpy().astype(np.float32)
  np.savez_compressed(os.path.join(OUT, "iid_seg_loss2.npz"), **out)
  print("iid_seg_loss2.npz written:", len(out), "arrays")


if __name__ == "__main__":
  main()
