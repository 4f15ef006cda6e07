"""Times csrc/augment.hip on the north-star batch shape (STL10 96x96x3 sources, crop 84 -> 96,
include_rgb): 660 tf1 views + 660 tf2 views per training step.  Algorithmic bytes per output
image: crop read 84*84*3 + output write 4*96*96*4."""
import argparse
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from iic_amd.augment import PairedAugmenter   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=660)
ap.add_argument("--dataset", type=int, default=8192)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
imgs = torch.randint(0, 256, (a.dataset, 96, 96, 3), dtype=torch.uint8, generator=g).cuda()
aug = PairedAugmenter(imgs, 84, 96, True, seed=0)
idx = np.random.RandomState(0).randint(0, a.dataset, a.n)
for mode in ("plain", "jittered"):
  ip, fp = aug.draw(idx, mode)
  ipd = torch.from_numpy(ip).cuda(); fpd = torch.from_numpy(fp).cuda()
  aug.apply(ip, fp)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(a.iters):
    aug.apply(ip, fp)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / a.iters
  byts = a.n * (84 * 84 * 3 + 4 * 96 * 96 * 4)
  print("%-9s n=%d  %.3f ms/launch (incl. param upload)  %.1f GB/s algorithmic  %.0f images/s"
        % (mode, a.n, ms, byts / ms / 1e6, a.n / ms * 1e3))
import time
t0 = time.time()
for _ in range(10):
  aug.draw(idx, "jittered")
print("host draw (jittered, n=%d): %.3f ms" % (a.n, (time.time() - t0) * 100))
