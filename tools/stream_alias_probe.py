"""How HIP streams map onto hardware queues, seen through concurrency: for the k-th stream created by the process, do
two spin kernels on (stream 0, stream k) run side by side?  (iic_amd.graph._pair_streams keeps a pair that does.)
    gpurun -- python tools/stream_alias_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iic_amd.graph import _streams_overlap  # noqa: E402

s0 = torch.cuda.Stream()
row = []
for k in range(1, 13):
  sk = torch.cuda.Stream()
  row.append((k, _streams_overlap(s0, sk)))
print("stream k created after stream 0 -> runs beside it:", row)
from iic_amd.graph import _pair_streams
a, b = _pair_streams()
print("pair chosen:", a, b, "overlap", _streams_overlap(a, b))
