"""Whole-train-step HIP graph: take the host out of the step.

One reference train step (cluster_sobel.py:235-272: sobel x2 -> net x2 -> IID_loss x sub-heads
-> backward -> Adam) is ~1100 kernel launches; issued one by one from Python/ctypes they cost
20-38 ms of host time against ~40 ms of GPU time.  Every kernel of
libiic_hip.so is enqueued asynchronously on the caller's stream with caller-owned buffers and
no hidden sync / allocation (include/iic_hip.h conventions), so a fixed-shape step can be
captured ONCE into a hipGraph and replayed with a single launch call.

    step = CapturedStep(fn)      # fn(): zero_grad -> forward -> loss -> backward -> opt.step()
    loss = step()                # replays; `loss` is the static tensor fn returned

Requirements on ``fn`` (the usual whole-network-capture rules): fixed shapes; inputs are
persistent device tensors that the caller overwrites in place between replays
(``batch.copy_(...)``); gradients are dropped with ``zero_grad(set_to_none=True)``; the optimiser
keeps its step count on the device (``iic_amd.optim.Adam(capturable=True)``); no host reads
(``.item()``) inside.  Nothing here is specific to one architecture.
"""
import torch

from .archs.cluster import bump_weights_epoch


class CapturedStep(object):
  def __init__(self, fn, warmup=2):
    assert torch.cuda.is_available(), "CapturedStep needs a device"
    self.fn = fn
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    # warm-up on a side stream (capture rule), also fills the PT buffer pool, the optimiser
    # state and the per-kernel attribute caches so that capture itself allocates next to nothing
    with torch.cuda.stream(side):
      for _ in range(max(1, warmup)):
        fn()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    self.graph = torch.cuda.CUDAGraph()
    # capture on the SAME stream the warm-up ran on: autograd's AccumulateGrad nodes remember the
    # stream they were created on; with a different capture stream the engine forks every
    # gradient accumulation onto the old stream (a branchy graph -- and on ROCm 7.0 consecutive
    # replays of such a graph were observed to overlap: tools/graph_debug.py)
    with torch.cuda.graph(self.graph, stream=side):
      self.out = fn()
    self.replays = 0

  def __call__(self):
    self.graph.replay()
    self.replays += 1
    # parameters were updated by raw-pointer kernels inside the graph: eager code that runs
    # after a replay (evaluation, an un-captured step) must re-derive its bf16 weight operands
    bump_weights_epoch()
    return self.out
