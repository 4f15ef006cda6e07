"""Data-parallel plumbing: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI.  Replaces the reference's single-process torch.nn.DataParallel
(/root/reference/code/scripts/cluster/cluster_sobel.py:146) -- SURVEY.md §8e:

  * the batch is sharded by image PAIR (same rows of all_imgs and all_imgs_tf on a rank);
  * BatchNorm statistics stay per rank (exactly DataParallel's per-replica semantics);
  * the loss needs the GLOBAL joint: every rank all-reduces (sum) the raw k x k joints of
    all sub-heads in one small message, then evaluates the identical loss / dLoss/dR;
  * parameter gradients are all-reduced (SUM, not mean: the loss is already a function of
    the global joint) in a few large flat buckets.

Only collectives live here; `gloo` on CPU is used by the world_size-2 tests.
"""
import os

import torch
import torch.distributed as dist

_STATE = {"enabled": False, "group": None, "force": False}


def enable(group=None, force=None):
  """Switch the collectives on for `group` (default: the world).  A group of ONE rank normally keeps them off (the sums
  are identities); `force=True` (or IIC_DIST_FORCE=1) keeps them on so that a single MI355X executes the whole N > 1
  path -- the capture cut at the raw-joint all-reduce, the staged backward, the third stream, the bucket all-reduces --
  through real RCCL calls (tests/test_gpu_rccl.py; the only multi-rank evidence a one-GPU box can give)."""
  assert dist.is_initialized(), "init torch.distributed first"
  _STATE["enabled"] = True
  _STATE["group"] = group
  _STATE["force"] = (os.environ.get("IIC_DIST_FORCE", "0") == "1") if force is None else bool(force)


def disable():
  _STATE["enabled"] = False
  _STATE["group"] = None
  _STATE["force"] = False


def enabled():
  return _STATE["enabled"] and dist.is_initialized() and (_STATE["force"] or dist.get_world_size(_STATE["group"]) > 1)


def backend():
  """Backend name of the active group ("nccl" = RCCL on ROCm), or None."""
  return dist.get_backend(_STATE["group"]) if enabled() else None


CALLS = {}      # collective -> number of calls issued by this process (what DESIGN.md section 5 reports as executed)


def _count(name):
  CALLS[name] = CALLS.get(name, 0) + 1


def world_size():
  return dist.get_world_size(_STATE["group"]) if enabled() else 1


def rank():
  return dist.get_rank(_STATE["group"]) if enabled() else 0


# While a train step is being captured into HIP graphs (iic_amd.graph.CapturedPairStep), a
# collective cannot be recorded as a graph node portably: the capture is CUT there instead -- the
# graph captured so far is closed, the collective is remembered as an eager call on the same
# tensor, and a new graph is opened.  Replay = graph, collective, graph, ...  The hook is set by
# the capturing code only for the duration of the capture.
_CAPTURE_CUT = [None]


def _all_reduce_now(t, grp):
  """SUM all-reduce of `t`, ordered on the current stream.  Issued as an ASYNCHRONOUS collective + wait(): RCCL's
  kernel and the work's completion event then live on the process group's own stream, never on the caller's.  A
  synchronous collective runs on the caller's stream (torch >= 2.8), and the process group's watchdog thread keeps
  polling its completion event for a while after it finished -- if that stream starts a graph capture in the meantime
  (the warm-up step of a CapturedPairStep runs on the very streams it then captures on), HIP refuses the query
  (hipErrorCapturedEvent: 'operation not permitted on an event last recorded in a capturing stream') and the watchdog
  takes the process down.  Seen in about one of ten one-rank RCCL runs (round 6, tools/r06_rccl_flaky.sh)."""
  _count("all_reduce")
  if dist.get_backend(grp) == "nccl":
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=grp, async_op=True).wait()
  else:
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=grp)


def all_reduce_sum_(t):
  """In-place SUM all-reduce (no-op when not distributed)."""
  if enabled():
    grp = _STATE["group"]
    if _CAPTURE_CUT[0] is not None:
      _CAPTURE_CUT[0](lambda: _all_reduce_now(t, grp))
    else:
      _all_reduce_now(t, grp)
  return t


def settle_before_capture():
  """Call between the eager warm-up of a step and its graph capture: completed collectives stay in the process group's
  watchdog list until its next poll (every 100 ms); give it two polls to retire them before any stream that carried a
  collective's event starts capturing (see _all_reduce_now; one-time cost per captured step)."""
  if enabled() and dist.get_backend(_STATE["group"]) == "nccl":
    import time
    torch.cuda.synchronize()
    time.sleep(0.25)


def shard_rows(n_rows, r=None, w=None):
  """Contiguous slice [lo, hi) of the batch dimension owned by rank r (pairs stay together)."""
  r = rank() if r is None else r
  w = world_size() if w is None else w
  per = (n_rows + w - 1) // w
  lo = min(n_rows, r * per)
  return lo, min(n_rows, lo + per)


# Set by iic_amd.run under torchrun: the unchanged reference scripts build the FULL batch on every
# rank (their loaders know nothing about ranks), so every architecture's forward keeps only this
# rank's contiguous rows -- in TRAINING forwards only.  Evaluation (net.eval() / torch.no_grad():
# cluster_eval._clustering_get_data writes bs rows of predictions per batch) runs the whole
# batch on every rank.
SHARD_INPUTS = [False]


def shard_batch(x, module):
  """Rows [lo, hi) of `x` owned by this rank when input sharding is on and `module` is in a
  training forward; `x` itself otherwise."""
  if SHARD_INPUTS[0] and enabled() and module.training and torch.is_grad_enabled():
    lo, hi = shard_rows(x.size(0))
    return x[lo:hi]
  return x


def shard_like(t, n_local):
  """Per-sample side inputs of a loss (masks, affine matrices) arrive for the FULL batch from
  the unchanged script while the network outputs are already sharded: slice them to match."""
  if SHARD_INPUTS[0] and enabled() and t is not None and t.size(0) != n_local:
    lo, hi = shard_rows(t.size(0))
    assert hi - lo == n_local, "side input does not match this rank's shard"
    return t[lo:hi]
  return t


def broadcast_module_state(module, src=0):
  """Identical parameters AND buffers on every rank (the scripts seed nothing)."""
  if not enabled():
    return
  with torch.no_grad():
    for t in list(module.parameters()) + list(module.buffers()):
      _count("broadcast")
      dist.broadcast(t.data, src, group=_STATE["group"])


def all_reduce_grads(params, bucket_bytes=64 << 20):
  """SUM all-reduce of .grad over ranks in flat buckets (xGMI ring collectives are per-link
  bound: few large messages, not one per tensor).  Grads are copied into/out of a flat
  bucket; with 21.5 M fp32 params that is 2 x 86 MB of HBM traffic per step (~30 us)."""
  if not enabled():
    return
  bucket, size = [], 0
  def flush():
    if not bucket:
      return
    flat = torch.cat([p.grad.reshape(-1) for p in bucket])
    all_reduce_sum_(flat)
    off = 0
    for p in bucket:
      n = p.grad.numel()
      p.grad.copy_(flat[off:off + n].view_as(p.grad))
      off += n
  for p in params:
    if p.grad is None:
      continue
    bucket.append(p)
    size += p.grad.numel() * p.grad.element_size()
    if size >= bucket_bytes:
      flush()
      bucket, size = [], 0
  flush()


def all_reduce_grad_groups(groups):
  """SUM all-reduce of .grad, ONE flat bucket per parameter group, in list order -- the eager counterpart
  of the staged backward of iic_amd.graph.CapturedPairStep (same collectives, same sizes, same order, so
  ranks that replay graphs and ranks that launch eagerly stay compatible)."""
  if not enabled():
    return
  for grp in groups:
    members = [p for p in grp if p.grad is not None]
    if not members:
      continue
    flat = torch.cat([p.grad.reshape(-1) for p in members])
    all_reduce_sum_(flat)
    off = 0
    for p in members:
      n = p.grad.numel()
      p.grad.copy_(flat[off:off + n].view_as(p.grad))
      off += n


class GradReducer(object):
  """Bucketed SUM all-reduce of parameter gradients OVERLAPPED with the backward pass.

  Parameters are bucketed in reverse registration order (the order backward produces their
  gradients).  A post-accumulate-grad hook counts a bucket down; when its last gradient is
  final the bucket is flattened and all-reduced asynchronously (RCCL runs on its own stream,
  ordered after the work already enqueued on the compute stream), while the rest of backward
  keeps running.  ``finish()`` (between backward() and optimizer.step()) waits and scatters the
  sums back into ``.grad``.  Parameters that receive no gradient in a step (inactive head of a
  two-head net) are detected in finish(): incomplete buckets are reduced there.
  """

  def __init__(self, params, bucket_bytes=32 << 20):
    self.params = [p for p in params if p.requires_grad]
    self.buckets = []          # lists of params
    cur, size = [], 0
    for p in reversed(self.params):
      cur.append(p)
      size += p.numel() * p.element_size()
      if size >= bucket_bytes:
        self.buckets.append(cur)
        cur, size = [], 0
    if cur:
      self.buckets.append(cur)
    self.bucket_of = {}
    for bi, b in enumerate(self.buckets):
      for p in b:
        self.bucket_of[p] = bi
    self._reset()
    self.handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

  def _reset(self):
    self.pending = [len(b) for b in self.buckets]
    self.ready = [set() for _ in self.buckets]
    self.inflight = {}         # bucket index -> (flat, work, params)

  def _launch(self, bi, members):
    if not members:
      return
    flat = torch.cat([p.grad.reshape(-1) for p in members])
    _count("all_reduce_async")
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_STATE["group"], async_op=True)
    self.inflight[bi] = (flat, work, members)

  def _hook(self, p):
    if not enabled():
      return
    bi = self.bucket_of[p]
    self.ready[bi].add(p)
    self.pending[bi] -= 1
    if self.pending[bi] == 0:
      self._launch(bi, self.buckets[bi])

  def finish(self):
    """Call after backward(): reduces buckets that never completed (parameters without a
    gradient this step are skipped -- every rank skips the same ones), waits, scatters."""
    if enabled():
      for bi, b in enumerate(self.buckets):
        if bi not in self.inflight:
          self._launch(bi, [p for p in b if p.grad is not None])
      for bi in sorted(self.inflight):
        flat, work, members = self.inflight[bi]
        work.wait()
        off = 0
        for p in members:
          n = p.grad.numel()
          p.grad.copy_(flat[off:off + n].view_as(p.grad))
          off += n
    self._reset()

  def remove(self):
    for h in self.handles:
      h.remove()
