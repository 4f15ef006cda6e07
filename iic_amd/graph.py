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
import gc
import os
import sys

import torch

from .archs.cluster import bump_weights_epoch


def _no_gc(fn):
  """Run a capturing constructor with the cyclic garbage collector off: an object it collects while a stream is
  capturing (a dropped CUDAGraph, a pooled tensor) is freed by a HIP call that is illegal under capture and aborts
  the process (seen in round 4: `Fatal Python error: Aborted ... Garbage-collecting` inside a capture)."""
  def wrapped(*a, **k):
    was = gc.isenabled()
    gc.disable()
    try:
      return fn(*a, **k)
    finally:
      if was:
        gc.enable()
  wrapped.__doc__ = fn.__doc__
  return wrapped


_PAIR_STREAMS = {}     # device index -> (stream 1, stream 2)
PROBE_LOG = []         # one dict per concurrency probe (what was measured, what was kept); bench.py reports it
_PROBE = [os.environ.get("IIC_STREAM_PROBE", "1") != "0"]      # IIC_STREAM_PROBE=0: take streams as they come


def _log_probe(**kw):
  PROBE_LOG.append(kw)
  if os.environ.get("IIC_STREAM_PROBE_LOG", "0") == "1":
    sys.stderr.write("iic_amd stream probe: %s\n" % ", ".join("%s=%s" % (k, kw[k]) for k in sorted(kw)))


def pick_stream(beside=(), what="stream", collective_free=False, tries=8):
  """A new stream that really runs beside every stream in `beside` -- and, with `collective_free`, whose hardware queue
  is not the one the process group's collectives are issued on.  HIP multiplexes its streams onto a few hardware queues
  (4 by default) in creation order, and which queue a new stream lands on depends on how many streams the process
  has created before (a process group's initialisation creates some): candidates are created until one passes the
  probes (kept: the last one, if none does -- logged either way).
  Data parallel: the collective probe issues real collectives, so every rank must run the SAME sequence of them whatever
  its own measurements say -- each candidate is probed unconditionally and the verdict is agreed on (all-reduce MIN)
  before anyone moves on: all ranks try the same number of candidates."""
  from . import dist as idist
  st = torch.cuda.Stream()
  if not _PROBE[0] or torch.cuda.is_current_stream_capturing():
    return st
  agree = collective_free and idist.enabled()
  for i in range(tries):
    ok = all([_streams_overlap(b, st) for b in beside])
    if collective_free and _collective_blocks(st):
      ok = False
    if agree:
      flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
      torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=idist._STATE["group"])
      idist._count("all_reduce")
      ok = bool(flag.item() > 0.5)
    if ok:
      _log_probe(probe="pick", what=what, candidate=i, kept=True)
      return st
    if i + 1 < tries:
      st = torch.cuda.Stream()
  _log_probe(probe="pick", what=what, candidate=tries - 1, kept=False)
  return st


def _pair_streams():
  """The two streams of a paired step: ONE pair per device, shared by every CapturedPairStep of the process.  HIP
  multiplexes its streams onto a few hardware queues (4 by default), assigned as streams are created: a second
  CapturedPairStep with a fresh pair of its own (the two-head configs capture one step per head; bench.py captures
  the replica-de-duplication variant beside the headline step) can find both of its streams on ONE hardware queue --
  its two views then run one after the other (seen in the kernel trace of the MNIST two-head step: every dispatch of
  the head-B step on queue 4, profiles/r05_mnist6c_pair_timeline.txt).  The steps of a process run one at a time, so
  they can share the pair.  Data parallel (round 6): RCCL's collectives are issued on a stream torch's process group
  owns, which sits on one of the same few hardware queues -- a view whose stream shares that queue would be held up
  behind every bucket all-reduce (which itself waits for the fold of BOTH views): the pair is chosen among candidates
  that a pending collective does not block (_collective_blocks).  (Round 4 measured three alternatives -- the views on
  disjoint halves of the chip through CU-masked streams, half the CUs of every XCD each, view A at high priority:
  neutral, worse, neutral; LAB.md section 7.7 -- and round 5 removed the switches.)"""
  dev = torch.cuda.current_device()
  pair = _PAIR_STREAMS.get(dev)
  if pair is None:
    s1 = pick_stream((), "pair stream 1", collective_free=True)
    s2 = pick_stream((s1,), "pair stream 2", collective_free=True)
    pair = _PAIR_STREAMS[dev] = (s1, s2)
  return pair


def _streams_overlap(s1, s2, cycles=1200000):
  """Do two spin kernels (torch.cuda._sleep, ~0.5 ms each), one per stream, run side by side?  Two streams that share
  a hardware queue take twice as long as one."""
  if not _PROBE[0]:
    return True
  try:
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    with torch.cuda.stream(s1):
      torch.cuda._sleep(1000)          # (code object load / first-launch cost outside the timed part)
      ev[0].record()
      torch.cuda._sleep(cycles)
      ev[1].record()
    torch.cuda.synchronize()
    single = ev[0].elapsed_time(ev[1])
    s2.wait_stream(s1)
    with torch.cuda.stream(s1):
      ev[2].record()
      torch.cuda._sleep(cycles)
      ev[3].record()
    with torch.cuda.stream(s2):
      torch.cuda._sleep(cycles)
      ev[4].record()
    torch.cuda.synchronize()
    both = max(ev[2].elapsed_time(ev[3]), ev[2].elapsed_time(ev[4]))
    _log_probe(probe="overlap", single_ms=round(single, 3), paired_ms=round(both, 3), overlap=bool(both < 1.5 * single))
    return both < 1.5 * single
  except Exception:      # noqa: BLE001  (a probe never takes the step down)
    return True


def _collective_blocks(s, cycles=1200000):
  """Does a collective of the active process group that is waiting for its input hold up work on stream `s`?  torch's
  RCCL process group issues its collectives on a stream of its own, ordered after the issuing stream by an event; if
  that stream and `s` share a hardware queue, the queue's in-order packet processing parks `s` behind the wait.
  Probe: a scratch stream spins for 3 units and issues an asynchronous all-reduce (which therefore waits 3 units);
  `s` spins for 1 unit meanwhile.  1 unit = free, 4 units = blocked.  False without a process group (or on gloo,
  whose collectives are host-side)."""
  from . import dist as idist
  if not _PROBE[0] or not idist.enabled() or idist.backend() != "nccl":
    return False
  try:
    import torch.distributed as tdist
    grp = idist._STATE["group"]
    buf = torch.zeros(1024, device="cuda")
    idist._all_reduce_now(buf, grp)      # (communicator + its stream exist from here on)
    if "scratch" not in _COLL_PROBE:
      _COLL_PROBE["scratch"] = torch.cuda.Stream()
    sc = _COLL_PROBE["scratch"]
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(s):
      torch.cuda._sleep(1000)
      ev[0].record()
      torch.cuda._sleep(cycles)
      ev[1].record()
    torch.cuda.synchronize()
    single = ev[0].elapsed_time(ev[1])
    # (no early return between here and the collective below: every rank issues the same collectives whatever it measures)
    with torch.cuda.stream(s):
      ev[2].record()
    sc.wait_stream(s)
    with torch.cuda.stream(sc):
      torch.cuda._sleep(3 * cycles)
      idist._count("all_reduce_async")
      w = tdist.all_reduce(buf, op=tdist.ReduceOp.SUM, group=grp, async_op=True)
    with torch.cuda.stream(s):
      torch.cuda._sleep(cycles)
      ev[3].record()
    torch.cuda.synchronize()
    w.wait()
    if not _streams_overlap(s, sc):          # the scratch stream itself shares s's queue: nothing can be told apart
      _log_probe(probe="collective", inconclusive=True)
      return False
    held = ev[2].elapsed_time(ev[3])
    _log_probe(probe="collective", single_ms=round(single, 3), beside_pending_collective_ms=round(held, 3),
               blocked=bool(held > 2.5 * single))
    return held > 2.5 * single
  except Exception as e:      # noqa: BLE001  (a probe never takes the step down)
    _log_probe(probe="collective", error="%s: %s" % (type(e).__name__, e))
    return False


_COLL_PROBE = {}
_FOLD_STREAMS = {}     # device index -> the staged backward's third stream (one per device, like the pair)


class CapturedStep(object):
  @_no_gc
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
    from . import dist as idist
    idist.settle_before_capture()
    self.graph = torch.cuda.CUDAGraph()
    # capture on the SAME stream the warm-up ran on: autograd's AccumulateGrad nodes remember the
    # stream they were created on; with a different capture stream the engine forks every
    # gradient accumulation onto the old stream (a branchy graph -- and on ROCm 7.0 consecutive
    # replays of such a graph were observed to overlap: tools/graph_probes.py graph_debug)
    with torch.cuda.graph(self.graph, stream=side):
      self.out = fn()
    self.replays = 0

  def __call__(self):
    from .optim import sync_captured_lr
    sync_captured_lr()           # update_lr() between replays: the optimiser reads its rate from device memory
    self.graph.replay()
    self.replays += 1
    # parameters were updated by raw-pointer kernels inside the graph: eager code that runs
    # after a replay (evaluation, an un-captured step) must re-derive its bf16 weight operands
    bump_weights_epoch()
    return self.out


class _Segmented(object):
  """A capture that may be CUT by eager calls (collectives: iic_amd.dist.all_reduce_sum_): a list
  of linear graphs sharing one memory pool with the eager calls between them, replayed in order
  on one stream.  With no cut it is exactly one graph."""

  def __init__(self, pool, stream, error_mode):
    self.pool, self.stream, self.error_mode = pool, stream, error_mode
    self.items = []
    self._g = None

  def _begin(self):
    self._g = torch.cuda.CUDAGraph()
    self._g.capture_begin(pool=self.pool, capture_error_mode=self.error_mode)

  def _end(self):
    self._g.capture_end()
    self.items.append(self._g)
    self._g = None

  def cut(self, eager_fn):
    self._end()
    self.items.append(eager_fn)       # not executed now: its input is only produced at replay
    self._begin()

  def __enter__(self):
    from . import dist as idist
    torch.cuda.synchronize()
    self._ctx = torch.cuda.stream(self.stream)
    self._ctx.__enter__()
    self._prev = idist._CAPTURE_CUT[0]
    idist._CAPTURE_CUT[0] = self.cut
    self._begin()
    return self

  def __exit__(self, et, ev, tb):
    from . import dist as idist
    idist._CAPTURE_CUT[0] = self._prev
    try:
      self._end()
    finally:
      self._ctx.__exit__(et, ev, tb)
    return False

  def replay(self):
    for it in self.items:
      if isinstance(it, torch.cuda.CUDAGraph):
        it.replay()
      else:
        it()

  @property
  def cuts(self):
    return sum(1 for it in self.items if not isinstance(it, torch.cuda.CUDAGraph))


class CapturedPairStep(object):
  """The paired step as linear graphs on two streams:

      [ view A forward ]  ||  [ view B forward ]        (stream 1 || stream 2)
                 loss forward + backward                 (stream 1)
      [ view A backward ] ||  [ view B backward ]
                 optimiser                               (stream 1)

  The two views of a step (net(all_imgs), net(all_imgs_tf): cluster_sobel.py:238-239) are
  independent until the loss, so they can overlap on the GPU: the tail of one view's launch is
  filled by the other view's next launch, and its HBM-bound BatchNorm passes run beside the other
  view's MFMA-bound convolutions (iic_amd.ops.branch).  A single captured graph with two branches
  does that too, but ROCm 7.0 launches a branched graph node by node (measured 20 ms of host time
  per replay for this step); a LINEAR graph is launched as one pre-built packet list (0.3 ms).  So
  each view's forward and backward is captured as its own linear graph, and the host orders the
  six replays with stream waits -- same kernels, same arithmetic, bit-identical results.

  view_a(), view_b(): forward of one view -> output tensor (view_b runs as branch 1);
  loss_fn(xa, xb) -> scalar loss; finish(): optimiser step; zero_grad(): drop all gradients
  (set_to_none).  Inputs are persistent tensors, as for CapturedStep.

  Data parallel (one process per GPU): loss_fn and finish may call iic_amd.dist collectives (the
  raw-joint all-reduce inside the loss; fold + gradient all-reduce before the optimiser in
  finish).  Their captures are cut at those calls and the collectives are issued eagerly on
  stream 1 between the graph segments at replay: the host enqueues ~10 launches per step instead
  of ~1100, and every rank issues the same collectives in the same order as the eager step does
  (so a rank that fell back to eager launches stays compatible with ranks that replay).

  STAGED backward (`grad_groups`, data parallel): with one all-reduce in `finish` the gradient exchange
  starts when both backwards are over and nothing is left to hide it behind.  Given the parameters by layer
  group in backward order (ClusterNet5g.grad_groups()) and views that also return the activations at the
  group boundaries (forward_packed_taps), each view's backward is captured as one linear graph PER GROUP
  (torch.autograd.grad from the group's output to its input activation and its parameters).  At replay, as
  soon as group g of both views has run, a third stream folds the two views' gradients of that group into
  one flat bucket and issues its SUM all-reduce (async: RCCL's own stream) while streams 1 and 2 replay
  group g+1; the optimiser graph waits for the buckets.  .grad of every parameter is a view into its bucket.
  `finish` must then issue the SAME collectives for the eager warm-up steps (fold, one all-reduce per group
  in order, optimiser step: iic_amd.dist.all_reduce_grad_groups) and `opt_step` is the optimiser step alone.
  `events` (optional list) receives ("bwd", g) / ("reduce", g) / ("opt",) in host issue order at replay."""

  @_no_gc
  def __init__(self, view_a, view_b, loss_fn, finish, zero_grad, warmup=2, grad_groups=None, opt_step=None,
               events=None, force_staged=False):
    from . import dist as idist, ops
    assert torch.cuda.is_available(), "CapturedPairStep needs a device"
    # (force_staged: the staged capture without a process group -- tests / probes on one GPU)
    self.staged = grad_groups is not None and (idist.enabled() or force_staged)
    assert not self.staged or opt_step is not None, "staged backward: pass opt_step (the optimiser step alone)"
    self.events = events
    if grad_groups is not None:      # views return (output, boundary activations)
      va, vb = view_a, view_b
      self._taps = {}

      def view_a():
        out, self._taps["a"] = va()
        return out

      def view_b():
        out, self._taps["b"] = vb()
        return out
    self.fns = (view_a, view_b, loss_fn, finish, zero_grad)
    cur = torch.cuda.current_stream()
    self.s1, self.s2 = _pair_streams()
    s1, s2 = self.s1, self.s2
    s1.wait_stream(cur)
    for _ in range(max(1, warmup)):
      self._eager_step()
    torch.cuda.synchronize()
    idist.settle_before_capture()
    zero_grad()
    ops.clear_branch_grads()
    pool_a, pool_b = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
    G = torch.cuda.CUDAGraph
    self.g_fa, self.g_fb, self.g_ba, self.g_bb = G(), G(), G(), G()
    # (a process group's watchdog thread polls events while we capture: thread-local error mode)
    mode = "thread_local" if torch.distributed.is_available() and torch.distributed.is_initialized() else "global"
    with torch.cuda.graph(self.g_fa, pool=pool_a, stream=s1, capture_error_mode=mode):
      xa = view_a()
    with torch.cuda.graph(self.g_fb, pool=pool_b, stream=s2, capture_error_mode=mode):
      with ops.on_branch(1, s2):
        xb = view_b()
    # the loss and the optimiser phase may contain collectives (data parallel: the raw-joint
    # all-reduce inside the loss, the gradient all-reduce before the optimiser): captured in
    # segments cut at those calls (iic_amd.dist._CAPTURE_CUT), the collectives replayed eagerly
    self.g_l = _Segmented(pool_a, s1, mode)
    with self.g_l:
      ops.flush_deferred_running()
      xa_d, xb_d = xa.detach().requires_grad_(True), xb.detach().requires_grad_(True)
      loss = loss_fn(xa_d, xb_d)
      loss.backward()
      ga, gb = xa_d.grad, xb_d.grad
      self.out = loss.detach()
    if self.staged:
      self._capture_staged(xa, xb, ga, gb, grad_groups, opt_step, pool_a, pool_b, mode)
    else:
      with torch.cuda.graph(self.g_ba, pool=pool_a, stream=s1, capture_error_mode=mode):
        xa.backward(ga)
      with torch.cuda.graph(self.g_bb, pool=pool_b, stream=s2, capture_error_mode=mode):
        xb.backward(gb)
      self.g_opt = _Segmented(pool_a, s1, mode)
      with self.g_opt:
        finish()
    self._keep = (xa, xb, xa_d, xb_d, ga, gb)     # buffers that cross graph boundaries
    cur.wait_stream(s1)
    self.replays = 0

  def _capture_staged(self, xa, xb, ga, gb, groups, opt_step, pool_a, pool_b, mode):
    from . import ops
    s1, s2 = self.s1, self.s2
    # a third stream that runs beside both views' (see _pair_streams); sharing a hardware queue with the collectives'
    # own stream is fine for this one: fold g -> all-reduce g is one chain
    s3 = _FOLD_STREAMS.get(torch.cuda.current_device())
    if s3 is None:
      s3 = _FOLD_STREAMS[torch.cuda.current_device()] = pick_stream((s1, s2), "fold / all-reduce stream")
    self.s3 = s3
    pool_c = torch.cuda.graph_pool_handle()
    n = len(groups)
    taps_a, taps_b = self._taps["a"], self._taps["b"]
    assert len(taps_a) == n - 1 and len(taps_b) == n - 1, "one boundary activation between consecutive groups"
    self.g_ba, self.g_bb, self.g_fold, self.buckets = [], [], [], []
    cur = [xa, xb]
    gcur = [ga, gb]
    keep = []
    for g, grp in enumerate(groups):
      grp = [p for p in grp if p.requires_grad]
      res = []
      for v, (stream, pool, taps, graphs) in enumerate(((s1, pool_a, taps_a, self.g_ba), (s2, pool_b, taps_b, self.g_bb))):
        leaves = grp if v == 0 else [ops.branch_leaf(p, 1) for p in grp]
        ins = ([taps[g]] if g < n - 1 else []) + leaves
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, pool=pool, stream=stream, capture_error_mode=mode):
          r = torch.autograd.grad([cur[v]], ins, [gcur[v]], allow_unused=True)
        graphs.append(gr)
        if g < n - 1:
          assert r[0] is not None, "group %d: no gradient reaches its input activation" % g
          cur[v], gcur[v] = taps[g], r[0]
          r = r[1:]
        res.append(r)
      keep.append(res)
      # fold: bucket = view A's gradients + view B's, flat, on the third stream
      members = [(p, a, b) for p, a, b in zip(grp, res[0], res[1]) if a is not None or b is not None]
      flat = torch.empty(sum(p.numel() for p, _, _ in members), dtype=torch.float32, device=xa.device)
      views, off = [], 0
      for p, _, _ in members:
        views.append(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
      s3.wait_stream(s1)
      s3.wait_stream(s2)
      gf = torch.cuda.CUDAGraph()
      with torch.cuda.graph(gf, pool=pool_c, stream=s3, capture_error_mode=mode):
        first = [a if a is not None else b for _, a, b in members]
        torch._foreach_copy_(views, first)
        both = [(v, b) for v, (_, a, b) in zip(views, members) if a is not None and b is not None]
        if both:
          torch._foreach_add_([v for v, _ in both], [b for _, b in both])
      self.g_fold.append(gf)
      self.buckets.append(flat)
      for (p, _, _), v in zip(members, views):
        p.grad = v                       # the optimiser graph reads the all-reduced bucket
      absent = set(id(p) for p in grp) - set(id(p) for p, _, _ in members)
      for p in grp:
        if id(p) in absent:
          p.grad = None
    ops.clear_branch_grads()
    s1.wait_stream(s3)
    self.g_opt = _Segmented(pool_a, s1, mode)
    with self.g_opt:
      opt_step()
    self._keep_staged = keep

  def _eager_step(self):
    from . import ops
    view_a, view_b, loss_fn, finish, zero_grad = self.fns
    s1, s2 = self.s1, self.s2
    zero_grad()
    ops.clear_branch_grads()
    s2.wait_stream(s1)
    with ops.on_branch(1, s2):
      xb = view_b()
    with torch.cuda.stream(s1):
      xa = view_a()
      s1.wait_stream(s2)
      ops.flush_deferred_running()
      xa_d, xb_d = xa.detach().requires_grad_(True), xb.detach().requires_grad_(True)
      loss = loss_fn(xa_d, xb_d)
      loss.backward()
    s2.wait_stream(s1)
    with torch.cuda.stream(s2):
      xb.backward(xb_d.grad)
    with torch.cuda.stream(s1):
      xa.backward(xa_d.grad)
      s1.wait_stream(s2)
      finish()
    return loss.detach()

  def _replay_staged(self):
    import torch.distributed as tdist
    from . import dist as idist
    s1, s2, s3 = self.s1, self.s2, self.s3
    ev = self.events
    works = []
    for g in range(len(self.g_fold)):
      with torch.cuda.stream(s2):
        self.g_bb[g].replay()
      with torch.cuda.stream(s1):
        self.g_ba[g].replay()
      if ev is not None:
        ev.append(("bwd", g))
      s3.wait_stream(s1)
      s3.wait_stream(s2)
      with torch.cuda.stream(s3):
        self.g_fold[g].replay()
        # asynchronous: RCCL works on its own stream, ordered after the fold; streams 1 and 2 go on
        # with the next group (gloo: a worker thread, the host does not wait here either)
        if idist.enabled():
          idist._count("all_reduce_async")
          works.append(tdist.all_reduce(self.buckets[g], op=tdist.ReduceOp.SUM, group=idist._STATE["group"],
                                        async_op=True))
      if ev is not None:
        ev.append(("reduce", g))
    with torch.cuda.stream(s1):
      s1.wait_stream(s3)
      for w in works:
        w.wait()                # (stream 1 waits for the collective's stream; gloo: the host does)
      if ev is not None:
        ev.append(("opt",))
      self.g_opt.replay()

  def __call__(self):
    from .optim import sync_captured_lr
    s1, s2 = self.s1, self.s2
    cur = torch.cuda.current_stream()
    sync_captured_lr()           # (on `cur`, which stream 1 waits for next)
    s1.wait_stream(cur)
    s2.wait_stream(s1)
    with torch.cuda.stream(s2):
      self.g_fb.replay()
    with torch.cuda.stream(s1):
      self.g_fa.replay()
      s1.wait_stream(s2)
      self.g_l.replay()
    s2.wait_stream(s1)
    if self.staged:
      self._replay_staged()
    else:
      with torch.cuda.stream(s2):
        self.g_bb.replay()
      with torch.cuda.stream(s1):
        self.g_ba.replay()
        s1.wait_stream(s2)
        self.g_opt.replay()
    cur.wait_stream(s1)
    self.replays += 1
    bump_weights_epoch()
    return self.out
