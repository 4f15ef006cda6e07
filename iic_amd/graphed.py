"""HIP-graph replay for the DROP-IN path: the training forwards (and their backwards) of an architecture
driven by an UNCHANGED reference script.

iic_amd.graph.CapturedStep / CapturedPairStep capture a whole step, which needs the step as ONE callable --
bench.py has that, the reference's scripts do not: between `net(all_imgs)`, `net(all_imgs_tf)`, the
per-sub-head `IID_loss` calls, `.item()`, `backward()` and `optimiser.step()`
(/root/reference/code/scripts/cluster/cluster_sobel.py:235-272) runs the script's own Python.  But ~95 % of
a step's ~1100 launches sit inside the two forwards and their two backwards, which the script enters through
OUR forward and autograd.  So each training forward of a given (input shape, head, position in the step) is
captured once -- after WARMUP eager occurrences -- as two HIP graphs:

    forward graph :  static input  -> the architecture's forward (weight re-layout, convs, BN, heads)
    backward graph:  static output gradients -> torch.autograd.grad w.r.t. every parameter

and a later call is `copy_ into the static input -> replay -> static outputs` inside an autograd.Function
whose backward is `copy_ the incoming gradients -> replay -> return the static parameter gradients` (autograd
accumulates them into .grad as in the eager run).  The losses, the optimiser and everything the script does
in between stay eager -- so `update_lr` (code/utils/cluster/general.py:20-23, an in-place edit of
optimiser.param_groups) keeps working with no device-side learning rate, and a shape the graph was not
captured for (the last, smaller batch of an epoch) simply runs eager.

Works under iic_amd.ops.auto_branch (the first forward of a step on a side stream): the BatchNorm
running-statistic updates a branch forward postpones to the join (ops._DEFERRED_RUNNING) are recorded at
capture time and re-queued at every replay; both positions re-lay their own bf16 weight operands inside
their forward graph (one operand set per branch, ops / archs.cluster._ConvHolder.weights).

Switch: ops.GRAPH_FORWARD[0] (env IIC_GRAPH_FORWARD; `python -m iic_amd.run` turns it on).

Gradient hand-over: a replayed view installs / adds its static parameter gradients into `.grad` itself and returns
nothing for the parameters to autograd (one multi-tensor add instead of 118 AccumulateGrad launches per view) -- which is
what `loss.backward()` followed by `optimiser.step()` wants, the only pattern the reference's scripts use.  Parameters
with hooks keep the autograd route.  `torch.autograd.grad(loss, parameters)` therefore sees no gradient for them while
`.grad` is written as a side effect: callers that need that API switch the fused hand-over off
(IIC_GRAPH_FUSED_ACC=0 / FUSED_ACCUMULATE[0] = False) or graph replay itself.
"""
import gc
import os
import sys
import weakref

import torch

from . import ops
from .archs import cluster as _cl

WARMUP = 2          # eager occurrences of a key before it is captured
FUSED_ACCUMULATE = [os.environ.get("IIC_GRAPH_FUSED_ACC", "1") != "0"]
_FAILED = object()


def _flat(out):
  if torch.is_tensor(out):
    return [out], None
  assert isinstance(out, (list, tuple)) and all(torch.is_tensor(t) for t in out), \
      "graphed forward: the architecture must return a tensor or a list of tensors"
  return list(out), type(out)


class _ViewGraph(object):
  def __init__(self, fwd, mod, x, args, kwargs, res_branch):
    self.params = [p for p in mod.parameters() if p.requires_grad]
    self.static_in = x.detach().clone()
    cur = torch.cuda.current_stream()
    self.cap = torch.cuda.Stream()
    self.cap.wait_stream(cur)
    pool = torch.cuda.graph_pool_handle()
    mode = "thread_local" if torch.distributed.is_available() and torch.distributed.is_initialized() else "global"
    n_def = len(ops._DEFERRED_RUNNING)
    _cl.bump_weights_epoch()                 # the bf16 operand re-layout belongs INSIDE the forward graph
    # The captured forward sees the parameters through leaf ALIASES created on the capture stream (same
    # storage): the parameters' own AccumulateGrad nodes live on the stream the script's warm-up steps
    # ran on -- usually the legacy default stream --, and the captured backward touching them makes the
    # autograd engine synchronise the capture stream with that stream, which ends the capture with a
    # crash in hipStreamEndCapture (tools/graph_probes.py graphed_probe: fine when the warm-up ran on a side stream).
    with torch.cuda.stream(self.cap):
      self.leaves = [p.detach().requires_grad_(True) for p in self.params]
    self.g_f = torch.cuda.CUDAGraph()
    ops._CAPTURE_PROXIES[0] = {id(p): q for p, q in zip(self.params, self.leaves)}
    # Resource namespace (PT buffer pool, statistic accumulators, scratch, weight operands are keyed by
    # ops.BRANCH): the forward and the backward of a view are captured back to back, so the pool believes the
    # view's saved activations are free again when the NEXT position is captured -- although at replay the
    # other view's forward runs between this view's forward and backward.  Positions that share a real
    # branch (one-stream runs) therefore get a namespace of their own.
    prev_branch = ops.BRANCH[0]
    ops.BRANCH[0] = res_branch
    # no cyclic garbage collection while a stream is capturing: a collected tensor / graph would be freed by a HIP call
    # that is illegal under capture (torch.cuda.graph collects once on entry, but a capture allocates plenty itself)
    gc_was = gc.isenabled()
    gc.disable()
    try:
      with torch.cuda.graph(self.g_f, pool=pool, stream=self.cap, capture_error_mode=mode):
        with torch.enable_grad():
          out = fwd(mod, self.static_in, *args, **kwargs)
    finally:
      ops._CAPTURE_PROXIES[0] = None
      ops.BRANCH[0] = prev_branch
      if gc_was:
        gc.enable()
    self.outs, self.out_type = _flat(out)
    # running-statistic updates this forward postponed to the join (branch mode): the same (static) tensors
    # at every replay
    self.deferred = list(ops._DEFERRED_RUNNING[n_def:])
    self.gouts = [torch.zeros_like(o) for o in self.outs]
    self.g_b = torch.cuda.CUDAGraph()
    gc.disable()
    try:
      with torch.cuda.graph(self.g_b, pool=pool, stream=self.cap, capture_error_mode=mode):
        grads = torch.autograd.grad(self.outs, self.leaves, self.gouts, allow_unused=True)
    finally:
      if gc_was:
        gc.enable()
    self.grads = list(grads)
    cur.wait_stream(self.cap)
    self.static_outs = [o.detach() for o in self.outs]
    self.first = True
    self.sig = _storage_sig(mod)
    self.bwd_event = None        # recorded after every backward replay (consumers of the static gradients wait on it)
    # (a weak reference: vg -> module state -> vg would be a cycle, and a captured graph that only the cyclic garbage
    #  collector frees can be destroyed in the middle of ANOTHER capture -- hipGraphDestroy under capture aborts)
    self.mod_ref = weakref.ref(mod)


def _storage_sig(mod):
  """Device addresses of every parameter and buffer of the module.  A captured graph has them baked in; the
  reference's scripts move the whole network to the host and back around every checkpoint
  (/root/reference/code/scripts/cluster/cluster_sobel.py:314-339: net.module.cpu() ... net.module.cuda()), which
  re-allocates all of them -- replaying the old graphs would then train on freed memory (found by running the real
  scripts on the GPU, round 4: epoch 2 of a graph-replay run diverged from the eager run)."""
  return tuple(p.data_ptr() for p in mod.parameters()) + tuple(b.data_ptr() for b in mod.buffers())


_SIG_FULL_EVERY = 256


def _apply_generation(mod):
  """How often nn.Module._apply ran on this module (or on a parent that holds it: the count lives in a shared cell)."""
  return _cl.APPLY_GENERATION[0]


def _sig_changed(mod, st, vg):
  """vg.sig != _storage_sig(mod), without walking the module tree at every forward (0.65 ms of host time per call for
  ClusterNet5g -- time by which the second view's forward graph starts after the first's): the Parameter / buffer
  OBJECTS are listed once per module state and only their addresses are compared; every _SIG_FULL_EVERY-th call the
  tree is walked again, which also catches a Parameter object that was replaced rather than moved."""
  ts = st.get("sig_tensors")
  n = st["sig_calls"] = st.get("sig_calls", 0) + 1
  # nn.Module._apply (.cpu() / .cuda() / .to()) keeps the Parameter objects and REPLACES the buffer objects: the cached
  # list would pin the old device buffers and keep comparing their unchanged addresses.  The architectures count their
  # _apply calls (archs.cluster._ApplyCounter): a new count means a fresh walk.
  # The count only sees _apply on the architecture object itself; net.trunk.cuda() / bn.to(...) on a SUBMODULE replaces
  # its buffer objects without passing there: the buffer dictionaries are therefore compared by identity
  # as well -- 3 entries per BatchNorm, ~20 us per forward for ClusterNet5g.
  gen = _apply_generation(mod)
  stale = ts is None or n % _SIG_FULL_EVERY == 0 or st.get("sig_gen") != gen
  if not stale:
    for d, name, t in st["sig_bufs"]:
      if d.get(name) is not t:
        stale = True
        break
  if stale:
    st["sig_gen"] = gen
    st["sig_bufs"] = [(m._buffers, name, t) for m in mod.modules() for name, t in m._buffers.items() if t is not None]
    ts = st["sig_tensors"] = list(mod.parameters()) + list(mod.buffers())
  if len(ts) != len(vg.sig):
    return True
  for t, a in zip(ts, vg.sig):
    if t.data_ptr() != a:
      return True
  return False


class _GraphedFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, vg, x, *params):
    vg.static_in.copy_(x)
    vg.g_f.replay()
    if not vg.first:                         # (at capture the forward itself queued them)
      ops._DEFERRED_RUNNING.extend(vg.deferred)
    vg.first = False
    ctx.vg = vg
    # the stream the script itself is on (inside ops.branch: the stream that forked): its optimiser will read the gradients
    ctx.caller_stream = ops._BRANCH_MAIN[0] or torch.cuda.current_stream()
    return tuple(o.detach() for o in vg.static_outs)       # fresh tensor objects over the static storage

  @staticmethod
  def backward(ctx, *gouts):
    vg = ctx.vg
    # a .grad that still aliases our static buffer (gradient accumulation over several backward passes,
    # or zero_grad(set_to_none=False)) must not be overwritten by the replay
    for p, g in zip(vg.params, vg.grads):
      if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
        p.grad = p.grad.clone()
    for s, g in zip(vg.gouts, gouts):
      if g is None:
        s.zero_()
      else:
        s.copy_(g)
    vg.g_b.replay()
    vg.bwd_event = torch.cuda.Event()
    vg.bwd_event.record()
    # The first view to arrive hands autograd fresh aliases of its static gradient buffers: AccumulateGrad takes
    # them as .grad without a copy (it owns the only reference to the alias).  A later view (every .grad already
    # set, by AccumulateGrad nodes that ran on this same engine thread) would cost one `add_` launch per
    # parameter there -- 118 for ClusterNet5g, 0.6 ms per step: one multi-tensor add instead, and nothing is
    # returned for those parameters.
    # The FIRST view to arrive (.grad still None) used to hand autograd aliases of its static gradient buffers; the 118
    # AccumulateGrad nodes then run one by one on the engine thread -- each with an event record + stream wait, because
    # the parameters' accumulators live on the stream the script's warm-up ran on and this view's graph on another --
    # BEFORE the engine reaches the other view's node: ~1.5 ms by which the second backward graph started late
    # (tools/graphed_perf.py --profile).  The aliases are now installed as .grad right here (parameters with hooks keep
    # the autograd route).
    tgt, src, ret = [], [], []
    for p, g in zip(vg.params, vg.grads):
      # (parameters with hooks -- iic_amd.dist.GradReducer's post-accumulate hook, a user's register_hook -- keep the
      #  autograd route in BOTH cases: the fused paths would skip the hooks)
      hooked = bool(p._backward_hooks) or bool(getattr(p, "_post_accumulate_grad_hooks", None))
      if g is not None and p.grad is not None and FUSED_ACCUMULATE[0] and p.grad.dtype == g.dtype and not hooked:
        tgt.append(p.grad)
        src.append(g)
        ret.append(None)
      elif g is not None and p.grad is None and FUSED_ACCUMULATE[0] and not hooked:
        p.grad = g.detach()
        ret.append(None)
      else:
        ret.append(None if g is None else g.detach())
    cur = torch.cuda.current_stream()
    if tgt:
      # the .grad buffers being added to were written by ANOTHER view's backward graph, possibly on another stream
      # (auto_branch): autograd's AccumulateGrad would have synchronised with it, so must this
      mod = vg.mod_ref()
      for other in (_state(mod)["graphs"].values() if mod is not None else ()):
        if other is not vg and other is not _FAILED and other.bwd_event is not None:
          cur.wait_event(other.bwd_event)
      if ctx.caller_stream != cur:
        # (the other view may have run EAGERLY on the caller's stream -- its capture failed, say -- and left no event:
        #  everything the caller's stream has been given so far includes that view's gradient accumulation)
        cur.wait_stream(ctx.caller_stream)
      torch._foreach_add_(tgt, src)
    # Whoever reads .grad next -- the script's optimiser, on the stream the script is on -- must come after this view's
    # graph and fold.  The engine only orders the caller's stream after LEAF streams (AccumulateGrad nodes), and none of
    # the gradients above went through one: without this wait the optimiser raced the side-stream view's backward as soon
    # as the host got ahead of the GPU (found in round 4 by test_graphed_two_streams_are_ordered_before_the_optimiser...:
    # wrong losses from the seventh step on).
    if ctx.caller_stream != cur:
      ctx.caller_stream.wait_event(cur.record_event())
    return (None, None) + tuple(ret)


def _state(mod):
  st = mod.__dict__.get("_iic_graphed")
  if st is None:
    st = {"epoch": None, "pos": 0, "graphs": {}, "warm": {}, "res": {}}
    mod.__dict__["_iic_graphed"] = st
  return st


def eligible(mod, x, args, kwargs):
  return (ops.GRAPH_FORWARD[0] and mod.training and torch.is_grad_enabled() and torch.is_tensor(x) and x.is_cuda
          and not x.requires_grad and not args and set(kwargs) <= {"head"}
          and (ops.BRANCH[0] == 0 or ops.BRANCH[0] in ops._NO_PROXY_BRANCHES)    # (parameters, not leaf aliases)
          and ops.PT_DTYPE[0] is ops.BF16 and not torch.cuda.is_current_stream_capturing())


def _epoch(mod):
  """Changes whenever the optimiser has stepped: iic_amd.optim.Adam bumps the weights epoch (its kernels write
  through raw pointers), a torch optimiser bumps the parameters' version counters."""
  p = next((q for q in mod.parameters() if q.requires_grad), None)
  return (_cl._WEIGHTS_EPOCH[0], None if p is None else p._version)


class _Plan(object):
  __slots__ = ("st", "key", "res", "vg", "mode")


def plan(mod, x, kwargs, branch):
  """What an eligible training forward that is about to run as branch `branch` will do: "replay" a captured graph,
  "capture" one, or run "eager" (a warm-up occurrence, a shape whose capture failed).  Called by ops.auto_branch BEFORE
  it enters the branch, because the two cases want different parameter handling there: a replayed / captured view
  accumulates its gradients itself (_GraphedFn.backward), an eager side-stream view must see the parameters through
  leaf aliases (see ops.auto_branch).  Advances the step position."""
  st = _state(mod)
  ep = _epoch(mod)
  if st["epoch"] != ep:                      # the optimiser stepped: a new step begins
    st["epoch"], st["pos"] = ep, 0
  pos = st["pos"]
  st["pos"] += 1
  key = (tuple(x.shape), x.dtype, x.device.index, kwargs.get("head"), pos, branch)
  # resource namespace of this position (see _ViewGraph): its real branch, unless an earlier position of the
  # step lives there already (one-stream runs) -- the eager warm-up occurrences use the same namespace, so
  # that its PT buffers exist (zero-filled ONCE) before the capture; a buffer first allocated inside a
  # capture would be zero-filled by every replay (measured: 1.2 ms per step for one view's activations)
  res = st["res"].get(key)
  if res is None:
    clash = any(k[:4] == key[:4] and k[4] < pos and r == branch for k, r in st["res"].items())
    res = st["res"][key] = (100 + pos) if clash else branch
    if clash:
      ops._NO_PROXY_BRANCHES.add(res)        # a namespace, not a stream branch: parameters stay themselves
  vg = st["graphs"].get(key)
  if vg is not None and vg is not _FAILED and _sig_changed(mod, st, vg):
    # the module's storage moved (.cpu() / .cuda() / .to()): every graph of this module is stale
    if os.environ.get("IIC_GRAPH_LOG"):
      sys.stderr.write("[iic_amd.graphed] parameters moved: dropped %d captured graphs\n" % len(st["graphs"]))
    st["graphs"].clear()
    st["warm"].clear()
    st.pop("sig_tensors", None)
    vg = None
  pl = _Plan()
  pl.st, pl.key, pl.res, pl.vg = st, key, res, vg
  if vg is _FAILED:
    pl.mode = "eager"
  elif vg is None:
    n = st["warm"].get(key, 0)
    if n < WARMUP:
      st["warm"][key] = n + 1
      pl.mode = "eager"
    else:
      pl.mode = "capture"
  else:
    pl.mode = "replay"
  return pl


def forward(fwd, mod, x, args, kwargs, pl=None):
  """Called by ops.auto_branch's wrapper for an eligible training forward (inside the branch context when
  there is one).  Returns the forward's result, from a replayed graph when this key has been captured."""
  if pl is None:
    pl = plan(mod, x, kwargs, ops.BRANCH[0])
  st, key, res, vg = pl.st, pl.key, pl.res, pl.vg

  def eager():
    main = ops._BRANCH_MAIN[0]
    if main is not None and torch.cuda.current_stream() != main:
      # A capture failed INSIDE a forked branch (ops.auto_branch entered it because a captured view was planned).  Eager
      # launches do not run on the side stream (their gradient accumulation would cross the two streams): leave it --
      # the caller's stream waits for what this view has queued so far, the eager forward runs there with the
      # parameters themselves, and the fork stays pending, so the pair's second forward does not fork in its place.
      sys.stderr.write("[iic_amd.graphed] eager forward of %r runs on the caller's stream\n" % (key,))
      main.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(main):
        return eager_here()
    return eager_here()

  def eager_here():
    if res == ops.BRANCH[0]:
      return fwd(mod, x, *args, **kwargs)
    prev = ops.BRANCH[0]
    ops.BRANCH[0] = res
    try:
      return fwd(mod, x, *args, **kwargs)
    finally:
      ops.BRANCH[0] = prev
  if pl.mode == "eager":
    return eager()
  if pl.mode == "capture":
    n_def = len(ops._DEFERRED_RUNNING)
    try:
      vg = _ViewGraph(fwd, mod, x, args, kwargs, res)
    except Exception as e:                   # noqa: BLE001 -- never take a run down over an optimisation
      sys.stderr.write("[iic_amd.graphed] capture failed for %r (%s: %s): eager launches for this shape\n"
                       % (key, type(e).__name__, e))
      st["graphs"][key] = _FAILED
      # Undo what the aborted capture left on the host: running-statistic updates it queued (their kernels never
      # ran), operand sets whose re-layout was only recorded (a fresh weights epoch makes every holder re-lay
      # them eagerly), and the step position (the capture bumped the epoch once already).
      del ops._DEFERRED_RUNNING[n_def:]
      _cl.bump_weights_epoch()
      st["epoch"] = _epoch(mod)
      return eager()
    st["graphs"][key] = vg
    if os.environ.get("IIC_GRAPH_LOG"):
      sys.stderr.write("[iic_amd.graphed] captured forward + backward graphs for %r\n" % (key,))
    # the capture bumped the weights epoch (so that the re-layout kernels are part of the graph); this
    # forward still belongs to the step that was running
    st["epoch"] = _epoch(mod)
  outs = _GraphedFn.apply(vg, x, *vg.params)
  outs = list(outs)
  if vg.out_type is list:
    ops.tag_pack(outs)          # (the eager forward tags its list the same way: iic_amd.losses.IID_loss batches the sub-heads)
  return outs[0] if vg.out_type is None else vg.out_type(outs)
