"""Thin torch-tensor wrappers over the C ABI (include/iic_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every arithmetic op on
the hot path is a kernel of libiic_hip.so.  All wrappers enqueue on torch's current stream.
"""
import ctypes
import os
import threading
import weakref

import torch

from . import _lib
from ._lib import IIC_STAT_STRIPES, check, lib, ptr, stream_ptr

BF16 = torch.bfloat16
F32 = torch.float32


# ------------------------------------------------------------------------------------
# Branches: two independent forwards of a step (net(all_imgs), net(all_imgs_tf):
# cluster_sobel.py:238-239) can run on two HIP streams -- as two parallel branches of the captured
# step graph -- so that the tail of one view's kernel is filled by the other view's next kernel
# (measured: 42.6 -> 38.2 ms for the two forward+backward passes, tools/graph_probes.py dual_branch_probe).
# Everything the kernels share through HBM is therefore keyed by the branch index: PT buffers,
# BatchNorm statistic accumulators, split-K / stem scratch, the bf16 weight operands.  A branch's
# autograd Functions remember their branch (ctx.branch) and restore it in backward, where the
# engine already runs them on the stream they were recorded on.
#   with ops.branch():            # fork: side stream waits for the current one
#     xo = net(all_imgs)          # the view that comes FIRST is enqueued on the side stream
#   xt = net(all_imgs_tf)         # main stream, concurrently
#   ops.join()                    # the current stream waits for the side stream (the losses and
#                                 # the optimiser call it themselves if it is still pending)
# Inside a branch: parameters are seen through per-branch leaf aliases (`pv`), so that gradient
# accumulation of shared parameters never synchronises the branches (the optimiser adds the two
# gradient sets, iic_amd.optim.Adam); while a branch is pending, BatchNorm running-statistic
# updates of BOTH views are postponed to the join and applied there in call order -- the order a
# sequential run would have used (bn_finalize -> _DEFERRED_RUNNING).
# ------------------------------------------------------------------------------------
class BranchContext(object):
  """Everything the two-stream execution keeps between calls, in ONE object per thread (round 6; until round 5 these
  were nine module-level lists / dicts, which is the style that produced round 5's read-before-join race): the branch
  the calling code is on, the side streams, the per-branch leaf aliases of the parameters, the forks not joined yet
  and the running-statistic updates they postponed.  One process drives one device (the data-parallel design: one
  process per GPU), so a context is per thread, and its streams are keyed by device inside.  The autograd engine runs
  backward nodes on its own threads: every Function records its branch at forward time (ctx.branch) and
  `branch_backward` restores it there, so a backward thread never depends on the forward thread's context.
  The module-level names below (BRANCH[0], _PENDING_JOIN, ...) are views of the calling thread's context."""
  __slots__ = ("branch", "streams", "proxies", "deferred_running", "pending_join", "branch_main", "no_proxy_branches",
               "capture_proxies", "solo_first")

  def __init__(self):
    self.branch = 0                 # BRANCH[0]: 0 = the caller's stream, >= 1 = a side branch / resource namespace
    self.streams = {}               # _BRANCH_STREAM: (device, branch index) -> side stream
    self.proxies = {}               # _PROXIES: id(param) -> (param, {branch: leaf alias sharing its storage})
    self.deferred_running = []      # _DEFERRED_RUNNING: (coef, running_mean, running_var, num_batches_tracked, C)
    self.pending_join = []          # _PENDING_JOIN: (main stream, side stream) of branches not joined yet
    self.branch_main = None         # _BRANCH_MAIN[0]: inside `with branch():` the stream the caller was on
    self.no_proxy_branches = set()  # _NO_PROXY_BRANCHES (see below)
    self.capture_proxies = None     # _CAPTURE_PROXIES[0]: iic_amd.graphed, {id(param): leaf alias} during a capture
    self.solo_first = 0             # _SOLO_FIRST[0] (see auto_branch)


_TLS = threading.local()


def context():
  """The calling thread's BranchContext."""
  c = getattr(_TLS, "ctx", None)
  if c is None:
    c = _TLS.ctx = BranchContext()
  return c


class _CtxCell(object):
  """`NAME[0]` view of a scalar field of the calling thread's context."""
  __slots__ = ("_attr",)

  def __init__(self, attr):
    self._attr = attr

  def __getitem__(self, i):
    return getattr(context(), self._attr)

  def __setitem__(self, i, v):
    setattr(context(), self._attr, v)


class _CtxView(object):
  """Container view (list / dict / set) of a field of the calling thread's context."""
  __slots__ = ("_attr",)

  def __init__(self, attr):
    self._attr = attr

  def _o(self):
    return getattr(context(), self._attr)

  def __getattr__(self, name):
    return getattr(self._o(), name)

  def __len__(self):
    return len(self._o())

  def __bool__(self):
    return bool(self._o())

  def __iter__(self):
    return iter(self._o())

  def __contains__(self, x):
    return x in self._o()

  def __getitem__(self, k):
    return self._o()[k]

  def __setitem__(self, k, v):
    self._o()[k] = v

  def __delitem__(self, k):
    del self._o()[k]


BRANCH = _CtxCell("branch")
_BRANCH_STREAM = _CtxView("streams")
_PROXIES = _CtxView("proxies")                    # id(param) -> (param, {branch: leaf alias sharing its storage})
_DEFERRED_RUNNING = _CtxView("deferred_running")  # (coef, running_mean, running_var, num_batches_tracked, C) of a branch
_PENDING_JOIN = _CtxView("pending_join")          # (main stream, side stream) of branches not joined yet
_BRANCH_MAIN = _CtxCell("branch_main")            # inside `with branch():` the stream the caller was on (iic_amd.graphed orders it after a view's backward)


# IIC_BRANCH_PROXIES=0 (debugging only): side branches use the parameters themselves and autograd accumulates both
# views' gradients into p.grad ACROSS the two streams.
USE_PROXIES = [os.environ.get("IIC_BRANCH_PROXIES", "1") != "0"]


# Branch indices that see the parameters themselves (no leaf aliases): the side stream of ops.auto_branch -- only
# captured / replayed views run there, and those hand their gradients over explicitly (iic_amd/graphed.py) -- and the
# resource namespaces iic_amd.graphed gives to positions that share a real branch.
# (Rounds 3-4 also had an EAGER two-stream mode for unchanged scripts -- leaf aliases plus an end-of-backward fold of the
# alias gradients into .grad.  It was demoted to opt-in in round 4 because about 1 run in 13 differed from the one-stream
# run; round 5 found the cause -- the loss stacked the forked view's outputs before joining it, iic_amd/losses.py -- and
# removed the mode rather than carry a second gradient hand-over for launches that are host-bound anyway.)
_NO_PROXY_BRANCHES = _CtxView("no_proxy_branches")


_CAPTURE_PROXIES = _CtxCell("capture_proxies")   # iic_amd.graphed: {id(param): leaf alias} while a view's graphs are captured


def pv(p):
  """Parameter as seen by the current branch (the parameter itself on the main branch)."""
  cp = _CAPTURE_PROXIES[0]
  if cp is not None and p is not None:
    q = cp.get(id(p))
    if q is not None:
      return q
  b = BRANCH[0]
  if b == 0 or p is None or not p.requires_grad or not USE_PROXIES[0] or b in _NO_PROXY_BRANCHES:
    return p
  ent = _PROXIES.get(id(p))
  if ent is None or ent[0] is not p:
    ent = (p, {})
    _PROXIES[id(p)] = ent
  q = ent[1].get(b)
  if q is None or q.data_ptr() != p.data_ptr():
    q = p.detach().requires_grad_(True)
    ent[1][b] = q
  return q


def branch_leaf(p, index):
  """The autograd leaf branch `index` saw for parameter p (its alias, or p itself when the branch ran
  without aliases / never touched p)."""
  ent = _PROXIES.get(id(p))
  if ent is None or ent[0] is not p or index in _NO_PROXY_BRANCHES:
    return p
  return ent[1].get(index, p)


def branch_grads(p):
  """Gradients the side branches accumulated for parameter p (list, possibly empty)."""
  ent = _PROXIES.get(id(p))
  if ent is None or ent[0] is not p:
    return []
  return [q.grad for q in ent[1].values() if q.grad is not None]


def fold_branch_grads(params):
  """p.grad += the gradients the side branches accumulated for p (and drop those): for callers that
  need ONE gradient per parameter before the optimiser -- e.g. a gradient all-reduce."""
  tgt, src = [], []
  for p in params:
    for g in branch_grads(p):
      if p.grad is None:
        p.grad = g.clone()
      else:
        tgt.append(p.grad)
        src.append(g)
  if tgt:
    torch._foreach_add_(tgt, src)
  clear_branch_grads()


def clear_branch_grads():
  for _, d in _PROXIES.values():
    for q in d.values():
      q.grad = None


class branch(object):
  """Fork the enclosed forward onto a side stream / graph branch (see above).  Not re-entrant."""

  def __init__(self, index=1, proxies=True):
    assert index >= 1
    self.index = index
    self.proxies = proxies

  def __enter__(self):
    assert BRANCH[0] == 0, "branches do not nest"
    (_NO_PROXY_BRANCHES.discard if self.proxies else _NO_PROXY_BRANCHES.add)(self.index)
    dev = torch.cuda.current_device()
    key = (dev, self.index)
    self.main = torch.cuda.current_stream()
    st = _BRANCH_STREAM.get(key)
    if st is None:
      # HIP multiplexes streams onto a few hardware queues: keep a side stream that really runs beside the
      # caller's and is not parked behind the process group's collectives (iic_amd.graph.pick_stream; a pair on
      # one queue would run the two views one after the other)
      from .graph import pick_stream
      st = _BRANCH_STREAM[key] = pick_stream((self.main,), "side stream of ops.branch", collective_free=True)
    self.side = st
    st.wait_stream(self.main)                       # fork
    for _, d in _PROXIES.values():                  # last step's branch gradients are consumed
      q = d.get(self.index)
      if q is not None:
        q.grad = None
    self.ctx = torch.cuda.stream(st)
    self.ctx.__enter__()
    BRANCH[0] = self.index
    _BRANCH_MAIN[0] = self.main
    return self

  def __exit__(self, *exc):
    BRANCH[0] = 0
    _BRANCH_MAIN[0] = None
    self.ctx.__exit__(*exc)
    _PENDING_JOIN.append((self.main, self.side))    # joined later: the main view runs meanwhile
    if exc and exc[0] is not None:
      join()            # the forward raised: nobody downstream will join -- do not leave the fork pending
    return False


class on_branch(object):
  """Low-level: run the enclosed code as branch `index` on `stream` WITHOUT any fork / join
  synchronisation (iic_amd.graph.CapturedPairStep orders its per-view graphs itself)."""

  def __init__(self, index, stream):
    self.index, self.stream = index, stream

  def __enter__(self):
    assert BRANCH[0] == 0, "branches do not nest"
    # (this view sees the parameters through leaf aliases whatever an earlier auto_branch run left behind: with the
    #  index still marked alias-free, its gradients would be accumulated into .grad across the two streams)
    _NO_PROXY_BRANCHES.discard(self.index)
    self.ctx = torch.cuda.stream(self.stream)
    self.ctx.__enter__()
    BRANCH[0] = self.index
    return self

  def __exit__(self, *exc):
    BRANCH[0] = 0
    self.ctx.__exit__(*exc)
    return False


# Automatic two-stream execution for UNCHANGED training scripts (IIC_AUTO_BRANCH=1; `python -m
# iic_amd.run` switches it on).  The reference's step calls net(all_imgs) and net(all_imgs_tf) one
# after the other (cluster_sobel.py:238-239, segmentation_twohead.py:300-306) and hands both
# results to the loss: the first training forward since the last join is put on the side stream,
# the second runs on the caller's stream meanwhile, and the loss (ours) joins.  No parameter
# aliases here -- autograd accumulates both views into p.grad, so any optimiser works.  Contract:
# the outputs of the first forward must not be consumed by anything but this library's losses
# before the join (true of every reference script); evaluation / no_grad forwards never branch.
AUTO_BRANCH = [os.environ.get("IIC_AUTO_BRANCH", "0") == "1"]
# A forward of the pair that runs EAGERLY -- graph replay off, a warm-up occurrence, a shape that was not captured --
# stays on the caller's stream; the two streams are for captured / replayed views, whose gradient hand-over is explicit
# (iic_amd/graphed.py).
_SOLO_FIRST = _CtxCell("solo_first")   # 1: the first forward of a pair ran on the caller's stream (the second one must not fork either)
# forwards that hand back FEATURES (semisup heads: sup_head5.py:34-35, net6c_two_head.py:78-94,
# k-means feature extraction) are consumed by modules outside this library, which know nothing about
# the side stream: they never branch
_FEATURE_FLAGS = ("trunk_features", "penultimate_features", "kmeans_use_features")
# running-statistic updates postponed by branch forwards (bn_finalize): flushed at every join; the
# optimiser and the losses join.  A caller that does neither would grow the list without bound and
# evaluate on stale running statistics -- past this many entries the next forward joins by itself.
_DEFERRED_LIMIT = 1024


class HeadPack(object):
  """What the sub-head outputs of ONE forward share: the reference hands the loss a python list of per-sub-head
  [bn, k] tensors (net5g.py:76-80) and calls IID_loss once per sub-head (cluster_sobel.py:241-253).  The list's
  tensors are tagged with their pack and index so that iic_amd.losses.IID_loss can evaluate all sub-head pairs of
  two packs in ONE set of launches at the first call and hand the other calls their share."""
  __slots__ = ("tensors", "cache", "__weakref__")

  def __init__(self, tensors):
    # weak references: a tensor -> pack -> tensor cycle would keep a step's autograd graph (and with it ~18 GB of
    # saved activations at the north-star batch) alive until the cyclic garbage collector happens to run
    self.tensors = [weakref.ref(t) for t in tensors]
    self.cache = {}

  def alive(self):
    ts = [r() for r in self.tensors]
    return ts if all(t is not None for t in ts) else None


def tag_pack(tensors):
  """Tag a list of per-sub-head output tensors (returns the list)."""
  if len(tensors) > 1 and all(torch.is_tensor(t) and t.dim() == 2 for t in tensors):
    pack = HeadPack(tensors)
    for i, t in enumerate(tensors):
      t._iic_pack = (pack, i)
  return tensors


# Graph replay of the training forwards / backwards for unchanged scripts (iic_amd/graphed.py);
# `python -m iic_amd.run` switches it on (IIC_GRAPH_FORWARD=0 keeps eager launches).
GRAPH_FORWARD = [os.environ.get("IIC_GRAPH_FORWARD", "0") == "1"]


def auto_branch(fwd):
  """Decorator for the architectures' forward()."""
  def wrapped(self, x, *a, **k):
    if BRANCH[0] == 0 and (_PENDING_JOIN or _DEFERRED_RUNNING) and (
        not self.training or not torch.is_grad_enabled() or len(_DEFERRED_RUNNING) > _DEFERRED_LIMIT):
      join()      # evaluation must see up-to-date running statistics; bound the postponed list
    run = fwd
    will_branch = (AUTO_BRANCH[0] and self.training and torch.is_grad_enabled() and BRANCH[0] == 0
                   and not _PENDING_JOIN and not _SOLO_FIRST[0] and torch.is_tensor(x) and x.is_cuda
                   and not any(k.get(f) for f in _FEATURE_FLAGS))
    pl = None
    if GRAPH_FORWARD[0]:
      from . import graphed
      if graphed.eligible(self, x, a, k):
        pl = graphed.plan(self, x, k, 1 if will_branch else BRANCH[0])

        def run(self_, x_, *a_, **k_):      # captured-graph replay once this (shape, head, position) is warm
          return graphed.forward(fwd, self_, x_, a_, k_, pl)
    if will_branch:
      if pl is None or pl.mode == "eager":
        # eager launches stay on the caller's stream.  A planned position keeps its resource namespace (so that its
        # buffers exist before the capture) but sees the parameters themselves; the pair's second forward must not
        # fork in its place, and the running-statistic updates of both are applied at the join, in call order.
        _SOLO_FIRST[0] = 1
        if pl is not None:
          _NO_PROXY_BRANCHES.add(pl.res)
        try:
          return run(self, x, *a, **k)
        except BaseException:
          _SOLO_FIRST[0] = 0              # the forward raised: there will be no second view to wait for
          raise
      # a replayed / captured view: side stream, the parameters themselves, gradients handed over by its autograd node
      with branch(proxies=False) as br:
        x.record_stream(br.side)         # allocated on the caller's stream, consumed on the side stream
        return run(self, x, *a, **k)
    if _SOLO_FIRST[0] == 1 and self.training and torch.is_grad_enabled() and BRANCH[0] == 0:
      # the second view of a pair whose first view stayed on the caller's stream: its running-statistic updates are
      # postponed like the first view's (bn_finalize) and the pair ends with it.  Both views ran on the caller's
      # stream, so nothing is left to wait for: the postponed updates are applied right here, in call order -- a
      # state_dict() / checkpoint taken after the step sees them whoever owns the loss and the optimiser
      try:
        return run(self, x, *a, **k)
      finally:
        join()
    if not torch.is_grad_enabled():
      mark = POOL.mark()                 # evaluation: nothing will release the activations later
      try:
        return fwd(self, x, *a, **k)
      finally:
        POOL.sweep(mark)
    return run(self, x, *a, **k)
  wrapped.__name__ = getattr(fwd, "__name__", "forward")
  wrapped.__doc__ = fwd.__doc__
  wrapped.__wrapped__ = fwd          # inspect.signature() shows the architecture's own parameters
  return wrapped


def join():
  """Main stream waits for the side branches forked since the last join, then applies their
  postponed running-statistic updates (after the main view's own: sequential order)."""
  while _PENDING_JOIN:
    main, side = _PENDING_JOIN.pop()
    main.wait_stream(side)
  _SOLO_FIRST[0] = 0
  flush_deferred_running()


def flush_deferred_running():
  """Apply the running-statistic updates the branch forwards postponed (one launch)."""
  if not _DEFERRED_RUNNING:
    return
  items = list(_DEFERRED_RUNNING)
  del _DEFERRED_RUNNING[:]
  n = len(items)
  VP = ctypes.c_void_p * n
  IP = ctypes.c_int * n
  check(lib().iic_bn_running_update(
    n, VP(*[ptr(c) for c, _, _, _, _ in items]), VP(*[ptr(rm) for _, rm, _, _, _ in items]),
    VP(*[ptr(rv) for _, _, rv, _, _ in items]), VP(*[ptr(nb) for _, _, _, nb, _ in items]),
    IP(*[C for _, _, _, _, C in items]), BN_MOMENTUM, stream_ptr()), "iic_bn_running_update")


def branch_backward(fn):
  """Decorator for autograd Function.backward: run under the branch the forward recorded."""
  def wrapped(ctx, *grads):
    if getattr(ctx, "_iic_ran", False):
      raise RuntimeError("iic_amd: backward through this graph a second time -- the saved activation "
                         "buffers went back to the pool after the first pass (retain_graph is not "
                         "supported on the HIP path)")
    ctx._iic_ran = True
    prev, prev_dt = BRANCH[0], PT_DTYPE[0]
    BRANCH[0] = getattr(ctx, "branch", 0)
    PT_DTYPE[0] = getattr(ctx, "pt_dtype", prev_dt)     # (fp32_mode() forwards allocate fp32 in backward too)
    try:
      return fn(ctx, *grads)
    finally:
      BRANCH[0], PT_DTYPE[0] = prev, prev_dt
  wrapped.__name__ = getattr(fn, "__name__", "backward")
  return staticmethod(wrapped)


# ------------------------------------------------------------------------------------
# PT ("padded tile") activation buffers: bf16 [N, H+2P, W+2P, C], zero border.
# Kernels write interiors only, so buffers are zeroed ONCE and recycled through a pool
# (no per-step memset traffic).  A buffer is handed out by `alloc`, and returned with
# `release` once every kernel that reads it has been enqueued (stream-ordered reuse).
# ------------------------------------------------------------------------------------
# Storage type of PT tensors: bf16 (the product path) or, inside `with fp32_mode():`, fp32 -- the
# exact-fp32 parity path of csrc/f32_path.hip (SURVEY.md §8c tier T2): same orchestration, plain
# fp32 kernels, for whole-network comparisons with the reference's fp32 results.  Every wrapper
# below dispatches on the dtype of the tensors it is handed.
PT_DTYPE = [BF16]


class fp32_mode(object):
  def __enter__(self):
    self.prev = PT_DTYPE[0]
    PT_DTYPE[0] = F32
    return self

  def __exit__(self, *exc):
    PT_DTYPE[0] = self.prev
    return False


class PTPool(object):
  """Buffers are keyed by (shape, border P, device, branch, dtype): a recycled buffer is only valid
  for a tensor with the SAME interior/border split (its border must still be zero), and only on
  the stream (branch) whose kernels used it last.

  The pool keeps a reference to every buffer it created (`owned`), so an address can never come
  back from the caching allocator as some other tensor and be mistaken for a zero-border buffer;
  `release` only accepts buffers that are currently handed out (a second release -- e.g. a
  backward run twice -- is ignored instead of putting one buffer on the free list twice)."""

  def __init__(self):
    self.free = {}
    self.owned = {}           # data_ptr -> (tensor, P, branch) of every buffer this pool created
    self.live = {}            # data_ptr -> serial of the alloc() that handed it out
    self.serial = 0
    self.allocated_bytes = 0

  def alloc(self, shape, device, P=1):
    dt = PT_DTYPE[0]
    key = (tuple(shape), int(P), str(device), BRANCH[0], dt)
    lst = self.free.get(key)
    if lst:
      t = lst.pop()
    else:
      t = torch.zeros(shape, dtype=dt, device=device)
      self.owned[t.data_ptr()] = (t, int(P), BRANCH[0])
      self.allocated_bytes += t.numel() * t.element_size()
    self.serial += 1
    self.live[t.data_ptr()] = self.serial
    # a fresh tensor object per hand-out: the caller's autograd state (grad_fn of the Function that
    # returns it, user hooks registered on it) must not survive into the buffer's next life
    return t.detach()

  def release(self, t):
    if t is None:
      return
    ent = self.owned.get(t.data_ptr())
    if ent is None or ent[0].shape != t.shape:
      return                  # not one of ours (a user tensor, a view): never recycle it
    if self.live.pop(t.data_ptr(), None) is None:
      return                  # already back in the pool
    key = (tuple(t.shape), ent[1], str(t.device), ent[2], t.dtype)
    self.free.setdefault(key, []).append(ent[0])

  def mark(self):
    return self.serial

  def sweep(self, mark):
    """Return every buffer handed out since `mark` that is still out: the end of a forward that
    no backward will follow (torch.no_grad evaluation), whose consumers are all enqueued."""
    for dp in [dp for dp, ser in self.live.items() if ser > mark]:
      self.release(self.owned[dp][0])

  def clear(self):
    self.free.clear()
    self.owned.clear()
    self.live.clear()


POOL = PTPool()


def pt_alloc(N, H, W, C, P, device):
  return POOL.alloc((N, H + 2 * P, W + 2 * P, C), device, P)


def pt_from_nchw(x, P):
  """(test / boundary helper) NCHW float tensor -> PT (bf16, or fp32 inside fp32_mode())."""
  n, c, h, w = x.shape
  out = torch.zeros((n, h + 2 * P, w + 2 * P, c), dtype=PT_DTYPE[0], device=x.device)
  out[:, P:P + h, P:P + w, :] = x.permute(0, 2, 3, 1).to(PT_DTYPE[0])
  return out


def pt_to_nchw(x, P):
  n, hp, wp, c = x.shape
  return x[:, P:hp - P, P:wp - P, :].permute(0, 3, 1, 2).float().contiguous()


def new_stats(C, device):
  """Zeroed statistics accumulator for C channels (opaque exact fixed-point cells, see
  include/iic_hip.h: iic_stat_bytes)."""
  return torch.zeros(lib().iic_stat_bytes(C) // 8, dtype=torch.int64, device=device)


STAT_BINS, STAT_LSB0, STAT_SPACING = 8, -96, 24      # csrc/common.h


def stats_decode(st, C):
  """(test / debug helper, non-destructive) accumulator -> float64 [2, C] sums."""
  cells = st.view(IIC_STAT_STRIPES, C, 2, STAT_BINS).sum(0)          # exact int64
  val = torch.zeros((C, 2), dtype=torch.float64, device=st.device)
  for b in range(STAT_BINS - 2, -1, -1):
    val += torch.ldexp(cells[..., b].double(), torch.tensor(STAT_LSB0 + STAT_SPACING * b, device=st.device))
  val[cells[..., STAT_BINS - 1] != 0] = float("nan")
  return val.t().contiguous()


def stats_encode(st, C, values):
  """(test helper) overwrite the accumulator with the float32 sums `values` [2, C]."""
  st.zero_()
  v = values.to(st.device).float().t().contiguous().double()          # [C, 2]
  mant, exp = torch.frexp(v)
  m = torch.round(mant * (1 << 24)).long()
  pos = exp.long() - 24 - STAT_LSB0
  neg = pos < 0
  m = torch.where(neg, m >> (-pos).clamp(0, 62), m)
  pos = pos.clamp(min=0)
  b, sh = pos // STAT_SPACING, pos % STAT_SPACING
  cells = st.view(IIC_STAT_STRIPES, C, 2, STAT_BINS)
  cells[0].scatter_(2, b.clamp(max=STAT_BINS - 2).unsqueeze(-1), (m << sh).unsqueeze(-1))


# ------------------------------------------------------------------------------------
# conv
# ------------------------------------------------------------------------------------
def weight_prep(w, want_bwd=True):
  """fp32 OIHW parameter -> (bf16 [T][Co][Ci], bf16 [T][Ci][Co])  (row-major operands)."""
  co, ci, kh, kw = w.shape
  T = kh * kw
  wf = torch.empty((T, co, ci), dtype=BF16, device=w.device)
  wb = torch.empty((T, ci, co), dtype=BF16, device=w.device) if want_bwd else None
  check(lib().iic_weight_prep(ptr(w), ptr(wf), ptr(wb), co, ci, T, stream_ptr()), "iic_weight_prep")
  return wf, wb


def weight_prep_frag(w, bwd):
  """fp32 OIHW parameter -> bf16 MFMA-B-fragment order (include/iic_hip.h, iic_weight_prep_frag)."""
  co, ci, kh, kw = w.shape
  out = torch.empty((kh * kw * co * ci,), dtype=BF16, device=w.device)
  check(lib().iic_weight_prep_frag(ptr(w), ptr(out), co, ci, kh * kw, 1 if bwd else 0, stream_ptr()),
        "iic_weight_prep_frag")
  return out


# IIC_CONV_FRAG=0 keeps every conv on the first-generation kernel (row-major weight operand).
USE_FRAG = [os.environ.get("IIC_CONV_FRAG", "1") != "0"]


class PreppedWeights(object):
  """bf16 operands of one conv parameter, laid out lazily per consumer kernel.  `pw[0]` is the
  forward operand, `pw[1]` the backward-data operand (handles accepted by conv_igemm).

  The layouts a consumer asked for once are kept (same buffers) and RE-WRITTEN IN PLACE when the parameter
  has changed: `jobs()` lists them for the one-launch refresh of all convolutions of a network
  (refresh_prepped; archs.cluster._ConvHolder.weights)."""

  def __init__(self, w):
    self.w = w
    self._rows = None
    self._frag = [None, None]

  def rows(self, bwd):
    if self._rows is None:
      self._rows = weight_prep(self.w, want_bwd=True)
    return self._rows[1 if bwd else 0]

  def jobs(self):
    """(src ptr, dst ptr, Cout, Cin, T, mode) of every materialised layout (modes: include/iic_hip.h)."""
    co, ci, kh, kw = self.w.shape
    out = []
    for k in (0, 1):
      if self._frag[k] is not None:
        out.append((self.w.data_ptr(), self._frag[k].data_ptr(), co, ci, kh * kw, k))
    if self._rows is not None:
      for k in (0, 1):
        if self._rows[k] is not None:
          out.append((self.w.data_ptr(), self._rows[k].data_ptr(), co, ci, kh * kw, 2 + k))
    return out

  def frag(self, bwd):
    k = 1 if bwd else 0
    if self._frag[k] is None:
      self._frag[k] = weight_prep_frag(self.w, bwd)
    return self._frag[k]

  def __getitem__(self, i):
    return WOperand(self, bool(i))


class _PrepJob(ctypes.Structure):
  _fields_ = [("w", ctypes.c_void_p), ("out", ctypes.c_void_p), ("first_block", ctypes.c_longlong),
              ("Cout", ctypes.c_int32), ("Cin", ctypes.c_int32), ("T", ctypes.c_int32), ("mode", ctypes.c_int32)]


_PREP_TABLES = {}      # (device index, tuple of jobs) -> (device table, njobs, total blocks)
_PREP_PINNED = set()   # keys whose launch was recorded into a HIP graph: the table's address is baked in
MULTI_PREP = [os.environ.get("IIC_MULTI_PREP", "1") != "0"]


def refresh_prepped(pws, device):
  """Re-write every materialised layout of the given PreppedWeights from their (updated) fp32 parameters in
  ONE launch (iic_weight_prep_multi).  The job table lives in device memory and is cached by content: a
  network's jobs are the same every step, so the upload happens once, in the eager warm-up steps.  Returns
  False when nothing was launched because a NEW table would have to be uploaded while a stream is being
  captured (the caller then falls back to per-layout launches)."""
  jobs = tuple(j for pw in pws for j in pw.jobs())
  if not jobs:
    return True
  key = (device.index, jobs)
  ent = _PREP_TABLES.get(key)
  if ent is None:
    if torch.cuda.is_current_stream_capturing():
      return False
    arr = (_PrepJob * len(jobs))()
    blk = 0
    for a, (src, dst, co, ci, t, mode) in zip(arr, jobs):
      a.w, a.out, a.first_block, a.Cout, a.Cin, a.T, a.mode = src, dst, blk, co, ci, t, mode
      blk += lib().iic_weight_prep_multi_blocks(co, ci, t)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    ent = _PREP_TABLES[key] = (host.to(device), len(jobs), blk)
    if len(_PREP_TABLES) > 256:         # (tables of networks that no longer exist; never one a graph may replay)
      for k in [k for k in list(_PREP_TABLES)[:-128] if k not in _PREP_PINNED]:
        del _PREP_TABLES[k]
  if torch.cuda.is_current_stream_capturing():
    _PREP_PINNED.add(key)
  check(lib().iic_weight_prep_multi(ptr(ent[0]), ent[1], ent[2], stream_ptr()), "iic_weight_prep_multi")
  return True


class WOperand(object):
  def __init__(self, pw, bwd):
    self.pw, self.bwd = pw, bwd


def frag_supported(g):
  if not USE_FRAG[0]:
    return False
  ok = getattr(g, "_frag_ok", None)     # geometry objects are cached per (layer, shape)
  if ok is None:
    ok = bool(lib().iic_conv_igemm_frag_supported(ctypes.byref(g)))
    g._frag_ok = ok
  return ok


ACC_ADD, ACC_PREMASK = 1, 2     # include/iic_hip.h IIC_ACC_*


def red_supported(g, w_t):
  """Can this backward-data launch carry a fused BatchNorm-backward reduction (`red=`)?"""
  if PT_DTYPE[0] is not BF16 or not (isinstance(w_t, WOperand) and frag_supported(g)):
    return False
  ok = getattr(g, "_red_ok", None)
  if ok is None:
    ok = bool(lib().iic_conv_igemm_red_supported(ctypes.byref(g)))
    g._red_ok = ok
  return ok


def conv_igemm(g, x_pt, w_t, out_pt, stats=None, res_grad=None, res_act=None, accumulate=False,
               premask=False, red=None):
  """w_t: a row-major bf16 operand tensor (first-generation kernel) or a WOperand handle
  (second-generation weights-direct kernel wherever the geometry supports it).
  premask: out = (value [+ previous] [+ res_grad]) where res_act > 0 else 0 (IIC_ACC_PREMASK).
  red = (y, mask_coef | None, sums, y2 | None, sums2 | None): fused BatchNorm-backward reduction
  over the stored tile (iic_conv_igemm_frag_red); only where red_supported(g, w_t)."""
  acc = (ACC_ADD if accumulate else 0) | (ACC_PREMASK if premask else 0)
  if x_pt.dtype == F32:       # exact-fp32 parity path: the fp32 OIHW parameter itself is the operand
    assert isinstance(w_t, WOperand) and red is None and out_pt.dtype == F32
    w = w_t.pw.w
    check(lib().iic_f32_conv(ctypes.byref(g), ptr(x_pt), ptr(w), w.shape[2] * w.shape[3], 1 if w_t.bwd else 0,
                             ptr(out_pt), ptr(stats), ptr(res_grad), ptr(res_act), acc, stream_ptr()),
          "iic_f32_conv")
    return out_pt
  if red is not None:
    assert red_supported(g, w_t), "fused reduction needs the weights-direct kernel"
    ry, rcoef, rsums, ry2, rsums2 = red
    check(lib().iic_conv_igemm_frag_red(ctypes.byref(g), ptr(x_pt), ptr(w_t.pw.frag(w_t.bwd)),
                                        ptr(out_pt), ptr(stats), ptr(res_grad), ptr(res_act), acc,
                                        ptr(ry), ptr(rcoef), ptr(ry2), ptr(rsums), ptr(rsums2),
                                        stream_ptr()), "iic_conv_igemm_frag_red")
    return out_pt
  if isinstance(w_t, WOperand):
    if frag_supported(g):
      check(lib().iic_conv_igemm_frag(ctypes.byref(g), ptr(x_pt), ptr(w_t.pw.frag(w_t.bwd)),
                                      ptr(out_pt), ptr(stats), ptr(res_grad), ptr(res_act), acc,
                                      stream_ptr()), "iic_conv_igemm_frag")
      return out_pt
    w_t = w_t.pw.rows(w_t.bwd)
  check(lib().iic_conv_igemm(ctypes.byref(g), ptr(x_pt), ptr(w_t), ptr(out_pt), ptr(stats),
                             ptr(res_grad), ptr(res_act), acc, stream_ptr()),
        "iic_conv_igemm")
  return out_pt


_WG_PART = {}


def _conv_wgrad_launch(g, x_pt, dy_pt, wtaps, use_tr, out, accumulate, ns, key):
  need = ns * g.ntaps * g.Cout * g.Cin
  part = _WG_PART.get(key)
  if part is None or part.numel() < need:
    part = torch.empty(max(need, 1 << 22), dtype=F32, device=x_pt.device)
    _WG_PART[key] = part
  check(lib().iic_conv_wgrad(ctypes.byref(g), ptr(x_pt), ptr(dy_pt), ptr(part), ns,
                             1 if use_tr else 0, stream_ptr()), "iic_conv_wgrad")
  check(lib().iic_conv_wgrad_reduce(ptr(part), ns, wtaps, g.Cout, g.Cin, ptr(out),
                                    1 if accumulate else 0, stream_ptr()), "iic_conv_wgrad_reduce")


def conv_wgrad(g, x_pt, dy_pt, wtaps, use_tr=True, out=None, accumulate=False, nsplit=None):
  """Returns dW fp32 [Co][Ci][kh][kw] flattened as [Co, Ci, wtaps].  nsplit: override of the
  split-K factor (tests: few splits = many K-tiles per workgroup)."""
  if x_pt.dtype == F32:
    if out is None:
      out = torch.empty((g.Cout, g.Cin, wtaps), dtype=F32, device=x_pt.device)
    check(lib().iic_f32_wgrad(ctypes.byref(g), ptr(x_pt), ptr(dy_pt), ptr(out), wtaps, 1 if accumulate else 0,
                              stream_ptr()), "iic_f32_wgrad")
    return out
  ns = int(nsplit) if nsplit else lib().iic_conv_wgrad_nsplit(ctypes.byref(g))
  if out is None:
    out = torch.empty((g.Cout, g.Cin, wtaps), dtype=F32, device=x_pt.device)
  assert g.ntaps == wtaps, "wgrad geometry must list every weight tap once"
  _conv_wgrad_launch(g, x_pt, dy_pt, wtaps, use_tr, out, accumulate, ns, (str(x_pt.device), BRANCH[0]))
  return out


# ------------------------------------------------------------------------------------
# batch norm
# ------------------------------------------------------------------------------------
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# Replica de-duplication (opt-in, archs.cluster.DEDUP): while a de-duplicated forward runs, the
# unbiased running_var factor is computed for the TRUE batch (unique rows x this factor).
BN_REPLICAS = [1]


def bn_finalize(stats, gamma, beta, running_mean, running_var, nbt, C, count, training):
  """coef [5][C]: scale, shift, mean, invstd, unbiased batch variance."""
  coef = torch.empty((5, C), dtype=F32, device=gamma.device)
  if training and running_mean is not None and (BRANCH[0] != 0 or _PENDING_JOIN or _SOLO_FIRST[0]):
    # a side branch is (or may still be) running: both views update the same running statistics,
    # so every update is postponed to the join and applied there in CALL order -- the order a
    # sequential run would have used (fork the view that comes first in the script)
    _DEFERRED_RUNNING.append((coef, running_mean, running_var, nbt, C))
    running_mean = running_var = nbt = None
  check(lib().iic_bn_finalize(ptr(stats), ptr(gamma), ptr(beta), ptr(running_mean),
                              ptr(running_var), ptr(nbt), ptr(coef), C, count,
                              count * BN_REPLICAS[0], BN_EPS, BN_MOMENTUM, 1 if training else 0,
                              stream_ptr()), "iic_bn_finalize")
  return coef


def bn_apply(y, coef, out, N, H, W, P, C, res=None, y2=None, coef2=None, relu=True):
  if y.dtype == F32:
    check(lib().iic_f32_bn_apply(ptr(y), ptr(coef), ptr(res), ptr(y2), ptr(coef2), ptr(out), N, H, W, P, C,
                                 1 if relu else 0, stream_ptr()), "iic_f32_bn_apply")
    return out
  check(lib().iic_bn_apply(ptr(y), ptr(coef), ptr(res), ptr(y2), ptr(coef2), ptr(out), N, H, W, P,
                           C, 1 if relu else 0, stream_ptr()), "iic_bn_apply")
  return out


def bn_bwd_reduce(dout, act, y, sums, N, H, W, P, C, y2=None, sums2=None, mask_coef=None):
  """mask_coef: forward coef of this BN when act = relu(bn(y)) exactly (pass act=None): the ReLU
  mask is recomputed from y instead of reading the activation tensor."""
  if y.dtype == F32:
    check(lib().iic_f32_bn_bwd_reduce(ptr(dout), ptr(act), ptr(y), ptr(y2), ptr(sums), ptr(sums2),
                                      ptr(mask_coef), N, H, W, P, C, stream_ptr()), "iic_f32_bn_bwd_reduce")
    return
  check(lib().iic_bn_bwd_reduce(ptr(dout), ptr(act), ptr(y), ptr(y2), ptr(sums), ptr(sums2),
                                ptr(mask_coef), N, H, W, P, C, stream_ptr()), "iic_bn_bwd_reduce")


def bn_bwd_finalize(sums, gamma, coef, C, count):
  bcoef = torch.empty((3, C), dtype=F32, device=gamma.device)
  dgamma = torch.empty(C, dtype=F32, device=gamma.device)
  dbeta = torch.empty(C, dtype=F32, device=gamma.device)
  check(lib().iic_bn_bwd_finalize(ptr(sums), ptr(gamma), ptr(coef), ptr(bcoef), ptr(dgamma),
                                  ptr(dbeta), C, count, stream_ptr()), "iic_bn_bwd_finalize")
  return bcoef, dgamma, dbeta


def bn_bwd_apply(dout, act, y, bcoef, dy, N, H, W, P, C, y2=None, bcoef2=None, dy2=None,
                 mask_coef=None):
  if y.dtype == F32:
    check(lib().iic_f32_bn_bwd_apply(ptr(dout), ptr(act), ptr(y), ptr(bcoef), ptr(dy), ptr(y2), ptr(bcoef2),
                                     ptr(dy2), ptr(mask_coef), N, H, W, P, C, stream_ptr()),
          "iic_f32_bn_bwd_apply")
    return
  check(lib().iic_bn_bwd_apply(ptr(dout), ptr(act), ptr(y), ptr(bcoef), ptr(dy), ptr(y2),
                               ptr(bcoef2), ptr(dy2), ptr(mask_coef), N, H, W, P, C, stream_ptr()),
        "iic_bn_bwd_apply")


# ------------------------------------------------------------------------------------
# stem + sobel
# ------------------------------------------------------------------------------------
def sobel(imgs, include_rgb, using_IR=False):
  n, c, h, w = imgs.shape
  cout = {(False, False): 2, (True, False): 5, (False, True): 3, (True, True): 6}[
    (bool(include_rgb), bool(using_IR))]
  imgs = imgs.contiguous()
  out = torch.empty((n, cout, h, w), dtype=F32, device=imgs.device)
  check(lib().iic_sobel(ptr(imgs), ptr(out), n, c, h, w, 1 if include_rgb else 0,
                        1 if using_IR else 0, stream_ptr()), "iic_sobel")
  return out


def stem_stats(x, w, stats):
  n, c, h, wd = x.shape
  check(lib().iic_stem_stats(ptr(x), ptr(w), ptr(stats), n, c, h, wd, stream_ptr()), "iic_stem_stats")


def stem_apply_pool(x, w, coef, out_pt):
  n, c, h, wd = x.shape
  check(lib().iic_stem_apply_pool(ptr(x), ptr(w), ptr(coef), ptr(out_pt), n, c, h, wd, stream_ptr()),
        "iic_stem_apply_pool")


def stem_bwd_reduce(x, w, coef, dpool, sums):
  n, c, h, wd = x.shape
  check(lib().iic_stem_bwd_reduce(ptr(x), ptr(w), ptr(coef), ptr(dpool), ptr(sums), n, c, h, wd,
                                  stream_ptr()), "iic_stem_bwd_reduce")


_STEM_PART = {}


def stem_bwd_wgrad(x, w, coef, bcoef, dpool):
  n, c, h, wd = x.shape
  part = _stem_partials(x.device)
  dW = torch.empty_like(w)
  check(lib().iic_stem_bwd_wgrad(ptr(x), ptr(w), ptr(coef), ptr(bcoef), ptr(dpool), ptr(part),
                                 ptr(dW), n, c, h, wd, stream_ptr()), "iic_stem_bwd_wgrad")
  return dW


def _stem_partials(device):
  key = (str(device), BRANCH[0])
  part = _STEM_PART.get(key)
  if part is None:
    part = torch.empty(lib().iic_stem_wgrad_partial_floats(), dtype=F32, device=device)
    _STEM_PART[key] = part
  return part


def stem_bwd_fused_ok(cin):
  """One-pass stem backward: K = 9*Cin must fit one 32-wide MFMA column tile."""
  return 9 * cin <= 32


def stem_bwd_fused(x, w, coef, dpool, sums):
  """sums += (sum g, sum g*y) AND the coefficient-free dW GEMMs; returns the handle for
  stem_wgrad_combine."""
  n, c, h, wd = x.shape
  part = _stem_partials(x.device)
  nb = ctypes.c_int(0)
  check(lib().iic_stem_bwd_fused(ptr(x), ptr(w), ptr(coef), ptr(dpool), ptr(sums), ptr(part),
                                 ctypes.byref(nb), n, c, h, wd, stream_ptr()), "iic_stem_bwd_fused")
  return part, nb.value


def stem_wgrad_combine(handle, bcoef, w):
  part, nb = handle
  dW = torch.empty_like(w)
  check(lib().iic_stem_wgrad_combine(ptr(part), nb, ptr(bcoef), ptr(dW), w.shape[1], stream_ptr()),
        "iic_stem_wgrad_combine")
  return dW


def f32_nchw_to_pt(x, out_pt, P):
  n, c, h, w = x.shape
  check(lib().iic_f32_nchw_to_pt(ptr(x), ptr(out_pt), n, c, h, w, P, stream_ptr()), "iic_f32_nchw_to_pt")
  return out_pt


def f32_maxpool_s2p1_fwd(x_pt, out_pt, N, H, W, C):
  check(lib().iic_f32_maxpool_s2p1_fwd(ptr(x_pt), ptr(out_pt), N, H, W, C, stream_ptr()), "iic_f32_maxpool_s2p1_fwd")
  return out_pt


def f32_maxpool_s2p1_bwd(x_pt, dout_pt, din_pt, N, H, W, C):
  check(lib().iic_f32_maxpool_s2p1_bwd(ptr(x_pt), ptr(dout_pt), ptr(din_pt), N, H, W, C, stream_ptr()),
        "iic_f32_maxpool_s2p1_bwd")
  return din_pt


# ------------------------------------------------------------------------------------
# heads
# ------------------------------------------------------------------------------------
def avgpool_fwd(x_pt, N, H, W, P, C):
  feats = torch.empty((N, C), dtype=F32, device=x_pt.device)
  if x_pt.dtype == F32:
    check(lib().iic_f32_avgpool_fwd(ptr(x_pt), ptr(feats), N, H, W, P, C, stream_ptr()), "iic_f32_avgpool_fwd")
    return feats
  check(lib().iic_avgpool_fwd(ptr(x_pt), ptr(feats), N, H, W, P, C, stream_ptr()), "iic_avgpool_fwd")
  return feats


def avgpool_bwd(dfeats, out_pt, N, H, W, P, C, mask_act=None):
  if out_pt.dtype == F32:
    check(lib().iic_f32_avgpool_bwd(ptr(dfeats), ptr(out_pt), N, H, W, P, C, ptr(mask_act), stream_ptr()),
          "iic_f32_avgpool_bwd")
    return out_pt
  check(lib().iic_avgpool_bwd(ptr(dfeats), ptr(out_pt), N, H, W, P, C, ptr(mask_act), stream_ptr()),
        "iic_avgpool_bwd")
  return out_pt


_GEMM_WS = {}      # (branch, device) -> fp32 workspace of the K-split GEMMs (grown on demand, reused)
_GEMM_WS_RETIRED = []   # superseded workspaces: a captured graph may have their address baked in -- never freed


def gemm_f32(A, sam, sak, B, sbk, sbn, C, scm, M, N, K, bias=None, accumulate=False):
  need = lib().iic_gemm_f32_ws_floats(sam, sak, sbk, sbn, M, N, K)
  ws = None
  if need:
    key = (BRANCH[0], C.device.index)
    ws = _GEMM_WS.get(key)
    if ws is None or ws.numel() < need:
      if ws is not None:
        _GEMM_WS_RETIRED.append(ws)        # (head B's graphs kept replaying into a buffer that head A's
                                           #  larger request had freed)
      ws = _GEMM_WS[key] = torch.empty(max(need, 1 << 20), dtype=F32, device=C.device)
  check(lib().iic_gemm_f32_ws(ptr(A), sam, sak, ptr(B), sbk, sbn, ptr(bias), ptr(C), scm, M, N, K,
                              1 if accumulate else 0, ptr(ws), need, stream_ptr()), "iic_gemm_f32_ws")
  return C


def softmax_fwd(logits, rows, k):
  probs = torch.empty_like(logits)
  check(lib().iic_softmax_fwd(ptr(logits), ptr(probs), rows, k, stream_ptr()), "iic_softmax_fwd")
  return probs


def softmax_bwd(probs, dprobs, rows, k):
  dl = torch.empty_like(probs)
  check(lib().iic_softmax_bwd(ptr(probs), ptr(dprobs), ptr(dl), rows, k, stream_ptr()),
        "iic_softmax_bwd")
  return dl


def colsum(A, rows, cols):
  out = torch.empty(cols, dtype=F32, device=A.device)
  check(lib().iic_colsum_f32(ptr(A), ptr(out), rows, cols, 0, stream_ptr()), "iic_colsum_f32")
  return out


# ------------------------------------------------------------------------------------
# VGG-style trunks: first-layer conv from the image, 2x2 max-pool
# ------------------------------------------------------------------------------------
def firstconv_fwd(x, w, out_pt, stats, K, pad, P):
  n, c, h, wd = x.shape
  check(lib().iic_firstconv_fwd(ptr(x), ptr(w), ptr(out_pt), ptr(stats), n, c, h, wd, K, pad, P,
                                stream_ptr()), "iic_firstconv_fwd")
  return out_pt


_FC_PART = {}


def firstconv_wgrad(x, dy_pt, w_shape, K, pad, P):
  n, c, h, wd = x.shape
  key = (str(x.device), BRANCH[0])
  part = _FC_PART.get(key)
  if part is None:
    part = torch.empty(lib().iic_firstconv_wgrad_partial_floats(), dtype=F32, device=x.device)
    _FC_PART[key] = part
  dW = torch.empty(w_shape, dtype=F32, device=x.device)
  check(lib().iic_firstconv_wgrad(ptr(x), ptr(dy_pt), ptr(part), ptr(dW), n, c, h, wd, K, pad, P,
                                  stream_ptr()), "iic_firstconv_wgrad")
  return dW


def maxpool2_fwd(x_pt, out_pt, N, H, W, Pi, Po, C):
  if x_pt.dtype == F32:
    check(lib().iic_f32_maxpool2_fwd(ptr(x_pt), ptr(out_pt), N, H, W, Pi, Po, C, stream_ptr()), "iic_f32_maxpool2_fwd")
    return out_pt
  check(lib().iic_maxpool2_fwd(ptr(x_pt), ptr(out_pt), N, H, W, Pi, Po, C, stream_ptr()),
        "iic_maxpool2_fwd")
  return out_pt


def bn_relu_maxpool2_fwd(y_pt, coef, out_pt, N, H, W, Pi, Po, C):
  """out = maxpool2(relu(bn(y))) without storing the activation (bf16 PT tensors only; include/iic_hip.h)."""
  assert y_pt.dtype == BF16
  check(lib().iic_bn_relu_maxpool2_fwd(ptr(y_pt), ptr(coef), ptr(out_pt), N, H, W, Pi, Po, C, stream_ptr()),
        "iic_bn_relu_maxpool2_fwd")
  return out_pt


def bn_relu_maxpool2_bwd(y_pt, coef, dout_pt, din_pt, N, H, W, Pi, Po, C):
  """din = gradient w.r.t. a = relu(bn(y)) of that pool: dout at the first arg-max of the recomputed a."""
  assert y_pt.dtype == BF16
  check(lib().iic_bn_relu_maxpool2_bwd(ptr(y_pt), ptr(coef), ptr(dout_pt), ptr(din_pt), N, H, W, Pi, Po, C,
                                       stream_ptr()), "iic_bn_relu_maxpool2_bwd")
  return din_pt


def maxpool2_bwd(x_pt, dout_pt, din_pt, N, H, W, Pi, Po, C):
  if x_pt.dtype == F32:
    check(lib().iic_f32_maxpool2_bwd(ptr(x_pt), ptr(dout_pt), ptr(din_pt), N, H, W, Pi, Po, C, stream_ptr()),
          "iic_f32_maxpool2_bwd")
    return din_pt
  check(lib().iic_maxpool2_bwd(ptr(x_pt), ptr(dout_pt), ptr(din_pt), N, H, W, Pi, Po, C,
                               stream_ptr()), "iic_maxpool2_bwd")
  return din_pt
