"""Fused multi-tensor Adam on the HIP kernel (iic_amd/csrc/optim.hip).

Drop-in for ``torch.optim.Adam(params, lr=...)`` as the reference uses it
(/root/reference/code/utils/cluster/general.py:5-9, cluster_sobel.py:149,272): default
betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad.  ``param_groups[i]['lr']`` can be
scaled in place exactly like update_lr() does (general.py:20-23).

``state[p]['step']`` is a Python int (torch.optim.Adam checkpoints, which store it as a tensor,
are normalised on load and on first use).  ``capturable=True`` keeps the step count in device
memory instead (one int32 counter shared by all tensors that have always been updated together)
so that the launch arguments never change and ``step()`` can be captured in a HIP graph
(iic_amd.graph.CapturedStep); ``state[p]['step']`` then reads that counter.
"""
import ctypes
import weakref

import torch

from . import ops
from ._lib import check, lib, stream_ptr
from .archs.cluster import bump_weights_epoch


_CAPTURABLE = weakref.WeakSet()      # live capturable optimisers (their learning rates live on the device)


def sync_captured_lr():
  """Push param_groups[i]["lr"] of every capturable optimiser to its device copy where it changed since the
  last push: called by iic_amd.graph.Captured*Step before every replay, so that update_lr()
  (/root/reference/code/utils/cluster/general.py:12-23: an in-place edit of param_groups) takes effect on the
  next replayed step although no host code of step() runs at replay."""
  for opt in list(_CAPTURABLE):
    opt.sync_lr()


def _as_int(step):
  if torch.is_tensor(step):
    return int(round(float(step)))
  return int(step)


class Adam(torch.optim.Optimizer):
  def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
               capturable=False):
    if weight_decay != 0 or amsgrad:
      raise NotImplementedError("iic_amd.optim.Adam: weight_decay / amsgrad are not used by the "
                                "reference (general.py:5-9) and are not implemented")
    super(Adam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps))
    self.capturable = bool(capturable)
    self._members = {}     # id(counter tensor) -> (counter, set of param ids sharing it)
    if self.capturable:
      _CAPTURABLE.add(self)

  def _lr_dev(self, group, device):
    """Device copy of the group's learning rate (capturable mode), created at the first step -- i.e. in the
    eager warm-up steps, before any capture."""
    t = group.get("_lr_dev")
    if t is None or t.device != device:
      t = group["_lr_dev"] = torch.full((1,), float(group["lr"]), dtype=torch.float32, device=device)
      group["_lr_pushed"] = float(group["lr"])
    return t

  def sync_lr(self):
    for group in self.param_groups:
      t = group.get("_lr_dev")
      if t is not None and group.get("_lr_pushed") != float(group["lr"]):
        t.fill_(float(group["lr"]))
        group["_lr_pushed"] = float(group["lr"])

  # ---- checkpoint compatibility with torch.optim.Adam (step stored as a tensor there) ----
  def load_state_dict(self, state_dict):
    # captured graphs read the learning rate through the EXISTING device tensors: keep them across the
    # replacement of param_groups and refill them with the loaded rates
    lr_devs = [g.get("_lr_dev") for g in self.param_groups]
    super(Adam, self).load_state_dict(state_dict)
    for g, t in zip(self.param_groups, lr_devs):
      if t is not None:
        t.fill_(float(g["lr"]))
        g["_lr_dev"], g["_lr_pushed"] = t, float(g["lr"])
    self._members = {}
    for st in self.state.values():
      if "step" in st:
        st["step"] = _as_int(st["step"])
      st.pop("_counter", None)

  def state_dict(self):
    if self.capturable:
      self._sync_steps_to_host()
    sd = super(Adam, self).state_dict()
    # (the per-parameter dicts are the live ones: copy before dropping the private entry)
    sd["state"] = {k: {n: v for n, v in st.items() if n != "_counter"}
                   for k, st in sd["state"].items()}
    sd["param_groups"] = [{n: v for n, v in g.items() if not n.startswith("_lr_")} for g in sd["param_groups"]]
    return sd

  def _sync_steps_to_host(self):
    for st in self.state.values():
      c = st.get("_counter")
      if c is not None:
        st["step"] = int(c.item())

  def _init_state(self, p):
    st = self.state[p]
    if "exp_avg" not in st:
      st["step"] = 0
      st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
      st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
    elif not isinstance(st["step"], int):
      st["step"] = _as_int(st["step"])
    assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
    if p.grad is not None and not p.grad.is_contiguous():
      p.grad = p.grad.contiguous()
    return st

  @staticmethod
  def _branch_grad(p):
    """Gradient a side branch (iic_amd.ops.branch) accumulated for p through its leaf alias."""
    gs = ops.branch_grads(p)
    if not gs:
      return None
    assert len(gs) == 1, "one side branch per step"
    return gs[0] if gs[0].is_contiguous() else gs[0].contiguous()

  @staticmethod
  def _tables(grp, state):
    n = len(grp)
    VP = ctypes.c_void_p * n
    LP = ctypes.c_long * n
    g2 = [Adam._branch_grad(p) for p in grp]
    keep = [t for t in g2 if t is not None]       # contiguous copies must outlive the launch call
    return keep, (n, VP(*[p.data_ptr() for p in grp]),
                  VP(*[p.grad.data_ptr() if p.grad is not None else None for p in grp]),
                  VP(*[t.data_ptr() if t is not None else None for t in g2]) if keep else None,
                  VP(*[state[p]["exp_avg"].data_ptr() for p in grp]),
                  VP(*[state[p]["exp_avg_sq"].data_ptr() for p in grp]),
                  LP(*[p.numel() for p in grp]))

  def _counter_groups(self, ps):
    """Capturable mode: partition `ps` into groups that share one device step counter.  A
    counter is split (host work, first occurrence of a new update pattern only) when just a
    part of its tensors receives a gradient -- e.g. head A / head B of a two-head net."""
    by = {}
    for p in ps:
      st = self.state[p]
      c = st.get("_counter")
      by.setdefault(("dev", id(c)) if c is not None else ("host", st["step"]), []).append(p)
    groups = []
    for key, grp in by.items():
      if key[0] == "host":
        c = torch.full((1,), key[1], dtype=torch.int32, device=grp[0].device)
        self._members[id(c)] = (c, set(id(p) for p in grp))
      else:
        c, members = self._members[key[1]]
        ids = set(id(p) for p in grp)
        if ids != members:
          members -= ids
          c = c.clone()
          self._members[id(c)] = (c, ids)
      for p in grp:
        self.state[p]["_counter"] = c
      groups.append((c, grp))
    return groups

  def zero_grad(self, set_to_none=True):
    """Also drops the gradients the side branches accumulated through the parameters' leaf aliases
    (iic_amd.ops.branch): a branched step followed by an un-branched one must not see them again."""
    super(Adam, self).zero_grad(set_to_none=set_to_none)
    ops.clear_branch_grads()

  @torch.no_grad()
  def step(self, closure=None):
    loss = None
    if closure is not None:
      with torch.enable_grad():
        loss = closure()
    # a forward forked by ops.branch / ops.auto_branch that no loss of this library consumed (e.g. a
    # semi-supervised head with cross-entropy) is joined here at the latest: the side stream's
    # gradients must have landed, and its postponed running-statistic updates are applied
    ops.join()
    for group in self.param_groups:
      ps = [p for p in group["params"] if p.grad is not None or ops.branch_grads(p)]
      if not ps:
        continue
      for p in ps:
        self._init_state(p)
      hyper = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
               float(group["eps"]))
      if self.capturable:
        lr_dev = self._lr_dev(group, ps[0].device)
        if not torch.cuda.is_current_stream_capturing():
          self.sync_lr()         # (recorded into a graph, the fill would reset the rate at every replay: ADVICE r3)
        for counter, grp in self._counter_groups(ps):
          keep, tab = self._tables(grp, self.state)
          check(lib().iic_adam_step_devlr(*tab, lr_dev.data_ptr(), *hyper, counter.data_ptr(), stream_ptr()),
                "iic_adam_step_devlr")
          del keep
        continue
      # torch.optim.Adam keeps a step count PER PARAMETER: a two-head net only produces
      # gradients for the head that was used (net5g_two_head.py:62-81), so head-A and head-B
      # parameters advance at different rates.  One fused launch per distinct step count.
      by_step = {}
      for p in ps:
        by_step.setdefault(self.state[p]["step"], []).append(p)
      for step0, grp in by_step.items():
        step = step0 + 1
        keep, tab = self._tables(grp, self.state)
        check(lib().iic_adam_step(*tab, *hyper, step, stream_ptr()), "iic_adam_step")
        del keep
        for p in grp:
          self.state[p]["step"] = step
    bump_weights_epoch()
    return loss
