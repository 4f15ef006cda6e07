"""Fused multi-tensor Adam on the HIP kernel (iic_amd/csrc/optim.hip).

Drop-in for ``torch.optim.Adam(params, lr=...)`` as the reference uses it
(/root/reference/code/utils/cluster/general.py:5-9, cluster_sobel.py:149,272): default
betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad.  ``param_groups[i]['lr']`` can be
scaled in place exactly like update_lr() does (general.py:20-23).
"""
import ctypes

import torch

from ._lib import check, lib, stream_ptr
from .archs.cluster import bump_weights_epoch


class Adam(torch.optim.Optimizer):
  def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    super(Adam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps))

  @torch.no_grad()
  def step(self, closure=None):
    loss = None
    if closure is not None:
      with torch.enable_grad():
        loss = closure()
    for group in self.param_groups:
      ps = [p for p in group["params"] if p.grad is not None]
      if not ps:
        continue
      for p in ps:
        st = self.state[p]
        if not st:
          st["step"] = 0
          st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
          st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
        if not p.grad.is_contiguous():
          p.grad = p.grad.contiguous()
      # torch.optim.Adam keeps a step count PER PARAMETER: a two-head net only produces
      # gradients for the head that was used (net5g_two_head.py:62-81), so head-A and head-B
      # parameters advance at different rates.  One fused launch per distinct step count.
      by_step = {}
      for p in ps:
        by_step.setdefault(self.state[p]["step"], []).append(p)
      for step0, grp in by_step.items():
        step = step0 + 1
        n = len(grp)
        VP = ctypes.c_void_p * n
        LP = ctypes.c_long * n
        check(lib().iic_adam_step(
          n, VP(*[p.data_ptr() for p in grp]), VP(*[p.grad.data_ptr() for p in grp]),
          VP(*[self.state[p]["exp_avg"].data_ptr() for p in grp]),
          VP(*[self.state[p]["exp_avg_sq"].data_ptr() for p in grp]),
          LP(*[p.numel() for p in grp]), float(group["lr"]), float(group["betas"][0]),
          float(group["betas"][1]), float(group["eps"]), step, stream_ptr()), "iic_adam_step")
        for p in grp:
          self.state[p]["step"] = step
    bump_weights_epoch()
    return loss
