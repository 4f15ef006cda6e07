"""MI355X-native ClusterNet6c / ClusterNet6cTwoHead -- drop-in for
/root/reference/code/archs/cluster/{net6c,net6c_two_head,vgg}.py.

``trunk.features`` is an nn.Sequential with the reference's layout (Conv2d, BatchNorm2d,
ReLU triples and MaxPool2d at the same indices => identical state_dict keys); the modules
are parameter holders only.  The computation is one autograd Function per conv "stage"
(conv -> BN(batch stats) -> ReLU [-> MaxPool 2x2]) on the HIP kernels:
  first stage  : fp32-MFMA conv from the NCHW image            (csrc/vgg.hip)
  other stages : bf16-MFMA implicit GEMM, 25 taps for 5x5      (csrc/conv_igemm.hip, conv_wgrad.hip)
  BN / ReLU    : csrc/bn.hip;  MaxPool: csrc/vgg.hip;  heads: csrc/head.hip
Activations are PT tensors with border P = 2 (the 5x5 convs' padding).
"""
import torch
import torch.nn as nn

from .. import geom as G
from .. import ops
from ..dist import shard_batch
from .cluster import FUSE_RED, _ApplyCounter, _ConvHolder, _HeadsFn, _bn_buffers, _bn_training

import os

# IIC_FUSE_POOL=0: store the pooled stages' activation and pool it in a separate pass (cross-check of the fused kernels)
FUSE_POOL = [os.environ.get("IIC_FUSE_POOL", "1") != "0"]

__all__ = ["ClusterNet6c", "ClusterNet6cTwoHead"]


class _StageFn(torch.autograd.Function):
  """conv + BN + ReLU (+ maxpool).  x: NCHW fp32 image (first stage) or PT bf16."""

  @staticmethod
  def forward(ctx, x, w, gamma, beta, st, prev=None, link=None):
    """prev / link: _Link objects of the trunk's forward (run_stages) -- `prev` describes the stage
    whose output is this stage's input, `link` is filled in for the next stage."""
    P = st.P
    dev = x.device
    bn = st.bn
    training = _bn_training(bn)
    rm, rv, nbt = _bn_buffers(bn)
    C = st.cout
    if st.first:
      assert x.is_cuda, "HIP path needs a device tensor -- no CPU fallback"
      x = x.contiguous().float()
      N, _, H, W = x.shape
      Ho, Wo = H, W
      y = ops.pt_alloc(N, Ho, Wo, C, P, dev)
      stt = st.holder.stats(dev) if training else None
      if ops.PT_DTYPE[0] is torch.float32:      # exact-fp32 parity path: generic conv on the PT image
        x = ops.f32_nchw_to_pt(x, ops.pt_alloc(N, H, W, x.shape[1], P, dev), P)
        gf, _ = st.holder.geoms(N, H, W)
        ops.conv_igemm(gf, x, st.holder.weights()[0], y, stats=stt)
      else:
        ops.firstconv_fwd(x, w.detach(), y, stt, st.K, st.pad, P)
    else:
      N, Hp, Wp, _ = x.shape
      H, W = Hp - 2 * P, Wp - 2 * P
      Ho, Wo = st.holder.spec.out_size(H), st.holder.spec.out_size(W)
      gf, _ = st.holder.geoms(N, H, W)
      y = ops.pt_alloc(N, Ho, Wo, C, P, dev)
      stt = st.holder.stats(dev) if training else None
      ops.conv_igemm(gf, x, st.holder.weights()[0], y, stats=stt)
    cnt = N * Ho * Wo
    if training:
      upd = bn.training
      coef = ops.bn_finalize(stt, gamma.detach(), beta.detach(), rm if upd else None,
                             rv if upd else None, nbt if upd else None, C, cnt, True)
    else:
      coef = ops.bn_finalize(None, gamma.detach(), beta.detach(), rm, rv, None, C, cnt, False)
    # pooled stages (bf16 path): the activation a = relu(bn(y)) is never stored -- the pool recomputes it from (y, coef)
    # bit for bit, forward and backward (csrc/vgg.hip maxpool2_*_kernel<true>): one full-tensor write + read less
    fused_pool = st.pool and FUSE_POOL[0] and ops.PT_DTYPE[0] is not torch.float32
    a = None
    if fused_pool:
      out = ops.pt_alloc(N, Ho // 2, Wo // 2, C, P, dev)
      ops.bn_relu_maxpool2_fwd(y, coef, out, N, Ho, Wo, P, P, C)
    else:
      a = ops.pt_alloc(N, Ho, Wo, C, P, dev)
      ops.bn_apply(y, coef, a, N, Ho, Wo, P, C, relu=True)
      out = a
      if st.pool:
        out = ops.pt_alloc(N, Ho // 2, Wo // 2, C, P, dev)
        ops.maxpool2_fwd(a, out, N, Ho, Wo, P, P, C)
    need_grad = any(ctx.needs_input_grad)
    if need_grad:
      ctx.st = st
      ctx.branch, ctx.pt_dtype = ops.BRANCH[0], ops.PT_DTYPE[0]
      ctx.dims = (N, H, W, Ho, Wo)
      ctx.training = training
      ctx.dout_prereduced = False
      # fused BatchNorm-backward reduction (cluster.FUSE_RED): this stage's backward-data conv produces
      # the output gradient of the PREVIOUS stage; when that stage has no pool in between, the conv
      # takes the previous BatchNorm's two sums in its epilogue and the previous stage skips its
      # reduction pass
      ctx.red_prev = None
      if (FUSE_RED[0] and prev is not None and prev.ctx is not None and training and not st.first
          and ops.PT_DTYPE[0] is not torch.float32):
        ctx.red_prev = prev
      if link is not None and training and not st.pool:
        link.ctx, link.y, link.coef, link.st = ctx, y, coef, st
      ctx.save_for_backward(x, w, gamma, y, a, coef, out if st.pool else None)
    else:
      ops.POOL.release(y)
      if st.pool and a is not None:
        ops.POOL.release(a)
    return out

  @ops.branch_backward
  def backward(ctx, dout):
    x, w, gamma, y, a, coef, pooled = ctx.saved_tensors
    st = ctx.st
    if not ctx.training:
      raise RuntimeError("HIP BatchNorm backward is implemented for batch statistics only")
    N, H, W, Ho, Wo = ctx.dims
    P, C, dev = st.P, st.cout, y.device
    dout = dout.contiguous()
    cnt = N * Ho * Wo
    if st.pool:
      da = ops.pt_alloc(N, Ho, Wo, C, P, dev)
      if a is None:          # fused pool: the arg-max is taken on the activation recomputed from (y, coef)
        ops.bn_relu_maxpool2_bwd(y, coef, dout, da, N, Ho, Wo, P, P, C)
      else:
        ops.maxpool2_bwd(a, dout, da, N, Ho, Wo, P, P, C)
      ops.POOL.release(dout)
      ops.POOL.release(pooled)
    else:
      da = dout
    sums = st.holder.stats(dev, "bwd")
    # a = relu(bn(y)) exactly: the ReLU mask is recomputed from (y, coef), a is not read here
    if not ctx.dout_prereduced:      # else: the next stage's backward-data conv took these sums
      ops.bn_bwd_reduce(da, None, y, sums, N, Ho, Wo, P, C, mask_coef=coef)
    bcoef, dgamma, dbeta = ops.bn_bwd_finalize(sums, gamma.detach(), coef, C, cnt)
    dy = ops.pt_alloc(N, Ho, Wo, C, P, dev)
    ops.bn_bwd_apply(da, None, y, bcoef, dy, N, Ho, Wo, P, C, mask_coef=coef)
    dx = None
    if st.first and x.dim() == 4 and x.dtype == torch.float32 and dy.dtype == torch.float32:
      gf, _ = st.holder.geoms(N, H, W)         # fp32 parity path: x is the PT copy of the image
      dW = ops.conv_wgrad(gf, x, dy, st.K * st.K).view(C, st.cin, st.K, st.K)
      ops.POOL.release(x)
    elif st.first:
      dW = ops.firstconv_wgrad(x, dy, tuple(w.shape), st.K, st.pad, P)
    else:
      gf, gb = st.holder.geoms(N, H, W)
      dx = ops.pt_alloc(N, H, W, st.cin, P, dev)
      w_t = st.holder.weights()[1]
      red, pl = None, ctx.red_prev
      if pl is not None and len(gb) == 1 and ops.red_supported(gb[0], w_t):
        red = (pl.y, pl.coef, pl.st.holder.stats(dev, "bwd"), None, None)
        pl.ctx.dout_prereduced = True
      for g in gb:
        ops.conv_igemm(g, dy, w_t, dx, red=red)
      dW = ops.conv_wgrad(gf, x, dy, st.K * st.K, True).view(C, st.cin, st.K, st.K)
    for t in (da, dy, y, a):
      ops.POOL.release(t)
    return dx, dW, dgamma, dbeta, None, None, None


class _Link(object):
  """Per trunk forward: what the next stage needs to know about the stage before it."""
  __slots__ = ("ctx", "y", "coef", "st")

  def __init__(self):
    self.ctx = self.y = self.coef = self.st = None


class _Stage(object):
  def __init__(self, conv, bn, pool, first, P):
    self.conv, self.bn, self.pool, self.first, self.P = conv, bn, pool, first, P
    self.cin, self.cout = conv.in_channels, conv.out_channels
    self.K, self.pad = conv.kernel_size[0], conv.padding[0]
    self.holder = _ConvHolder(conv, pad_in=P, pad_out=P)


class _FlattenFn(torch.autograd.Function):
  """PT [N, h+2P, w+2P, C] interior -> [N, C*h*w] fp32 in the reference's (c, h, w) flatten order (net6c.py:24-25):
  one strided copy each way, and the heads use their weights as they are (the (h, w, c) order of rounds 1-4 cost a
  permuted copy of every sub-head's weight per forward and its transpose back per backward: ~20 small launches per
  optimiser step of a launch-bound net)."""

  @staticmethod
  def forward(ctx, x, P):
    N, Hp, Wp, C = x.shape
    ctx.meta = (tuple(x.shape), P)
    ctx.branch, ctx.pt_dtype = ops.BRANCH[0], x.dtype
    out = torch.empty((N, C, Hp - 2 * P, Wp - 2 * P), dtype=torch.float32, device=x.device)
    out.copy_(x[:, P:Hp - P, P:Wp - P, :].permute(0, 3, 1, 2))
    return out.view(N, -1)

  @ops.branch_backward
  def backward(ctx, dfeat):
    shape, P = ctx.meta
    N, Hp, Wp, C = shape
    dx = ops.POOL.alloc(shape, dfeat.device, P)
    dx[:, P:Hp - P, P:Wp - P, :].copy_(dfeat.contiguous().view(N, C, Hp - 2 * P, Wp - 2 * P).permute(0, 2, 3, 1))
    return dx, None


class VGGTrunkHIP(nn.Module):
  """vgg.py:8-35 (_make_layers) with holders; forward runs the stage Functions."""
  P = 2

  def _make_layers(self, cfg, in_channels, conv_size, pad):
    layers = []
    cin = in_channels
    for out, dilation in cfg:
      if out == "M":
        layers += [nn.MaxPool2d(kernel_size=2, stride=2)]
      elif out == "A":
        raise NotImplementedError("AvgPool stages are not used by the hot-path architectures")
      else:
        layers += [nn.Conv2d(cin, out, kernel_size=conv_size, stride=1, padding=pad,
                             dilation=dilation, bias=False),
                   nn.BatchNorm2d(out, track_running_stats=self.batchnorm_track),
                   nn.ReLU(inplace=True)]
        cin = out
    return nn.Sequential(*layers)

  def _build_stages(self):
    mods = list(self.features)
    stages, i, first = [], 0, True
    while i < len(mods):
      m = mods[i]
      if isinstance(m, nn.Conv2d):
        pool = (i + 3 < len(mods)) and isinstance(mods[i + 3], nn.MaxPool2d)
        stages.append(_Stage(m, mods[i + 1], pool, first, self.P))
        first = False
        i += 4 if pool else 3
      else:
        raise AssertionError("unexpected layer order in VGG features")
    self._stages = stages

  def run_stages(self, x):
    x = shard_batch(x, self)          # unchanged scripts under torchrun: this rank's pairs only
    prev = None
    for st in self._stages:
      link = _Link()
      x = _StageFn.apply(x, ops.pv(st.conv.weight), ops.pv(st.bn.weight), ops.pv(st.bn.bias), st, prev, link)
      prev = link
    return x


class ClusterNet6cTrunk(VGGTrunkHIP):
  def __init__(self, config):
    super(ClusterNet6cTrunk, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.conv_size, self.pad = 5, 2
    self.cfg = ClusterNet6c.cfg
    self.in_channels = config.in_channels if hasattr(config, "in_channels") else 3
    self.features = self._make_layers(self.cfg, self.in_channels, self.conv_size, self.pad)
    self._build_stages()

  def forward(self, x):
    x = self.run_stages(x)
    return _FlattenFn.apply(x, self.P)   # the reference's (c, h, w) order


class ClusterNet6cHead(nn.Module):
  def __init__(self, config, output_k=None):
    super(ClusterNet6cHead, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.num_sub_heads = config.num_sub_heads
    self.output_k = config.output_k if output_k is None else output_k
    self.num_features = ClusterNet6c.cfg[-1][0]
    if config.input_sz == 24:
      self.sp = 3
    elif config.input_sz == 64:
      self.sp = 8
    else:
      raise ValueError("ClusterNet6c supports input_sz 24 or 64 (net6c.py:42-45)")
    self.heads = nn.ModuleList([nn.Sequential(
      nn.Linear(self.num_features * self.sp * self.sp, self.output_k),
      nn.Softmax(dim=1)) for _ in range(self.num_sub_heads)])

  def forward_packed(self, feats):
    k = self.output_k
    Wcat = torch.cat([ops.pv(h[0].weight) for h in self.heads], dim=0)
    bcat = torch.cat([ops.pv(h[0].bias) for h in self.heads], dim=0)
    return _HeadsFn.apply(feats, Wcat, bcat, self.num_sub_heads, k)

  def forward(self, x, kmeans_use_features=False):
    if kmeans_use_features:
      return [x for _ in range(self.num_sub_heads)]
    probs = self.forward_packed(x)
    return ops.tag_pack([probs[:, i, :] for i in range(self.num_sub_heads)])


def _initialize_weights_vgg(net, mode="fan_in"):
  """vgg.py:42-54."""
  for m in net.modules():
    if isinstance(m, nn.Conv2d):
      nn.init.kaiming_normal_(m.weight, mode=mode, nonlinearity="relu")
    elif isinstance(m, nn.BatchNorm2d):
      assert m.track_running_stats == net.batchnorm_track
      m.weight.data.fill_(1)
      m.bias.data.zero_()
    elif isinstance(m, nn.Linear):
      m.weight.data.normal_(0, 0.01)
      m.bias.data.zero_()


class ClusterNet6c(_ApplyCounter, nn.Module):
  cfg = [(64, 1), ("M", None), (128, 1), ("M", None), (256, 1), ("M", None), (512, 1)]

  def __init__(self, config):
    super(ClusterNet6c, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.trunk = ClusterNet6cTrunk(config)
    self.head = ClusterNet6cHead(config)
    _initialize_weights_vgg(self)

  def forward_packed(self, x):
    return self.head.forward_packed(self.trunk(x))

  @ops.auto_branch
  def forward(self, x, kmeans_use_features=False, trunk_features=False, penultimate_features=False):
    if penultimate_features:
      raise NotImplementedError("Not needed/implemented for this arch (net6c.py:78-80)")
    x = self.trunk(x)
    if trunk_features:
      return _to_chw_order(x, self.head)
    return self.head(x, kmeans_use_features=kmeans_use_features)


def _to_chw_order(feats, head):
  """Trunk features in the reference's (c, h, w) flatten order (net6c.py:24-25): what the trunk emits."""
  return feats


class ClusterNet6cTwoHead(_ApplyCounter, nn.Module):
  """net6c_two_head.py:53-98."""
  cfg = ClusterNet6c.cfg

  def __init__(self, config):
    super(ClusterNet6cTwoHead, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.trunk = ClusterNet6cTrunk(config)
    self.head_A = ClusterNet6cHead(config, output_k=config.output_k_A)
    semisup = hasattr(config, "semisup") and config.semisup
    assert not semisup, "semisup head is outside the IIC hot path (SURVEY.md §2 row 13)"
    self.head_B = ClusterNet6cHead(config, output_k=config.output_k_B)
    _initialize_weights_vgg(self)

  def forward_packed(self, x, head="B"):
    return (self.head_A if head == "A" else self.head_B).forward_packed(self.trunk(x))

  @ops.auto_branch
  def forward(self, x, head="B", kmeans_use_features=False, trunk_features=False,
              penultimate_features=False):
    if penultimate_features:
      raise NotImplementedError("Not needed/implemented for this arch")
    x = self.trunk(x)
    if trunk_features:
      return _to_chw_order(x, self.head_B)
    if head == "A":
      return self.head_A(x, kmeans_use_features=kmeans_use_features)
    elif head == "B":
      return self.head_B(x, kmeans_use_features=kmeans_use_features)
    raise AssertionError("head must be A or B")
