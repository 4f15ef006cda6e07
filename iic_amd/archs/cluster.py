"""MI355X-native ClusterNet5g / ClusterNet5gTwoHead -- drop-in for
/root/reference/code/archs/cluster/{net5g,net5g_two_head,residual}.py.

Same constructor (an argparse-like ``config`` with in_channels, input_sz, batchnorm_track,
num_sub_heads, output_k | output_k_A + output_k_B), same ``forward`` signature and return
type (a list of per-sub-head softmax tensors), same ``state_dict`` keys / shapes / dtypes
and the same init distributions (residual.py:75-85), so the reference's training scripts,
checkpoints and torch.optim.Adam drive it unchanged.

What differs is everything underneath: the nn.Conv2d / nn.BatchNorm2d / nn.Linear members
are only PARAMETER HOLDERS (their own forward is never called).  The computation is
three kinds of torch.autograd.Function whose forward/backward enqueue the hand-written
gfx950 kernels of libiic_hip.so:

  _StemFn    conv3x3 + BN + ReLU + maxpool, recomputed from the fp32 NCHW input (stem.hip)
  _BlockFn   BasicBlock: 2-3 bf16-MFMA implicit-GEMM convs with BN statistics fused in the
             conv epilogue, BN/ReLU/residual streaming kernels, MFMA weight-grad (conv_*.hip)
  _HeadsFn   avgpool + all sub-heads as one fp32-MFMA GEMM + softmax (head.hip)

Activations between Functions are "PT" tensors: bf16 [N, H+2, W+2, C] with a zero border.
There is no PyTorch / CPU fallback: a CPU input raises.
"""
import torch
import torch.nn as nn

from .. import geom as G
from .. import ops

__all__ = ["ClusterNet5g", "ClusterNet5gTwoHead"]

import os
import weakref

from ..dist import SHARD_INPUTS, shard_batch  # noqa: E402  (set by iic_amd.run under torchrun)
# Replica de-duplication (opt-in: IIC_DEDUP=<r> or DEDUP[0] = r).  The reference
# replicates imgs_curr num_dataloaders times in all_imgs (cluster_sobel.py:215-226).  When a training
# batch consists of r exact copies of its first B/r rows, the trunk runs on those rows only and
# the 512-d features are repeated r times (autograd sums the replicas' gradients): batch mean and
# biased variance of every BatchNorm are invariant under exact replication and the backward is
# linear in the upstream gradient, so outputs and parameter gradients are unchanged; the unbiased
# running_var factor uses the true batch size.  Removes 1/3 of view 1's conv work at r = 3.
# NOT used by bench.py (it would change the FLOP accounting of the metric).
DEDUP = [int(os.environ.get("IIC_DEDUP", "1"))]
_ASSERTED_REPLICAS = [1]


class replicated(object):
  """``with replicated(r): out = net(all_imgs)`` -- the CALLER asserts that the batch is r exact
  copies of its first B/r rows (it built it that way: cluster_sobel.py:215-226), so the trunk
  de-duplicates without comparing the rows.  The IIC_DEDUP / DEDUP[0] switch serves unchanged
  scripts, which cannot say which of their two forwards carries the replicated batch: there the
  rows are compared on the device and the verdict is read back (one host sync per forward)."""

  def __init__(self, r):
    self.r = int(r)

  def __enter__(self):
    self.prev = _ASSERTED_REPLICAS[0]
    _ASSERTED_REPLICAS[0] = self.r
    return self

  def __exit__(self, *exc):
    _ASSERTED_REPLICAS[0] = self.prev
    return False
# Pre-masked gradient chain through the residual trunk (IIC_PREMASK=0 disables it).  Every block
# hands its input gradient over already multiplied by the ReLU mask of that input (the conv
# backward-data epilogue applies it: IIC_ACC_PREMASK), and the average-pool backward does the same
# for the last block.  A block's gradient g = dout * (out > 0) then arrives ready-made: its second
# BatchNorm backward (reduce + apply) and the residual-gradient fusion no longer read `out` -- two
# full-tensor reads less per block and view, same values bit for bit (masking commutes with the
# bf16 rounding of the stored gradient).  Only valid where a block output has exactly one
# consumer, i.e. the trunk's own sequential forward; standalone blocks keep masking themselves.
PREMASK = [os.environ.get("IIC_PREMASK", "1") != "0"]
# Fused BatchNorm-backward reductions (IIC_FUSE_RED=0 disables them).  The two sums every BatchNorm
# backward needs (sum g, sum g*y) used to be a separate HBM-bound pass over the gradient g and the
# BatchNorm input y.  The gradient is produced by a backward-data convolution -- MFMA-bound, with
# HBM to spare -- so that launch takes the sums in its epilogue (iic_conv_igemm_frag_red) and only
# y is read in addition:
#   * a block's conv2 backward-data produces da1 -> sums of that block's bn1;
#   * a stride-1 block's conv1 backward-data produces the (pre-masked) gradient of its INPUT, which
#     is the output gradient of the PREVIOUS block -> sums of the previous block's bn2 (and of its
#     downsample BatchNorm).  The trunk links consecutive blocks through a _Chain object at
#     forward time; the previous block then skips its own reduction (ctx.dout_prereduced).
FUSE_RED = [os.environ.get("IIC_FUSE_RED", "1") != "0"]


class _Chain(object):
  """Per trunk forward: what the next block needs to know about the block before it."""
  __slots__ = ("ctx", "y2", "yd", "blk")

  def __init__(self):
    self.ctx = self.y2 = self.yd = self.blk = None

_WEIGHTS_EPOCH = [0]   # bumped by iic_amd.optim.Adam (raw-pointer updates do not bump _version)


def bump_weights_epoch():
  _WEIGHTS_EPOCH[0] += 1


_HOLDERS = {}     # branch -> WeakSet of _ConvHolder with an operand set in that branch


def _refresh_branch(b):
  stale = []
  for h in list(_HOLDERS.get(b, ())):
    ent = h._wbranch.get(b)
    w = h.conv.weight
    if ent is None or not w.is_cuda:
      continue
    key = (w.data_ptr(), w._version, _WEIGHTS_EPOCH[0])
    if ent[0] != key and ent[0][0] == key[0]:
      stale.append((h, ent, key))
  if not stale:
    return
  by_dev = {}
  for h, ent, key in stale:
    by_dev.setdefault(ent[1].w.device, []).append((ent, key))
  for dev, items in by_dev.items():
    if ops.refresh_prepped([ent[1] for ent, _ in items], dev):
      for ent, key in items:
        ent[0] = key


class _ConvHolder(object):
  """Per-conv caches: bf16 operand copies of the fp32 parameter, geometries, BN stat buffers."""

  def __init__(self, conv, pad_in=1, pad_out=1):
    self.conv = conv
    self.pad_in, self.pad_out = pad_in, pad_out
    self.spec = G.ConvSpec(conv.in_channels, conv.out_channels, conv.kernel_size[0],
                           conv.stride[0], conv.padding[0], conv.dilation[0])
    self._wbranch = {}      # branch -> [key, PreppedWeights]
    self._geoms = {}
    self._stats = {}

  def weights(self):
    # (one operand set per branch: a side branch must not read operands whose layout kernels
    # were enqueued on the other stream)
    w = self.conv.weight
    b = ops.BRANCH[0]
    key = (w.data_ptr(), w._version, _WEIGHTS_EPOCH[0])
    ent = self._wbranch.get(b)                  # [key, PreppedWeights]
    if ent is not None and ent[0] == key:
      return ent[1]
    if ent is None or ent[0][0] != key[0] or ent[1].w.device != w.device or not ops.MULTI_PREP[0]:
      # first use (or the parameter moved): a fresh operand set, layouts made on demand
      ent = self._wbranch[b] = [key, ops.PreppedWeights(w.detach())]
      _HOLDERS.setdefault(b, weakref.WeakSet()).add(self)
      return ent[1]
    # the optimiser has stepped: re-write the layouts of EVERY convolution of this branch that is stale, in one
    # launch, instead of one or two small launches in front of each convolution
    _refresh_branch(b)
    ent = self._wbranch[b]
    if ent[0] != key:                           # (refresh not possible right now: per-layout launches)
      ent = self._wbranch[b] = [key, ops.PreppedWeights(w.detach())]
    return ent[1]

  def geoms(self, N, H, W):
    key = (N, H, W)
    g = self._geoms.get(key)
    if g is None:
      g = (G.fwd_geom(self.spec, N, H, W, self.pad_in, self.pad_out),
           G.bwd_data_geoms(self.spec, N, H, W, self.pad_out, self.pad_in))
      self._geoms[key] = g
    return g

  def stats(self, device, which="fwd"):
    key = (str(device), which, ops.BRANCH[0])
    s = self._stats.get(key)
    if s is None:
      s = ops.new_stats(self.spec.cout, device)
      self._stats[key] = s
    return s


def _bn_buffers(bn):
  if bn.track_running_stats:
    return bn.running_mean, bn.running_var, bn.num_batches_tracked
  return None, None, None


def _bn_training(bn):
  # nn.BatchNorm2d semantics: batch statistics when training OR when no running stats exist
  return bn.training or not bn.track_running_stats


# ------------------------------------------------------------------------------------
# Stem
# ------------------------------------------------------------------------------------
BN_MASK_FROM_Y = [os.environ.get("IIC_BN_MASK_FROM_Y", "1") != "0"]
STEM_FUSED_BWD = [os.environ.get("IIC_STEM_FUSED", "1") != "0"]   # 0: two-pass backward (cross-check)


class _StemFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, w, gamma, beta, mod):
    assert x.is_cuda, "ClusterNet5g (HIP): input must be a device tensor -- no CPU fallback"
    x = x.contiguous().float()
    N, C, H, W = x.shape
    bn = mod.bn1
    rm, rv, nbt = _bn_buffers(bn)
    training = _bn_training(bn)
    wd = w.detach()
    if training:
      st = mod._h_conv1.stats(x.device)
      ops.stem_stats(x, wd, st)
      coef = ops.bn_finalize(st, gamma.detach(), beta.detach(), rm if bn.training else None,
                             rv if bn.training else None, nbt if bn.training else None, 64,
                             N * H * W, True)
    else:
      coef = ops.bn_finalize(None, gamma.detach(), beta.detach(), rm, rv, None, 64, N * H * W, False)
    Ho, Wo = H // 2 + 1, W // 2 + 1
    out = ops.pt_alloc(N, Ho, Wo, 64, 1, x.device)
    ops.stem_apply_pool(x, wd, coef, out)
    ctx.mod = mod
    ctx.training = training
    ctx.branch, ctx.pt_dtype = ops.BRANCH[0], ops.PT_DTYPE[0]
    ctx.save_for_backward(x, w, gamma, coef, out)
    return out

  @ops.branch_backward
  def backward(ctx, dpool):
    x, w, gamma, coef, out = ctx.saved_tensors
    if not ctx.training:
      raise RuntimeError("HIP BatchNorm backward is implemented for batch statistics only")
    N, C, H, W = x.shape
    dpool = dpool.contiguous()
    sums = ctx.mod._h_conv1.stats(x.device, "bwd")
    if STEM_FUSED_BWD[0] and ops.stem_bwd_fused_ok(C):
      # one recompute pass: BN-backward sums + coefficient-free dW GEMMs, coefficients applied after
      h = ops.stem_bwd_fused(x, w.detach(), coef, dpool, sums)
      bcoef, dgamma, dbeta = ops.bn_bwd_finalize(sums, gamma.detach(), coef, 64, N * H * W)
      dW = ops.stem_wgrad_combine(h, bcoef, w.detach())
    else:
      ops.stem_bwd_reduce(x, w.detach(), coef, dpool, sums)
      bcoef, dgamma, dbeta = ops.bn_bwd_finalize(sums, gamma.detach(), coef, 64, N * H * W)
      dW = ops.stem_bwd_wgrad(x, w.detach(), coef, bcoef, dpool)
    ops.POOL.release(dpool)
    ops.POOL.release(out)
    return None, dW, dgamma, dbeta, None


class _StemF32Fn(torch.autograd.Function):
  """The stem inside ops.fp32_mode(): conv3x3 -> BN -> ReLU -> MaxPool2d(2, 2, padding=1) as the
  plain fp32 kernels of csrc/f32_path.hip on materialised tensors (net5g.py:14-26, 42-45)."""

  @staticmethod
  def forward(ctx, x, w, gamma, beta, mod):
    assert x.is_cuda
    x = x.contiguous().float()
    N, C, H, W = x.shape
    bn, h = mod.bn1, mod._h_conv1
    rm, rv, nbt = _bn_buffers(bn)
    training = _bn_training(bn)
    dev = x.device
    xp = ops.f32_nchw_to_pt(x, ops.pt_alloc(N, H, W, C, 1, dev), 1)
    gf, _ = h.geoms(N, H, W)
    st = h.stats(dev) if training else None
    y = ops.pt_alloc(N, H, W, 64, 1, dev)
    ops.conv_igemm(gf, xp, h.weights()[0], y, stats=st)
    if training:
      coef = ops.bn_finalize(st, gamma.detach(), beta.detach(), rm if bn.training else None,
                             rv if bn.training else None, nbt if bn.training else None, 64, N * H * W, True)
    else:
      coef = ops.bn_finalize(None, gamma.detach(), beta.detach(), rm, rv, None, 64, N * H * W, False)
    a = ops.bn_apply(y, coef, ops.pt_alloc(N, H, W, 64, 1, dev), N, H, W, 1, 64, relu=True)
    Ho, Wo = H // 2 + 1, W // 2 + 1
    out = ops.f32_maxpool_s2p1_fwd(a, ops.pt_alloc(N, Ho, Wo, 64, 1, dev), N, H, W, 64)
    ctx.mod, ctx.training, ctx.branch, ctx.pt_dtype = mod, training, ops.BRANCH[0], ops.PT_DTYPE[0]
    ctx.dims = (N, C, H, W)
    ctx.save_for_backward(xp, y, a, coef, gamma)
    return out

  @ops.branch_backward
  def backward(ctx, dpool):
    xp, y, a, coef, gamma = ctx.saved_tensors
    if not ctx.training:
      raise RuntimeError("HIP BatchNorm backward is implemented for batch statistics only")
    N, C, H, W = ctx.dims
    dev, h = xp.device, ctx.mod._h_conv1
    with ops.fp32_mode():
      da = ops.f32_maxpool_s2p1_bwd(a, dpool.contiguous(), ops.pt_alloc(N, H, W, 64, 1, dev), N, H, W, 64)
      sums = h.stats(dev, "bwd")
      ops.bn_bwd_reduce(da, None, y, sums, N, H, W, 1, 64, mask_coef=coef)
      bcoef, dgamma, dbeta = ops.bn_bwd_finalize(sums, gamma.detach(), coef, 64, N * H * W)
      dy = ops.pt_alloc(N, H, W, 64, 1, dev)
      ops.bn_bwd_apply(da, None, y, bcoef, dy, N, H, W, 1, 64, mask_coef=coef)
      gf, _ = h.geoms(N, H, W)
      dW = ops.conv_wgrad(gf, xp, dy, 9).view(64, C, 3, 3)
    for t in (dpool, da, dy, xp, y, a):
      ops.POOL.release(t)
    return None, dW, dgamma, dbeta, None


# ------------------------------------------------------------------------------------
# BasicBlock  (residual.py:10-43)
# ------------------------------------------------------------------------------------
class _BlockFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, blk, chain=None):
    N, Hp, Wp, Cin = x.shape
    H, W = Hp - 2, Wp - 2
    dev = x.device
    h1, h2, hd = blk._h1, blk._h2, blk._hd
    planes = h1.spec.cout
    Ho, Wo = h1.spec.out_size(H), h1.spec.out_size(W)
    cnt = N * Ho * Wo
    need_grad = any(ctx.needs_input_grad)   # (grad mode is off inside Function.forward)

    def bn_coef(bn, holder, gamma, beta, y_stats):
      rm, rv, nbt = _bn_buffers(bn)
      if _bn_training(bn):
        upd = bn.training
        return ops.bn_finalize(y_stats, gamma.detach(), beta.detach(), rm if upd else None,
                               rv if upd else None, nbt if upd else None, planes, cnt, True)
      return ops.bn_finalize(None, gamma.detach(), beta.detach(), rm, rv, None, planes, cnt, False)

    gf1, _ = h1.geoms(N, H, W)
    gf2, _ = h2.geoms(N, Ho, Wo)
    st1 = h1.stats(dev) if _bn_training(blk.bn1) else None
    y1 = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
    ops.conv_igemm(gf1, x, h1.weights()[0], y1, stats=st1)
    coef1 = bn_coef(blk.bn1, h1, g1, b1, st1)
    # conv2 reads a1 = relu(bn1(y1)), made by a separate HBM pass.  (Round 4 built the fusion of that pass into conv2's
    # patch loader -- in LDS after the DMA -- bit-identical and 0.6-0.9 ms per step SLOWER; removed in round 5,
    # LAB.md section R5.3 has the arithmetic of why the register variant cannot pay either.)
    st2 = h2.stats(dev) if _bn_training(blk.bn2) else None
    y2 = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
    a1 = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
    ops.bn_apply(y1, coef1, a1, N, Ho, Wo, 1, planes, relu=True)
    ops.conv_igemm(gf2, a1, h2.weights()[0], y2, stats=st2)
    coef2 = bn_coef(blk.bn2, h2, g2, b2, st2)
    out = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
    yd = coefd = None
    if hd is not None:
      gfd, _ = hd.geoms(N, H, W)
      bnd = blk.downsample[1]
      std = hd.stats(dev) if _bn_training(bnd) else None
      yd = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
      ops.conv_igemm(gfd, x, hd.weights()[0], yd, stats=std)
      coefd = bn_coef(bnd, hd, gd, bd, std)
      ops.bn_apply(y2, coef2, out, N, Ho, Wo, 1, planes, y2=yd, coef2=coefd, relu=True)
    else:
      ops.bn_apply(y2, coef2, out, N, Ho, Wo, 1, planes, res=x, relu=True)

    if need_grad:
      ctx.blk = blk
      ctx.branch, ctx.pt_dtype = ops.BRANCH[0], ops.PT_DTYPE[0]
      ctx.dims = (N, H, W, Ho, Wo, Cin, planes)
      ctx.bn_batch = (_bn_training(blk.bn1) and _bn_training(blk.bn2))
      # (set by the trunk for the duration of its sequential forward, see PREMASK)
      ctx.dout_premasked, ctx.mask_dx = blk._dout_premasked, blk._mask_dx
      # fused reduction of the PREVIOUS block's bn2 in this block's conv1 backward-data
      ctx.dout_prereduced = False
      ctx.red_prev = None
      if chain is not None:
        _, gb1f = h1.geoms(N, H, W)
        if (chain.ctx is not None and FUSE_RED[0] and hd is None and blk._mask_dx and len(gb1f) == 1
            and ops.red_supported(gb1f[0], h1.weights()[1]) and ctx.bn_batch):
          ctx.red_prev = (chain.y2, chain.yd, chain.blk)
          chain.ctx.dout_prereduced = True
        if blk._dout_premasked and ctx.bn_batch:
          chain.ctx, chain.y2, chain.yd, chain.blk = ctx, y2, yd, blk
        else:
          chain.ctx = None
      ctx.save_for_backward(x, y1, a1, y2, out, yd, coef1, coef2, coefd, g1, g2, gd)
    else:
      if chain is not None:
        chain.ctx = None
      for t in (y1, a1, y2, yd):
        if t is not None:
          ops.POOL.release(t)
    return out

  @ops.branch_backward
  def backward(ctx, dout):
    x, y1, a1, y2, out, yd, coef1, coef2, coefd, g1, g2, gd = ctx.saved_tensors
    blk = ctx.blk
    if not ctx.bn_batch:
      raise RuntimeError("HIP BatchNorm backward is implemented for batch statistics only")
    N, H, W, Ho, Wo, Cin, planes = ctx.dims
    dev = x.device
    h1, h2, hd = blk._h1, blk._h2, blk._hd
    cnt = N * Ho * Wo
    dout = dout.contiguous()
    use_tr = blk._use_tr
    gf1, gb1 = h1.geoms(N, H, W)
    gf2, gb2 = h2.geoms(N, Ho, Wo)

    # ---- bn2 (+ downsample bn) backward; g = dout * (out > 0), or dout itself when the consumer
    # of `out` already applied that mask (PREMASK)
    pre, mask_dx = ctx.dout_premasked, ctx.mask_dx
    m_out = None if pre else out
    s2 = h2.stats(dev, "bwd")
    sd = hd.stats(dev, "bwd") if hd is not None else None
    if not ctx.dout_prereduced:      # else: the next block's conv1 backward-data took these sums
      ops.bn_bwd_reduce(dout, m_out, y2, s2, N, Ho, Wo, 1, planes, y2=yd, sums2=sd)
    bc2, dg2, db2 = ops.bn_bwd_finalize(s2, g2.detach(), coef2, planes, cnt)
    dy2 = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
    bcd = dgd = dbd = dyd = None
    if hd is not None:
      bcd, dgd, dbd = ops.bn_bwd_finalize(sd, gd.detach(), coefd, planes, cnt)
      dyd = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
    ops.bn_bwd_apply(dout, m_out, y2, bc2, dy2, N, Ho, Wo, 1, planes, y2=yd, bcoef2=bcd, dy2=dyd)

    # ---- conv2 backward: weight grad, data grad + bn1 backward
    dW2 = ops.conv_wgrad(gf2, a1, dy2, 9, use_tr).view(planes, planes, 3, 3)
    dWd = None
    if hd is not None:
      gfd, _ = hd.geoms(N, H, W)
      dWd = ops.conv_wgrad(gfd, x, dyd, 1, use_tr).view(planes, Cin, 1, 1)
    da1 = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
    # ---- bn1 backward; g1 = da1 * (a1 > 0)
    s1 = h1.stats(dev, "bwd")
    # a1 = relu(bn1(y1)) exactly: the ReLU mask comes from (y1, coef1), a1 is not read
    m_act, m_coef = (None, coef1) if BN_MASK_FROM_Y[0] else (a1, None)
    w2b = h2.weights()[1]
    if FUSE_RED[0] and m_coef is not None and len(gb2) == 1 and ops.red_supported(gb2[0], w2b):
      # the conv2 backward-data launch takes bn1's sums in its epilogue (reads y1 beside da1)
      ops.conv_igemm(gb2[0], dy2, w2b, da1, red=(y1, coef1, s1, None, None))
    else:
      for g in gb2:
        ops.conv_igemm(g, dy2, w2b, da1)
      ops.bn_bwd_reduce(da1, m_act, y1, s1, N, Ho, Wo, 1, planes, mask_coef=m_coef)
    bc1, dg1, db1 = ops.bn_bwd_finalize(s1, g1.detach(), coef1, planes, cnt)
    dy1 = ops.pt_alloc(N, Ho, Wo, planes, 1, dev)
    ops.bn_bwd_apply(da1, m_act, y1, bc1, dy1, N, Ho, Wo, 1, planes, mask_coef=m_coef)

    # ---- conv1 backward: weight grad, data grad (+ residual / downsample gradient)
    dW1 = ops.conv_wgrad(gf1, x, dy1, 9, use_tr).view(planes, Cin, 3, 3)
    dx = ops.pt_alloc(N, H, W, Cin, 1, dev)
    # dx = bwd-data(conv1) + identity-branch gradient [, times the ReLU mask of x: mask_dx]
    mx = x if mask_dx else None
    red_prev = None
    if ctx.red_prev is not None:
      py2, pyd, pblk = ctx.red_prev
      red_prev = (py2, None, pblk._h2.stats(dev, "bwd"), pyd,
                  pblk._hd.stats(dev, "bwd") if pyd is not None else None)
    if hd is None:
      for g in gb1:
        if pre or mask_dx:
          assert pre, "a block that pre-masks its input gradient needs a pre-masked output gradient"
          ops.conv_igemm(g, dy1, h1.weights()[1], dx, res_grad=dout, res_act=mx, premask=True,
                         red=red_prev)
        else:
          ops.conv_igemm(g, dy1, h1.weights()[1], dx, res_grad=dout, res_act=out)
    else:
      for g in gb1:
        ops.conv_igemm(g, dy1, h1.weights()[1], dx, res_act=mx, premask=mask_dx)
      _, gbd = hd.geoms(N, H, W)
      for g in gbd:      # (the mask is idempotent: pixels this launch adds to are masked again)
        ops.conv_igemm(g, dyd, hd.weights()[1], dx, accumulate=True, res_act=mx, premask=mask_dx)

    for t in (dout, dy2, dyd, da1, dy1, y1, a1, y2, yd, out):
      ops.POOL.release(t)
    return dx, dW1, dg1, db1, dW2, dg2, db2, dWd, dgd, dbd, None, None


# nn.Module._apply (.cpu() / .cuda() / .to(): the reference's scripts move the whole network around every checkpoint,
# cluster_sobel.py:314-339) keeps the Parameter objects and REPLACES the buffer objects; iic_amd.graphed compares a
# captured graph's baked-in addresses against a cached tensor list and must know when that list is stale.
APPLY_GENERATION = [0]


class _ApplyCounter(object):
  def _apply(self, fn, *a, **k):
    APPLY_GENERATION[0] += 1
    return super(_ApplyCounter, self)._apply(fn, *a, **k)


class BasicBlock(nn.Module):
  expansion = 1

  def __init__(self, inplanes, planes, stride=1, downsample=None, track_running_stats=None):
    super(BasicBlock, self).__init__()
    assert track_running_stats is not None
    self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
    self.bn1 = nn.BatchNorm2d(planes, track_running_stats=track_running_stats)
    self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
    self.bn2 = nn.BatchNorm2d(planes, track_running_stats=track_running_stats)
    self.downsample = downsample
    self.stride = stride
    self._h1 = _ConvHolder(self.conv1)
    self._h2 = _ConvHolder(self.conv2)
    self._hd = _ConvHolder(downsample[0]) if downsample is not None else None
    self._use_tr = True
    self._dout_premasked = False      # see PREMASK; only the trunk's forward turns these on
    self._mask_dx = False
    self._chain = None                # see FUSE_RED

  def forward(self, x, chain=None):
    chain = self._chain if chain is None else chain
    ds = self.downsample
    pv = ops.pv
    return _BlockFn.apply(
      x, pv(self.conv1.weight), pv(self.bn1.weight), pv(self.bn1.bias), pv(self.conv2.weight),
      pv(self.bn2.weight), pv(self.bn2.bias), pv(ds[0].weight) if ds is not None else None,
      pv(ds[1].weight) if ds is not None else None, pv(ds[1].bias) if ds is not None else None, self,
      chain)


# ------------------------------------------------------------------------------------
# avgpool + heads  (net5g.py:31-39,53,61-80)
# ------------------------------------------------------------------------------------
class _AvgPoolFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, premask):
    N, Hp, Wp, C = x.shape
    ctx.dims = (N, Hp - 2, Wp - 2, C)
    ctx.branch, ctx.pt_dtype = ops.BRANCH[0], ops.PT_DTYPE[0]
    ctx.premask = bool(premask)
    if premask:
      ctx.save_for_backward(x)      # its ReLU mask is applied to the gradient here (PREMASK)
    return ops.avgpool_fwd(x, N, Hp - 2, Wp - 2, 1, C)

  @ops.branch_backward
  def backward(ctx, dfeats):
    N, H, W, C = ctx.dims
    dx = ops.pt_alloc(N, H, W, C, 1, dfeats.device)
    act = ctx.saved_tensors[0] if ctx.premask else None
    return ops.avgpool_bwd(dfeats.contiguous(), dx, N, H, W, 1, C, mask_act=act), None


class _HeadsFn(torch.autograd.Function):
  """feats [N, F] fp32, Wcat [H*k, F], bcat [H*k] -> probs [N, H, k] (sample-major)."""

  @staticmethod
  def forward(ctx, feats, Wcat, bcat, H, k):
    N, F = feats.shape
    feats = feats.contiguous()
    Wcat = Wcat.contiguous()
    KT = H * k
    logits = torch.empty((N, KT), dtype=torch.float32, device=feats.device)
    # logits[n][j] = sum_c feats[n][c] * Wcat[j][c] + b[j]
    ops.gemm_f32(feats, F, 1, Wcat, 1, F, logits, KT, N, KT, F, bias=bcat.contiguous())
    probs = ops.softmax_fwd(logits, N * H, k)
    ctx.save_for_backward(feats, Wcat, probs)
    ctx.hk = (H, k)
    ctx.branch, ctx.pt_dtype = ops.BRANCH[0], ops.PT_DTYPE[0]   # (the K-split GEMM workspace is per branch)
    return probs.view(N, H, k)

  @ops.branch_backward
  def backward(ctx, dprobs):
    feats, Wcat, probs = ctx.saved_tensors
    H, k = ctx.hk
    N, F = feats.shape
    KT = H * k
    dlog = ops.softmax_bwd(probs, dprobs.contiguous().view(N, KT), N * H, k)
    # dW[j][c] = sum_n dlog[n][j] feats[n][c]
    dW = torch.empty((KT, F), dtype=torch.float32, device=feats.device)
    ops.gemm_f32(dlog, 1, KT, feats, F, 1, dW, F, KT, F, N)
    db = ops.colsum(dlog, N, KT)
    # dfeats[n][c] = sum_j dlog[n][j] Wcat[j][c]
    dfe = torch.empty((N, F), dtype=torch.float32, device=feats.device)
    ops.gemm_f32(dlog, KT, 1, Wcat, F, 1, dfe, F, N, F, KT)
    return dfe, dW, db, None, None


class ClusterNet5gHead(nn.Module):
  def __init__(self, config, output_k=None):
    super(ClusterNet5gHead, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.num_sub_heads = config.num_sub_heads
    self.output_k = config.output_k if output_k is None else output_k
    self.heads = nn.ModuleList([nn.Sequential(
      nn.Linear(512 * BasicBlock.expansion, self.output_k),
      nn.Softmax(dim=1)) for _ in range(self.num_sub_heads)])

  def forward_packed(self, feats):
    """probs [N, H, k] fp32 (all sub-heads, one GEMM)."""
    Wcat = torch.cat([ops.pv(h[0].weight) for h in self.heads], dim=0)
    bcat = torch.cat([ops.pv(h[0].bias) for h in self.heads], dim=0)
    return _HeadsFn.apply(feats, Wcat, bcat, self.num_sub_heads, self.output_k)

  def forward(self, x, kmeans_use_features=False):
    if kmeans_use_features:
      return [x for _ in range(self.num_sub_heads)]   # duplicates, as the reference
    probs = self.forward_packed(x)
    return ops.tag_pack([probs[:, i, :] for i in range(self.num_sub_heads)])


# ------------------------------------------------------------------------------------
# trunk  (net5g.py:10-58, residual.py:46-68)
# ------------------------------------------------------------------------------------
class ClusterNet5gTrunk(nn.Module):
  def __init__(self, config):
    super(ClusterNet5gTrunk, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    layers = [3, 4, 6, 3]
    self.inplanes = 64
    self.conv1 = nn.Conv2d(config.in_channels, 64, kernel_size=3, stride=1, padding=1, bias=False)
    self.bn1 = nn.BatchNorm2d(64, track_running_stats=self.batchnorm_track)
    self.layer1 = self._make_layer(64, layers[0])
    self.layer2 = self._make_layer(128, layers[1], stride=2)
    self.layer3 = self._make_layer(256, layers[2], stride=2)
    self.layer4 = self._make_layer(512, layers[3], stride=2)
    assert config.input_sz in (96, 64, 32)
    self.input_sz = config.input_sz
    self._h_conv1 = _ConvHolder(self.conv1)

  def _make_layer(self, planes, blocks, stride=1):
    downsample = None
    if stride != 1 or self.inplanes != planes:
      downsample = nn.Sequential(
        nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
        nn.BatchNorm2d(planes, track_running_stats=self.batchnorm_track))
    layers = [BasicBlock(self.inplanes, planes, stride, downsample,
                         track_running_stats=self.batchnorm_track)]
    self.inplanes = planes
    for _ in range(1, blocks):
      layers.append(BasicBlock(self.inplanes, planes, track_running_stats=self.batchnorm_track))
    return nn.Sequential(*layers)

  def forward(self, x, penultimate_features=False):
    x = shard_batch(x, self)
    asserted = _ASSERTED_REPLICAS[0] > 1
    r = _ASSERTED_REPLICAS[0] if asserted else DEDUP[0]
    if r > 1 and self.training and x.size(0) % r == 0 and x.size(0) >= 2 * r:
      u = x.size(0) // r
      if asserted or all(bool(torch.equal(x[:u], x[i * u:(i + 1) * u])) for i in range(1, r)):
        ops.BN_REPLICAS[0] = r
        try:
          f = self._run(x[:u], penultimate_features)
        finally:
          ops.BN_REPLICAS[0] = 1
        return f.repeat(r, 1)
    return self._run(x, penultimate_features)

  def _run(self, x, penultimate_features):
    # PREMASK: in the plain sequential forward every block output has one consumer (the next
    # block, or the average pool), so gradients can travel pre-masked; the penultimate-features
    # path taps layer3's output with torch ops and keeps the self-masking blocks
    blocks = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]
    chain = PREMASK[0] and not penultimate_features
    link = _Chain() if chain else None    # consecutive blocks of the pre-masked chain (FUSE_RED)
    for i, b in enumerate(blocks):
      b._dout_premasked = chain
      b._mask_dx = chain and i > 0        # block 0's input gradient goes to the stem (own masking)
      b._chain = link
    try:
      stem = _StemF32Fn if ops.PT_DTYPE[0] is torch.float32 else _StemFn
      # layer-group boundaries for a STAGED backward (iic_amd.graph.CapturedPairStep, data parallel: the
      # gradient bucket of a group is all-reduced while the groups below it still run backward)
      taps = getattr(self, "_tap_sink", None)
      x = stem.apply(x, ops.pv(self.conv1.weight), ops.pv(self.bn1.weight), ops.pv(self.bn1.bias), self)
      x = self.layer1(x)
      if taps is not None:
        taps.append(x)
      x = self.layer2(x)
      if taps is not None:
        taps.append(x)
      x = self.layer3(x)
      if taps is not None:
        taps.append(x)
      if penultimate_features:
        return ops.pt_to_nchw(x, 1).reshape(x.size(0), -1)
      x = self.layer4(x)
      return _AvgPoolFn.apply(x, chain)   # avg_pool_sz == final spatial size for 96 / 64 / 32 inputs
    finally:
      for b in blocks:
        b._dout_premasked = b._mask_dx = False
        b._chain = None


def _initialize_weights(net):
  """residual.py:75-85."""
  for m in net.modules():
    if isinstance(m, nn.Conv2d):
      nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    elif isinstance(m, nn.BatchNorm2d):
      assert m.track_running_stats == net.batchnorm_track
      m.weight.data.fill_(1)
      m.bias.data.zero_()
    elif isinstance(m, nn.Linear):
      m.weight.data.normal_(0, 0.01)
      m.bias.data.zero_()


class ClusterNet5g(_ApplyCounter, nn.Module):
  def __init__(self, config):
    super(ClusterNet5g, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.trunk = ClusterNet5gTrunk(config)
    self.head = ClusterNet5gHead(config)
    _initialize_weights(self)

  def set_wgrad_tr(self, flag):
    for m in self.modules():
      if isinstance(m, BasicBlock):
        m._use_tr = bool(flag)

  def forward_packed(self, x):
    """All sub-head outputs as one [N, H, k] tensor (feeds IID_loss_heads; 3 loss launches)."""
    return self.head.forward_packed(self.trunk(x))

  def grad_groups(self):
    """Parameters by layer group in the order backward finishes their gradients: [heads + layer4, layer3,
    layer2, layer1 + stem] -- the gradient buckets of the data-parallel step."""
    t = self.trunk
    return [list(self.head.parameters()) + list(t.layer4.parameters()), list(t.layer3.parameters()),
            list(t.layer2.parameters()),
            list(t.layer1.parameters()) + list(t.conv1.parameters()) + list(t.bn1.parameters())]

  def forward_packed_taps(self, x):
    """forward_packed plus the activations at the group boundaries, [input of group 0, of group 1, ...]
    (= outputs of layer3, layer2, layer1): what a staged backward differentiates through."""
    sink = []
    self.trunk._tap_sink = sink
    try:
      out = self.forward_packed(x)
    finally:
      self.trunk._tap_sink = None
    return out, sink[::-1]

  @ops.auto_branch
  def forward(self, x, kmeans_use_features=False, trunk_features=False, penultimate_features=False):
    x = self.trunk(x, penultimate_features=penultimate_features)
    if trunk_features:
      return x
    return self.head(x, kmeans_use_features=kmeans_use_features)


class ClusterNet5gTwoHead(_ApplyCounter, nn.Module):
  """net5g_two_head.py:42-81 (head A = overclustering output_k_A, head B = output_k_B)."""

  def __init__(self, config):
    super(ClusterNet5gTwoHead, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.trunk = ClusterNet5gTrunk(config)
    self.head_A = ClusterNet5gHead(config, output_k=config.output_k_A)
    semisup = hasattr(config, "semisup") and config.semisup
    assert not semisup, "semisup head is outside the IIC hot path (SURVEY.md §2 row 13)"
    self.head_B = ClusterNet5gHead(config, output_k=config.output_k_B)
    _initialize_weights(self)

  set_wgrad_tr = ClusterNet5g.set_wgrad_tr

  def forward_packed(self, x, head="B"):
    return (self.head_A if head == "A" else self.head_B).forward_packed(self.trunk(x))

  @ops.auto_branch
  def forward(self, x, head="B", kmeans_use_features=False, trunk_features=False,
              penultimate_features=False):
    x = self.trunk(x, penultimate_features=penultimate_features)
    if trunk_features:
      return x
    if head == "A":
      return self.head_A(x, kmeans_use_features=kmeans_use_features)
    elif head == "B":
      return self.head_B(x, kmeans_use_features=kmeans_use_features)
    raise AssertionError("head must be A or B")
