"""MI355X-native SegmentationNet10a / SegmentationNet10aTwoHead -- drop-in for
/root/reference/code/archs/segmentation/{net10a,net10a_twohead}.py.

Trunk = VGG-style stages (first: fp32-MFMA conv from the image; others: bf16-MFMA implicit
GEMM incl. the two dilation-2 convs, expressed as tap offsets) on PT tensors with border
P = 3 (the backward-data of a dilation-2 3x3 conv with padding 1 reaches 3 pixels out).
Head = 1x1 conv with padding 1 as an fp32-MFMA GEMM over the (Hf+2)x(Wf+2) window of the PT
feature map, Softmax2d, bilinear up-sampling to input_sz -- all fp32 (feeds the loss).
state_dict keys / init follow the reference (vgg.py:8-54, net10a.py:34-80).
"""
import os

import torch
import torch.nn as nn

from .. import ops
from .._lib import check, lib, ptr, stream_ptr
from .cluster import _ApplyCounter
from .vgg import VGGTrunkHIP, _initialize_weights_vgg

__all__ = ["SegmentationNet10a", "SegmentationNet10aTwoHead"]
F32 = torch.float32


# 1: the head's three GEMMs run directly on the bf16 PT window (iic_seg_head_*); 0: window gather ->
# generic fp32 GEMM -> scatter (the exact-fp32 parity path always takes the latter).  A/B + tests.
FUSED_HEAD = [os.environ.get("IIC_SEG_FUSED_HEAD", "1") != "0"]


class _SegHeadFn(torch.autograd.Function):
  """x: PT bf16 [N, Hf+2P, Wf+2P, C]; w: [k, C, 1, 1] -> probabilities [N, k, S, S] fp32."""

  @staticmethod
  def forward(ctx, x, w, P, S):
    N, Hp, Wp, C = x.shape
    Hw, Ww, off = Hp - 2 * P + 2, Wp - 2 * P + 2, P - 1
    k = w.shape[0]
    M = N * Hw * Ww
    L, s = lib(), stream_ptr()
    W2 = w.detach().reshape(k, C).contiguous()
    logits = torch.empty((M, k), dtype=F32, device=x.device)
    # fused path: the GEMMs read / write the bf16 window in place (csrc/seg_head.hip)
    fused = FUSED_HEAD[0] and x.dtype != F32 and bool(L.iic_seg_head_supported(C, k))
    Fm = None
    if fused:
      check(L.iic_seg_head_fwd(ptr(x), ptr(W2), ptr(logits), N, Hw, Ww, Hp, Wp, off, C, k, s), "iic_seg_head_fwd")
    else:
      Fm = torch.empty((M, C), dtype=F32, device=x.device)
      if x.dtype == F32:      # exact-fp32 parity path
        check(L.iic_f32_window_gather(ptr(x), ptr(Fm), N, Hw, Ww, Hp, Wp, off, C, s), "iic_f32_window_gather")
      else:
        check(L.iic_seg_window_gather(ptr(x), ptr(Fm), N, Hw, Ww, Hp, Wp, off, C, s), "iic_seg_window_gather")
      ops.gemm_f32(Fm, C, 1, W2, 1, C, logits, k, M, k, C)
    probs = ops.softmax_fwd(logits, M, k)
    out = torch.empty((N, k, S, S), dtype=F32, device=x.device)
    check(L.iic_bilinear_fwd(ptr(probs), ptr(out), N, Hw, Ww, k, S, s), "iic_bilinear_fwd")
    ctx.save_for_backward(x if fused else Fm, W2, probs)
    ctx.fused = fused
    ctx.meta = (tuple(x.shape), P, S, Hw, Ww, off, k)
    ctx.branch, ctx.pt_dtype = ops.BRANCH[0], x.dtype
    return out

  @ops.branch_backward
  def backward(ctx, dout):
    Fm, W2, probs = ctx.saved_tensors
    shape, P, S, Hw, Ww, off, k = ctx.meta
    N, Hp, Wp, C = shape
    M = N * Hw * Ww
    L, s = lib(), stream_ptr()
    dprobs = torch.empty((M, k), dtype=F32, device=dout.device)
    check(L.iic_bilinear_bwd(ptr(dout.contiguous()), ptr(dprobs), N, Hw, Ww, k, S, s), "iic_bilinear_bwd")
    dlog = ops.softmax_bwd(probs, dprobs, M, k)
    dx = ops.POOL.alloc(shape, dout.device, P)
    if ctx.fused:
      x = Fm                                      # (the PT feature tensor itself was saved)
      nch = L.iic_seg_head_wgrad_chunks(M)
      part = torch.empty((nch, k * C), dtype=F32, device=dout.device)
      check(L.iic_seg_head_wgrad(ptr(dlog), ptr(x), ptr(part), N, Hw, Ww, Hp, Wp, off, C, k, s), "iic_seg_head_wgrad")
      dW = torch.empty((k, C), dtype=F32, device=dout.device)
      check(L.iic_colsum_f32(ptr(part), ptr(dW), nch, k * C, 0, s), "iic_colsum_f32")
      check(L.iic_seg_head_bwd_dx(ptr(dlog), ptr(W2), ptr(dx), N, Hw, Ww, Hp, Wp, off, C, k, s), "iic_seg_head_bwd_dx")
      return dx, dW.view(k, C, 1, 1), None, None
    dW = torch.zeros((k, C), dtype=F32, device=dout.device)
    splitk = max(1, min(512, M // 2048))
    check(L.iic_gemm_f32_splitk(ptr(dlog), 1, k, ptr(Fm), C, 1, ptr(dW), C, k, C, M, splitk, s),
          "iic_gemm_f32_splitk")
    dF = torch.empty((M, C), dtype=F32, device=dout.device)
    ops.gemm_f32(dlog, k, 1, W2, C, 1, dF, C, M, C, k)
    if dx.dtype == F32:
      check(L.iic_f32_window_scatter(ptr(dF), ptr(dx), N, Hw, Ww, Hp, Wp, off, C, s), "iic_f32_window_scatter")
    else:
      check(L.iic_seg_window_scatter(ptr(dF), ptr(dx), N, Hw, Ww, Hp, Wp, off, C, s), "iic_seg_window_scatter")
    return dx, dW.view(k, C, 1, 1), None, None


class SegmentationNet10aTrunk(VGGTrunkHIP):
  P = 3

  def __init__(self, config, cfg):
    super(SegmentationNet10aTrunk, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    assert config.input_sz % 2 == 0
    self.conv_size, self.pad = 3, 1
    self.cfg = cfg
    self.in_channels = config.in_channels if hasattr(config, "in_channels") else 3
    self.features = self._make_layers(self.cfg, self.in_channels, self.conv_size, self.pad)
    self._build_stages()

  def forward(self, x):
    return self.run_stages(x)   # PT bf16, not flattened


class SegmentationNet10aHead(nn.Module):
  def __init__(self, config, output_k, cfg):
    super(SegmentationNet10aHead, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.cfg = cfg
    num_features = self.cfg[-1][0]
    self.num_sub_heads = config.num_sub_heads
    self.heads = nn.ModuleList([nn.Sequential(
      nn.Conv2d(num_features, output_k, kernel_size=1, stride=1, dilation=1, padding=1, bias=False),
      nn.Softmax2d()) for _ in range(self.num_sub_heads)])
    self.input_sz = config.input_sz

  def forward(self, x):
    return [_SegHeadFn.apply(x, ops.pv(self.heads[i][0].weight), SegmentationNet10aTrunk.P, self.input_sz)
            for i in range(self.num_sub_heads)]


class SegmentationNet10a(_ApplyCounter, nn.Module):
  cfg = [(64, 1), (128, 1), ("M", None), (256, 1), (256, 1), (512, 2), (512, 2)]

  def __init__(self, config):
    super(SegmentationNet10a, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.trunk = SegmentationNet10aTrunk(config, cfg=SegmentationNet10a.cfg)
    self.head = SegmentationNet10aHead(config, output_k=config.output_k, cfg=SegmentationNet10a.cfg)
    _initialize_weights_vgg(self)

  @ops.auto_branch
  def forward(self, x):
    return self.head(self.trunk(x))


class SegmentationNet10aTwoHead(_ApplyCounter, nn.Module):
  """net10a_twohead.py:8-31."""
  cfg = SegmentationNet10a.cfg

  def __init__(self, config):
    super(SegmentationNet10aTwoHead, self).__init__()
    self.batchnorm_track = config.batchnorm_track
    self.trunk = SegmentationNet10aTrunk(config, cfg=SegmentationNet10a.cfg)
    self.head_A = SegmentationNet10aHead(config, output_k=config.output_k_A, cfg=SegmentationNet10a.cfg)
    self.head_B = SegmentationNet10aHead(config, output_k=config.output_k_B, cfg=SegmentationNet10a.cfg)
    _initialize_weights_vgg(self)

  @ops.auto_branch
  def forward(self, x, head="B"):
    x = self.trunk(x)
    if head == "A":
      return self.head_A(x)
    elif head == "B":
      return self.head_B(x)
    raise AssertionError("head must be A or B")
