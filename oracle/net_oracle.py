"""CPU restatement (plain torch functional ops) of the reference architectures on
the IIC hot path.  TEST INFRASTRUCTURE ONLY.

Follows:
  * sobel_process           /root/reference/code/utils/cluster/transforms.py:47-96
  * ClusterNet5g            /root/reference/code/archs/cluster/net5g.py:10-103
      BasicBlock / _make_layer / init   .../residual.py:10-85
  * ClusterNet6c            /root/reference/code/archs/cluster/net6c.py:10-88, vgg.py:8-54
  * SegmentationNet10a      /root/reference/code/archs/segmentation/net10a.py:13-80
  * train step              /root/reference/code/scripts/cluster/cluster_sobel.py:235-272

All functions take a flat ``params`` dict whose keys are the reference's
``state_dict`` keys (``trunk.conv1.weight`` ...), so the same dict drives the
imported reference module (gen_golden.py), this oracle and the HIP modules.

Weights are generated with numpy's Generator (bit-stable across machines) using the
reference's init *distributions* (kaiming-normal fan_out / fan_in, BN 1/0, Linear
N(0, 0.01)) -- residual.py:75-85, vgg.py:42-54.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------
# sobel
# ----------------------------------------------------------------------------

def sobel_process(imgs, include_rgb, using_IR=False):
  """transforms.py:47-96 (device-agnostic)."""
  bn, c, h, w = imgs.size()
  if not using_IR:
    if not include_rgb:
      assert c == 1
      grey_imgs = imgs
    else:
      assert c == 4
      grey_imgs = imgs[:, 3, :, :].unsqueeze(1)
      rgb_imgs = imgs[:, :3, :, :]
  else:
    if not include_rgb:
      assert c == 2
      grey_imgs = imgs[:, 0, :, :].unsqueeze(1)
      ir_imgs = imgs[:, 1, :, :].unsqueeze(1)
    else:
      assert c == 5
      rgb_imgs = imgs[:, :3, :, :]
      grey_imgs = imgs[:, 3, :, :].unsqueeze(1)
      ir_imgs = imgs[:, 4, :, :].unsqueeze(1)
  s1 = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]], dtype=imgs.dtype,
                    device=imgs.device).view(1, 1, 3, 3)
  s2 = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]], dtype=imgs.dtype,
                    device=imgs.device).view(1, 1, 3, 3)
  dx = F.conv2d(grey_imgs, s1, padding=1)
  dy = F.conv2d(grey_imgs, s2, padding=1)
  sobel_imgs = torch.cat([dx, dy], dim=1)
  if not using_IR:
    if include_rgb:
      sobel_imgs = torch.cat([rgb_imgs, sobel_imgs], dim=1)
  else:
    if include_rgb:
      sobel_imgs = torch.cat([rgb_imgs, sobel_imgs, ir_imgs], dim=1)
    else:
      sobel_imgs = torch.cat([sobel_imgs, ir_imgs], dim=1)
  return sobel_imgs


# ----------------------------------------------------------------------------
# parameter construction (reference key names / shapes / init distributions)
# ----------------------------------------------------------------------------

def _kaiming(rng, shape, mode):
  co, ci, kh, kw = shape
  fan = (co if mode == "fan_out" else ci) * kh * kw
  std = math.sqrt(2.0 / fan)
  return torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32))


def _bn(params, prefix, c, track):
  params[prefix + ".weight"] = torch.ones(c)
  params[prefix + ".bias"] = torch.zeros(c)
  if track:
    params[prefix + ".running_mean"] = torch.zeros(c)
    params[prefix + ".running_var"] = torch.ones(c)
    params[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


NET5G_LAYERS = [3, 4, 6, 3]
NET5G_PLANES = [64, 128, 256, 512]


def make_net5g_params(in_channels=2, output_k=70, num_sub_heads=5, batchnorm_track=True,
                      seed=0, heads=("head",), output_ks=None, randomize_bn=False,
                      head_std=0.01):
  """Keys/shapes of ClusterNet5g.state_dict() (net5g.py:10-103); ``heads`` =
  ("head_A","head_B") with ``output_ks`` for the TwoHead variant."""
  rng = np.random.default_rng(seed)
  p = {}
  p["trunk.conv1.weight"] = _kaiming(rng, (64, in_channels, 3, 3), "fan_out")
  _bn(p, "trunk.bn1", 64, batchnorm_track)
  inplanes = 64
  for li, (planes, nblk) in enumerate(zip(NET5G_PLANES, NET5G_LAYERS)):
    for b in range(nblk):
      pre = "trunk.layer%d.%d" % (li + 1, b)
      stride = 2 if (b == 0 and li > 0) else 1
      p[pre + ".conv1.weight"] = _kaiming(rng, (planes, inplanes, 3, 3), "fan_out")
      _bn(p, pre + ".bn1", planes, batchnorm_track)
      p[pre + ".conv2.weight"] = _kaiming(rng, (planes, planes, 3, 3), "fan_out")
      _bn(p, pre + ".bn2", planes, batchnorm_track)
      if stride != 1 or inplanes != planes:
        p[pre + ".downsample.0.weight"] = _kaiming(rng, (planes, inplanes, 1, 1), "fan_out")
        _bn(p, pre + ".downsample.1", planes, batchnorm_track)
      inplanes = planes
  ks = output_ks if output_ks is not None else [output_k] * len(heads)
  for hname, k in zip(heads, ks):
    for i in range(num_sub_heads):
      p["%s.heads.%d.0.weight" % (hname, i)] = torch.from_numpy(
        (rng.standard_normal((k, 512)) * head_std).astype(np.float32))
      p["%s.heads.%d.0.bias" % (hname, i)] = torch.zeros(k)
  if randomize_bn:
    _randomize_bn(p, rng)
  return p


def _randomize_bn(p, rng):
  """Non-trivial gamma/beta so parity tests exercise the affine part."""
  for key in list(p.keys()):
    if key.endswith(".running_mean") or key.endswith(".running_var") or \
       key.endswith("num_batches_tracked"):
      continue
    t = p[key]
    if t.dim() == 1 and ("bn" in key or "downsample.1" in key or "features" in key):
      if key.endswith(".weight"):
        p[key] = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(t.shape)).astype(np.float32))
      elif key.endswith(".bias"):
        p[key] = torch.from_numpy((0.1 * rng.standard_normal(t.shape)).astype(np.float32))


def vgg_feature_index(cfg):
  """Index of each conv / bn inside VGGTrunk.features (vgg.py:8-35)."""
  idx, out = 0, []
  for c, dil in cfg:
    if c in ("M", "A"):
      out.append(("pool", idx, c, None))
      idx += 1
    else:
      out.append(("conv", idx, c, dil))
      idx += 3  # conv, bn, relu
  return out


NET6C_CFG = [(64, 1), ("M", None), (128, 1), ("M", None), (256, 1), ("M", None), (512, 1)]
NET10A_CFG = [(64, 1), (128, 1), ("M", None), (256, 1), (256, 1), (512, 2), (512, 2)]


def make_vgg_params(cfg, conv_size, in_channels, batchnorm_track, rng):
  p = {}
  cin = in_channels
  for kind, idx, c, dil in vgg_feature_index(cfg):
    if kind == "conv":
      p["trunk.features.%d.weight" % idx] = _kaiming(rng, (c, cin, conv_size, conv_size), "fan_in")
      _bn(p, "trunk.features.%d" % (idx + 1), c, batchnorm_track)
      cin = c
  return p


def make_net6c_params(in_channels=1, input_sz=24, output_k=10, num_sub_heads=5,
                      batchnorm_track=True, seed=0, heads=("head",), output_ks=None,
                      randomize_bn=False, head_std=0.01):
  rng = np.random.default_rng(seed)
  p = make_vgg_params(NET6C_CFG, 5, in_channels, batchnorm_track, rng)
  sp = {24: 3, 64: 8}[input_sz]
  ks = output_ks if output_ks is not None else [output_k] * len(heads)
  for hname, k in zip(heads, ks):
    for i in range(num_sub_heads):
      p["%s.heads.%d.0.weight" % (hname, i)] = torch.from_numpy(
        (rng.standard_normal((k, 512 * sp * sp)) * head_std).astype(np.float32))
      p["%s.heads.%d.0.bias" % (hname, i)] = torch.zeros(k)
  if randomize_bn:
    _randomize_bn(p, rng)
  return p


def make_net10a_params(in_channels=4, output_k=3, num_sub_heads=1, batchnorm_track=True,
                       seed=0, heads=("head",), output_ks=None, randomize_bn=False):
  rng = np.random.default_rng(seed)
  p = make_vgg_params(NET10A_CFG, 3, in_channels, batchnorm_track, rng)
  ks = output_ks if output_ks is not None else [output_k] * len(heads)
  for hname, k in zip(heads, ks):
    for i in range(num_sub_heads):
      p["%s.heads.%d.0.weight" % (hname, i)] = _kaiming(rng, (k, 512, 1, 1), "fan_in")
  if randomize_bn:
    _randomize_bn(p, rng)
  return p


# ----------------------------------------------------------------------------
# functional forwards (training-mode BN uses batch stats; updates running stats
# in ``params`` in place exactly like nn.BatchNorm2d, momentum 0.1, unbiased var)
# ----------------------------------------------------------------------------

def _batchnorm(x, params, prefix, training):
  w, b = params[prefix + ".weight"], params[prefix + ".bias"]
  rm = params.get(prefix + ".running_mean")
  rv = params.get(prefix + ".running_var")
  use_batch = training or rm is None  # track_running_stats=False => batch stats in eval
  if rm is not None and training:
    params[prefix + ".num_batches_tracked"] += 1
  return F.batch_norm(x, rm, rv, w, b, use_batch, BN_MOMENTUM, BN_EPS)


def net5g_trunk(params, x, training=True, input_sz=96, penultimate_features=False):
  """net5g.py:41-58."""
  x = F.conv2d(x, params["trunk.conv1.weight"], stride=1, padding=1)
  x = _batchnorm(x, params, "trunk.bn1", training)
  x = F.relu(x)
  x = F.max_pool2d(x, kernel_size=2, stride=2, padding=1)
  for li, nblk in enumerate(NET5G_LAYERS):
    if penultimate_features and li == 3:
      break
    for b in range(nblk):
      pre = "trunk.layer%d.%d" % (li + 1, b)
      stride = 2 if (b == 0 and li > 0) else 1
      residual = x
      out = F.conv2d(x, params[pre + ".conv1.weight"], stride=stride, padding=1)
      out = _batchnorm(out, params, pre + ".bn1", training)
      out = F.relu(out)
      out = F.conv2d(out, params[pre + ".conv2.weight"], stride=1, padding=1)
      out = _batchnorm(out, params, pre + ".bn2", training)
      if (pre + ".downsample.0.weight") in params:
        residual = F.conv2d(x, params[pre + ".downsample.0.weight"], stride=stride)
        residual = _batchnorm(residual, params, pre + ".downsample.1", training)
      x = F.relu(out + residual)
  if not penultimate_features:
    x = F.avg_pool2d(x, {96: 7, 64: 5, 32: 3}[input_sz], stride=1)
  return x.view(x.size(0), -1)


def heads_forward(params, feats, head="head", num_sub_heads=5):
  """net5g.py:73-80 / net6c.py:52-59: Linear + Softmax(dim=1) per sub-head."""
  return [F.softmax(F.linear(feats, params["%s.heads.%d.0.weight" % (head, i)],
                             params["%s.heads.%d.0.bias" % (head, i)]), dim=1)
          for i in range(num_sub_heads)]


def net5g_forward(params, x, training=True, input_sz=96, head="head", num_sub_heads=5):
  return heads_forward(params, net5g_trunk(params, x, training, input_sz), head, num_sub_heads)


def vgg_trunk(params, x, cfg, conv_size, pad, training=True):
  """vgg.py:8-35."""
  for kind, idx, c, dil in vgg_feature_index(cfg):
    if kind == "pool":
      x = F.max_pool2d(x, 2, 2) if c == "M" else F.avg_pool2d(x, 2, 2)
    else:
      x = F.conv2d(x, params["trunk.features.%d.weight" % idx], stride=1, padding=pad,
                   dilation=dil)
      x = _batchnorm(x, params, "trunk.features.%d" % (idx + 1), training)
      x = F.relu(x)
  return x


def net6c_forward(params, x, training=True, head="head", num_sub_heads=5):
  """net6c.py:22-26, 52-59, 76-88."""
  f = vgg_trunk(params, x, NET6C_CFG, 5, 2, training)
  return heads_forward(params, f.view(f.size(0), -1), head, num_sub_heads)


def net10a_forward(params, x, input_sz, training=True, head="head", num_sub_heads=1):
  """net10a.py:29-31, 52-59, 77-80: 1x1 conv pad 1 + Softmax2d + bilinear upsample."""
  f = vgg_trunk(params, x, NET10A_CFG, 3, 1, training)
  outs = []
  for i in range(num_sub_heads):
    y = F.conv2d(f, params["%s.heads.%d.0.weight" % (head, i)], padding=1)
    y = F.softmax(y, dim=1)
    y = F.interpolate(y, size=input_sz, mode="bilinear", align_corners=False)
    outs.append(y)
  return outs


# ----------------------------------------------------------------------------
# synthetic batches and the reference train step
# ----------------------------------------------------------------------------

def make_paired_batch(n_pairs, input_sz=96, num_dataloaders=3, seed=0):
  """all_imgs / all_imgs_tf in [0,1], fp32 [n_pairs,1,S,S]: smoothed-noise base
  images replicated num_dataloaders x; tf = flip + brightness jitter + noise."""
  assert n_pairs % num_dataloaders == 0
  nb = n_pairs // num_dataloaders
  rng = np.random.default_rng(seed)
  base = rng.random((nb, 1, input_sz, input_sz)).astype(np.float32)
  base = torch.from_numpy(base)
  base = F.avg_pool2d(F.pad(base, (2, 2, 2, 2), mode="replicate"), 5, stride=1)
  all_imgs = base.repeat(num_dataloaders, 1, 1, 1)
  rng1 = np.random.default_rng(seed + 1)
  gain = torch.from_numpy(rng1.uniform(0.6, 1.4, (n_pairs, 1, 1, 1)).astype(np.float32))
  noise = torch.from_numpy((rng1.standard_normal(all_imgs.shape) * 0.05).astype(np.float32))
  all_imgs_tf = torch.clamp(torch.flip(all_imgs, dims=[3]) * gain + noise, 0.0, 1.0)
  return all_imgs.contiguous(), all_imgs_tf.contiguous()


def make_mild_pair(n_pairs, input_sz=64, num_dataloaders=3, seed=0):
  """Like make_paired_batch, but the second view is a MILD transform of the first (gain in [0.9, 1.1],
  noise 0.01, no flip): a randomly initialised trunk is not flip-invariant, so only a mild transform
  leaves the two views' features correlated -- the whole-net fixtures that need a loss well away from
  MI = 0 (tests/golden/net5g_large.npz) use this pair."""
  all_imgs, _ = make_paired_batch(n_pairs, input_sz, num_dataloaders, seed)
  rng1 = np.random.default_rng(seed + 1)
  gain = torch.from_numpy(rng1.uniform(0.9, 1.1, (n_pairs, 1, 1, 1)).astype(np.float32))
  noise = torch.from_numpy((rng1.standard_normal(all_imgs.shape) * 0.01).astype(np.float32))
  return all_imgs.contiguous(), torch.clamp(all_imgs * gain + noise, 0.0, 1.0).contiguous()


def net5g_train_step_loss(params, all_imgs, all_imgs_tf, lamb=1.0, input_sz=96,
                          num_sub_heads=5, head="head"):
  """cluster_sobel.py:235-253: sobel x2, two train-mode forwards, mean IID_loss."""
  from .iid_oracle import IID_loss
  a = sobel_process(all_imgs, False)
  b = sobel_process(all_imgs_tf, False)
  x_outs = net5g_forward(params, a, True, input_sz, head, num_sub_heads)
  x_tf_outs = net5g_forward(params, b, True, input_sz, head, num_sub_heads)
  tot, tot_nl = None, None
  for i in range(num_sub_heads):
    l, lnl = IID_loss(x_outs[i], x_tf_outs[i], lamb=lamb)
    tot = l if tot is None else tot + l
    tot_nl = lnl if tot_nl is None else tot_nl + lnl
  return tot / num_sub_heads, tot_nl / num_sub_heads, x_outs, x_tf_outs


# ----------------------------------------------------------------------------
# bf16-storage emulation of the HIP pipeline (parity tier T3)
# ----------------------------------------------------------------------------
# The HIP path stores activations in bf16 and feeds bf16 operands to the MFMA convs while
# accumulating / taking BatchNorm statistics in fp32.  This restatement applies the SAME
# rounding points to the fp32 oracle (straight-through for autograd), so that what remains
# between it and the GPU result is accumulation order only.  It documents, next to the pure
# fp32 oracle, how much of a difference is inherent to bf16 storage.

def _rbf(t):
  """Round to bf16 (value), identity gradient."""
  return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def _bn_emu(y, params, prefix, training, round_y=True):
  """BatchNorm whose statistics come from the unrounded fp32 conv output and whose affine
  map is applied to the bf16-stored copy (what conv epilogue + bn_apply do)."""
  w, b = params[prefix + ".weight"], params[prefix + ".bias"]
  rm, rv = params.get(prefix + ".running_mean"), params.get(prefix + ".running_var")
  if training or rm is None:
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    if rm is not None:
      with torch.no_grad():
        n = y.numel() / y.size(1)
        rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
        rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
        params[prefix + ".num_batches_tracked"] += 1
  else:
    mean, var = rm, rv
  scale = w / torch.sqrt(var + BN_EPS)
  shift = b - mean * scale
  ys = _rbf(y) if round_y else y
  return ys * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def block_bf16emu(params, pre, xcur, stride, training=True):
  """One BasicBlock (residual.py:29-43) with the HIP path's rounding points; xcur is the
  bf16-representable block input (NCHW float)."""
  wq = lambda k: _rbf(params[k])
  y1 = F.conv2d(xcur, wq(pre + ".conv1.weight"), stride=stride, padding=1)
  a1 = _rbf(F.relu(_bn_emu(y1, params, pre + ".bn1", training)))
  y2 = F.conv2d(a1, wq(pre + ".conv2.weight"), stride=1, padding=1)
  o = _bn_emu(y2, params, pre + ".bn2", training)
  if (pre + ".downsample.0.weight") in params:
    yd = F.conv2d(xcur, wq(pre + ".downsample.0.weight"), stride=stride)
    o = o + _bn_emu(yd, params, pre + ".downsample.1", training)
  else:
    o = o + xcur
  return _rbf(F.relu(o))


def net5g_forward_bf16emu(params, x, training=True, input_sz=96, head="head", num_sub_heads=5):
  """ClusterNet5g forward with the HIP path's rounding points."""
  y = F.conv2d(x, params["trunk.conv1.weight"], stride=1, padding=1)        # stem: exact fp32
  a = F.relu(_bn_emu(y, params, "trunk.bn1", training, round_y=False))
  xcur = _rbf(F.max_pool2d(a, kernel_size=2, stride=2, padding=1))          # stored bf16
  for li, nblk in enumerate(NET5G_LAYERS):
    for bidx in range(nblk):
      pre = "trunk.layer%d.%d" % (li + 1, bidx)
      stride = 2 if (bidx == 0 and li > 0) else 1
      xcur = block_bf16emu(params, pre, xcur, stride, training)
  feats = F.avg_pool2d(xcur, {96: 7, 64: 5, 32: 3}[input_sz], stride=1).view(xcur.size(0), -1)
  return heads_forward(params, feats, head, num_sub_heads)


def vgg_stage_bf16emu(params, idx, x, pad, dil, pool, first, training=True):
  """One VGG stage (conv -> BN -> ReLU [-> MaxPool2]) with the HIP path's rounding points.
  First stage: exact fp32 conv of the image; later stages: bf16 operands."""
  w = params["trunk.features.%d.weight" % idx]
  y = F.conv2d(x, w if first else _rbf(w), stride=1, padding=pad, dilation=dil)
  a = _rbf(F.relu(_bn_emu(y, params, "trunk.features.%d" % (idx + 1), training)))
  return F.max_pool2d(a, 2, 2) if pool else a


def net6c_forward_bf16emu(params, x, training=True, head="head", num_sub_heads=5):
  layers = vgg_feature_index(NET6C_CFG)
  first = True
  for li, (kind, idx, c, dil) in enumerate(layers):
    if kind != "conv":
      continue
    pool = li + 1 < len(layers) and layers[li + 1][0] == "pool"
    x = vgg_stage_bf16emu(params, idx, x, 2, dil, pool, first, training)
    first = False
  return heads_forward(params, x.reshape(x.size(0), -1), head, num_sub_heads)


def net10a_forward_bf16emu(params, x, input_sz, training=True, head="head", num_sub_heads=1):
  """SegmentationNet10a with the HIP path's rounding points (trunk bf16 storage, fp32 head)."""
  layers = vgg_feature_index(NET10A_CFG)
  first = True
  for li, (kind, idx, c, dil) in enumerate(layers):
    if kind != "conv":
      continue
    pool = li + 1 < len(layers) and layers[li + 1][0] == "pool"
    x = vgg_stage_bf16emu(params, idx, x, 1, dil, pool, first, training)
    first = False
  outs = []
  for i in range(num_sub_heads):
    y = F.conv2d(x, params["%s.heads.%d.0.weight" % (head, i)], padding=1)
    y = F.softmax(y, dim=1)
    outs.append(F.interpolate(y, size=input_sz, mode="bilinear", align_corners=False))
  return outs
