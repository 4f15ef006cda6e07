"""World-size-2 (gloo, CPU) checks of the data-parallel path: the collectives in
iic_amd.dist and the sharded-joint algebra of SURVEY.md §8e (sum of per-rank raw joints ->
identical global loss; per-rank dz rows == rows of the full-batch gradient)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from iic_amd import dist as idist
  from oracle import iid_oracle
  idist.enable()
  assert idist.enabled() and idist.world_size() == world and idist.rank() == rank
  bn, k = 101, 7     # ragged split
  z, zt = iid_oracle.make_softmax_pair(bn, k, "trained", 3, np.float64)
  lo, hi = idist.shard_rows(bn)
  R = torch.from_numpy(iid_oracle.raw_joint_np(z[lo:hi], zt[lo:hi]))
  idist.all_reduce_sum_(R)
  loss, loss_nl, dR = iid_oracle.loss_and_grad_from_raw_np(R.numpy(), 1.5)
  dz_local = zt[lo:hi] @ dR.T
  full = iid_oracle.iid_loss_np(z, zt, 1.5)
  ok = abs(loss - full[0]) < 1e-12 and np.abs(dz_local - full[2][lo:hi]).max() < 1e-12
  # bucketed gradient all-reduce: SUM over ranks
  ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
  for i, p in enumerate(ps):
    p.grad = torch.full_like(p, float(rank + 1 + i))
  idist.all_reduce_grads(ps, bucket_bytes=32)
  tot = sum(range(1, world + 1))
  ok = ok and all(torch.allclose(p.grad, torch.full_like(p, float(tot + world * i))) for i, p in enumerate(ps))
  # overlapped reducer: hooks fire during backward, incomplete buckets (a parameter without a
  # gradient) are handled in finish(); result == plain SUM over ranks
  torch.manual_seed(0)
  ws = [torch.nn.Parameter(torch.randn(6, 4)), torch.nn.Parameter(torch.randn(4, 3)),
        torch.nn.Parameter(torch.randn(3)), torch.nn.Parameter(torch.randn(5))]     # last: never used
  red = idist.GradReducer(ws, bucket_bytes=64)
  for it in range(2):
    for w in ws:
      w.grad = None
    x = torch.full((2, 6), float(rank + 1 + it))
    y = ((x @ ws[0]) @ ws[1] + ws[2]).sum()
    y.backward()
    local = [w.grad.clone() if w.grad is not None else None for w in ws]
    red.finish()
    for w, l in zip(ws, local):
      if l is None:
        ok = ok and w.grad is None
        continue
      tot = l.clone()
      dist.all_reduce(tot)
      ok = ok and torch.allclose(w.grad, tot)
  red.remove()
  q.put((rank, bool(ok), lo, hi))
  dist.destroy_process_group()


def test_sharded_joint_and_grad_allreduce_world2():
  world = 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = [q.get(timeout=120) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
  res.sort()
  assert all(r[1] for r in res), res
  assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == 101


def test_shard_rows_cover_batch():
  from iic_amd import dist as idist
  for n in (660, 661, 7, 1):
    for w in (1, 2, 4, 8):
      spans = [idist.shard_rows(n, r, w) for r in range(w)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


# ---------------------------------------------------------------------------------------------
# Segmentation: the sharded per-shift joint (BASELINE configs[3] / [4] are multi-GPU by name).
# The reference's joint is a conv2d whose "batch" contraction runs over the samples
# (segmentation/IID_losses.py:125-126: x1 [k, bn, h, w] * weight x2 [k, bn, h, w] -> [k, k, 2T+1, 2T+1]),
# i.e. additive over samples exactly like the clustering joint: each rank contracts its own pairs,
# ONE all-reduce (SUM) of the (2T+1)^2 k^2 raw joints follows, every rank then evaluates the identical
# loss and back-propagates its own rows.  The unchanged scripts hand the loss FULL-batch masks and
# affine matrices beside sharded network outputs (segmentation_twohead.py:318-325): the product's
# shard_like slices them (iic_amd/seg_losses.py:165-166).
# ---------------------------------------------------------------------------------------------

def _seg_inputs(bn, k, h, w, seed):
  g = torch.Generator().manual_seed(seed)
  x1 = torch.softmax(2.0 * torch.randn(bn, k, h, w, generator=g, dtype=torch.float64), dim=1)
  x2 = torch.softmax(2.0 * torch.randn(bn, k, h, w, generator=g, dtype=torch.float64), dim=1)
  aff = torch.zeros(bn, 2, 3, dtype=torch.float64)
  aff[:, 0, 0] = torch.where(torch.rand(bn, generator=g) < 0.5, -1.0, 1.0).double()      # x-flip for about half (potsdam.py:189-202)
  aff[:, 1, 1] = 1.0
  mask = (torch.rand(bn, h, w, generator=g) < 0.7).double()
  return x1, x2, aff, mask


class _AllReduceSum(torch.autograd.Function):
  """SUM over ranks in forward; the upstream gradient unchanged in backward (d sum_r R_r / d R_local = 1)."""

  @staticmethod
  def forward(ctx, t):
    from iic_amd import dist as idist
    return idist.all_reduce_sum_(t.clone())

  @staticmethod
  def backward(ctx, g):
    return g


def _seg_worker(rank, world, port, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from iic_amd import dist as idist
  from oracle import iid_oracle
  idist.enable()
  idist.SHARD_INPUTS[0] = True
  bn, k, h, w = 5, 4, 9, 8        # ragged split: 3 + 2 pairs
  out = []
  for variant, fn in (("uncollapsed", iid_oracle.IID_segmentation_loss_uncollapsed), ("collapsed", iid_oracle.IID_segmentation_loss)):
    for T in (1, 2):
      x1, x2, aff, mask = _seg_inputs(bn, k, h, w, seed=7 + T)
      # full batch, one process (the oracle as it is)
      f1, f2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
      full, full_nl = fn(f1, f2, all_affine2_to_1=aff, all_mask_img1=mask, lamb=1.5, half_T_side_dense=T)
      full.backward()
      # this rank: its own rows of the two views, FULL-batch side inputs through the product's slicer, the raw joint
      # all-reduced before the oracle's own normalise / symmetrise / MI stage
      lo, hi = idist.shard_rows(bn)
      l1, l2 = x1[lo:hi].clone().requires_grad_(True), x2[lo:hi].clone().requires_grad_(True)
      aff_l, mask_l = idist.shard_like(aff, hi - lo), idist.shard_like(mask, hi - lo)
      assert aff_l.size(0) == hi - lo and mask_l.size(0) == hi - lo
      orig = iid_oracle._seg_joint
      iid_oracle._seg_joint = lambda *a, **kw: _AllReduceSum.apply(orig(*a, **kw))
      try:
        loc, loc_nl = fn(l1, l2, all_affine2_to_1=aff_l, all_mask_img1=mask_l, lamb=1.5, half_T_side_dense=T)
      finally:
        iid_oracle._seg_joint = orig
      loc.backward()
      out.append((variant, T, abs(float(loc) - float(full)), abs(float(loc_nl) - float(full_nl)),
                  float((l1.grad - f1.grad[lo:hi]).abs().max()), float((l2.grad - f2.grad[lo:hi]).abs().max()),
                  float(f1.grad.abs().max())))
  q.put((rank, out))
  idist.SHARD_INPUTS[0] = False
  idist.disable()
  dist.destroy_process_group()


def test_segmentation_sharded_joint_and_side_input_slicing_world2():
  world = 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_seg_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = [q.get(timeout=180) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
  assert len(res) == world
  for rank, out in res:
    assert len(out) == 4
    for variant, T, dl, dnl, d1, d2, gmax in out:
      assert dl < 1e-12 and dnl < 1e-12, (rank, variant, T, dl, dnl)
      assert gmax > 0 and d1 < 1e-12 * max(1.0, gmax) + 1e-13 and d2 < 1e-12 * max(1.0, gmax) + 1e-13, (rank, variant, T, d1, d2)


def test_forced_single_rank_group_keeps_collectives_on():
  """iic_amd.dist.enable(force=True): a one-rank group still issues its collectives (identities) -- what lets ONE
  MI355X execute the N > 1 path through RCCL (tests/test_gpu_rccl.py); the default keeps them off at world size 1."""
  from iic_amd import dist as idist
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(_free_port())
  dist.init_process_group("gloo", rank=0, world_size=1)
  try:
    idist.enable()
    assert not idist.enabled() and idist.world_size() == 1
    idist.enable(force=True)
    assert idist.enabled() and idist.world_size() == 1 and idist.rank() == 0 and idist.backend() == "gloo"
    n0 = idist.CALLS.get("all_reduce", 0)
    t = torch.arange(6.0)
    idist.all_reduce_sum_(t)
    assert torch.equal(t, torch.arange(6.0)) and idist.CALLS["all_reduce"] == n0 + 1
    assert idist.shard_rows(10) == (0, 10)
  finally:
    idist.disable()
    dist.destroy_process_group()
  assert not idist.enabled()
