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
