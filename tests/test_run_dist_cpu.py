"""World-size-2 (gloo, CPU) run of the WHOLE hook chain `python -m iic_amd.run` sets up for the
unchanged scripts under torchrun (VERDICT r1 item 8 / ADVICE r1 #1): seeded construction ->
training forward shards the full batch each rank was handed -> raw-joint all-reduce inside the
loss -> SUM gradient all-reduce hooked onto optimiser.step -> evaluation forward sees the whole
batch.  The network and the loss are tiny CPU stand-ins that call the PRODUCT's helpers
(iic_amd.dist.shard_batch / all_reduce_sum_, iic_amd.run.setup_distributed); the result must
equal a single-process run on the full batch."""
import os
import socket
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

B, D, K, HEADS = 12, 6, 5, 2


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


class TinyNet(torch.nn.Module):
  """Stand-in with the architectures' contract: forward(x) -> list of per-sub-head softmax
  tensors, rows sharded by the product's helper in training forwards."""

  def __init__(self, config):
    super(TinyNet, self).__init__()
    self.heads = torch.nn.ModuleList([torch.nn.Linear(D, K) for _ in range(HEADS)]).double()

  def forward(self, x):
    from iic_amd import dist as idist
    x = idist.shard_batch(x, self)
    return [torch.softmax(h(x), dim=1) for h in self.heads]


class _LossFn(torch.autograd.Function):
  """Two-phase loss exactly as iic_amd.losses._IIDLossFn stages it: local raw joint ->
  all-reduce (SUM) -> loss / dLoss/dR of the GLOBAL joint -> gradients of the local rows."""

  @staticmethod
  def forward(ctx, z, zt):
    from iic_amd import dist as idist
    from oracle import iid_oracle
    R = torch.from_numpy(iid_oracle.raw_joint_np(z.detach().numpy(), zt.detach().numpy()))
    idist.all_reduce_sum_(R)
    loss, _, dR = iid_oracle.loss_and_grad_from_raw_np(R.numpy(), 1.0)
    ctx.save_for_backward(z, zt, torch.from_numpy(dR))
    return torch.tensor(loss, dtype=torch.float64)

  @staticmethod
  def backward(ctx, g):
    z, zt, dR = ctx.saved_tensors
    return g * (zt @ dR.T), g * (z @ dR)


def _train(net, x, xt, steps=2):
  opt = torch.optim.Adam(net.parameters(), lr=0.05)
  for _ in range(steps):
    net.zero_grad()
    a, b = net(x), net(xt)
    loss = sum(_LossFn.apply(a[i], b[i]) for i in range(HEADS)) / HEADS
    loss.backward()
    opt.step()
  return float(loss)


def _data():
  g = torch.Generator().manual_seed(5)
  x = torch.randn(B, D, generator=g, dtype=torch.float64)
  return x, x + 0.1 * torch.randn(B, D, generator=g, dtype=torch.float64)


def _worker(rank, world, port, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  os.environ["IIC_INIT_SEED"] = "11"
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from iic_amd import dist as idist
  from iic_amd import run
  fake = types.ModuleType("code.archs")       # what install() leaves behind: arch names bound
  fake.ClusterNet5g = TinyNet
  sys.modules["code.archs"] = fake
  handle = run.setup_distributed(ref_modules=("code.archs",))
  torch.manual_seed(100 + rank)               # the scripts seed nothing: ranks differ here
  net = sys.modules["code.archs"].ClusterNet5g(None).train()
  x, xt = _data()                             # every rank holds the FULL batch
  n_train_rows = net(x)[0].size(0)
  loss = _train(net, x, xt)
  net.eval()
  with torch.no_grad():
    n_eval_rows = net(x)[0].size(0)           # cluster_eval._clustering_get_data needs all rows
  net.train()
  with torch.no_grad():
    n_nograd_rows = net(x)[0].size(0)
  flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
  q.put((rank, n_train_rows, n_eval_rows, n_nograd_rows, loss, flat.numpy(), torch.save is not run.torch_save_orig))
  handle.remove()
  idist.SHARD_INPUTS[0] = False
  idist.disable()
  dist.destroy_process_group()


def test_run_hook_chain_world2_matches_single_process():
  from iic_amd import run
  world, port = 2, _free_port()
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in ps:
    p.start()
  res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
  for p in ps:
    p.join(60)
  # single process, full batch, same seeded construction
  with torch.random.fork_rng(devices=[]):
    torch.manual_seed(11)
    ref = TinyNet(None).train()
  x, xt = _data()
  ref_loss = _train(ref, x, xt)
  ref_flat = torch.cat([p.detach().reshape(-1) for p in ref.parameters()]).numpy()
  for rank, n_train, n_eval, n_nograd, loss, flat, save_patched in res:
    assert n_train == B // world            # training forward: this rank's pairs only
    assert n_eval == B and n_nograd == B    # eval / no_grad forwards: the whole batch
    assert abs(loss - ref_loss) < 1e-12     # every rank evaluates the GLOBAL loss
    assert np.abs(flat - ref_flat).max() < 1e-10, np.abs(flat - ref_flat).max()
    assert save_patched == (rank != 0)      # one checkpoint writer
  assert np.array_equal(res[0][5], res[1][5])


def test_per_rank_out_root_rewrites_only_ranks_above_zero():
  """The unchanged scripts write config.pickle / config.txt / figures with plain open() on every rank
  (cluster_sobel.py:117-124): under torchrun ranks > 0 are pointed at <out_root>/.rank<r>."""
  from iic_amd.run import per_rank_out_root
  argv = ["code.scripts.cluster.cluster_sobel", "--out_root", "/x/y", "--model_ind", "3"]
  assert per_rank_out_root(argv, 0) == (argv, "/x/y", "/x/y")
  got, root0, mine = per_rank_out_root(argv, 2)
  assert got == argv[:2] + ["/x/y/.rank2"] + argv[3:] and root0 == "/x/y" and mine == "/x/y/.rank2"
  assert per_rank_out_root(["m", "--out_root=/x/y"], 1)[0] == ["m", "--out_root=/x/y/.rank1"]
  assert per_rank_out_root(["m", "--model_ind", "3"], 1) == (["m", "--model_ind", "3"], None, None)
