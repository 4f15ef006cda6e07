"""Single-process probe of the cut-graph step: the collectives are replaced by a host round trip
(D2H copy, stream sync, H2D copy) -- what gloo does -- to see whether the eager calls between graph
segments stall.  python tools/graph_cut_probe.py [pairs]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as tdist
from iic_amd import archs, ops, dist as idist
from iic_amd.graph import CapturedPairStep
from iic_amd.losses import IID_loss_heads
from iic_amd.optim import Adam
from iic_amd.transforms import sobel_process

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 66
mode = sys.argv[2] if len(sys.argv) > 2 else "host"
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
net = archs.ClusterNet5g(cfg).to(dev).train()
opt = Adam(net.parameters(), lr=1e-4, capturable=True)
params = list(net.parameters())
imgs = torch.rand(pairs, 1, 96, 96, device=dev)
imgs_tf = torch.rand(pairs, 1, 96, 96, device=dev)

idist.enabled = lambda: True           # pretend: every all_reduce_sum_ becomes a cut


def fake_all_reduce(t, op=None, group=None):
  if mode == "host":
    h = t.cpu()                        # D2H + stream sync
    t.copy_(h.to(t.device))
  else:
    t.mul_(1.0)                        # device-only stand-in (what RCCL looks like to the host)


tdist.all_reduce = fake_all_reduce
idist.dist.all_reduce = fake_all_reduce


def loss_fn(a, b):
  l, _ = IID_loss_heads(a, b, lamb=1.0)
  return l.mean()


def finish():
  ops.fold_branch_grads(params)
  idist.all_reduce_grads(params)
  opt.step()


run = CapturedPairStep(lambda: net.forward_packed(sobel_process(imgs, False)),
                       lambda: net.forward_packed(sobel_process(imgs_tf, False)),
                       loss_fn, finish, lambda: net.zero_grad(set_to_none=True), warmup=1)
print("segments: loss %d graphs + %d cuts, optimiser %d graphs + %d cuts" % (
  len(run.g_l.items) - run.g_l.cuts, run.g_l.cuts, len(run.g_opt.items) - run.g_opt.cuts, run.g_opt.cuts))
torch.cuda.synchronize()
for i in range(6):
  t0 = time.perf_counter()
  out = run()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print("step %d: enqueue %.2f ms, +sync %.2f ms, loss %.3e" % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t0), float(out)))
