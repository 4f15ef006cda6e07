"""Eager vs HIP-graph replay of the north-star step: ms/step and host ms."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_batch
from iic_amd import archs
from iic_amd.graph import CapturedStep
from iic_amd.losses import IID_loss_heads
from iic_amd.optim import Adam
from iic_amd.transforms import sobel_process

dev = torch.device("cuda:0")
pairs = int(os.environ.get("PAIRS", "660"))


def build(capturable):
  torch.manual_seed(0)
  cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
  net = archs.ClusterNet5g(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=1e-4, capturable=capturable)
  imgs, imgs_tf = make_batch(pairs, 96, dev, seed=0)

  def step():
    net.zero_grad(set_to_none=True)
    xo = net.forward_packed(sobel_process(imgs, False))
    xt = net.forward_packed(sobel_process(imgs_tf, False))
    loss, _ = IID_loss_heads(xo, xt, lamb=1.0)
    loss = loss.mean()
    loss.backward()
    opt.step()
    return loss
  return net, opt, step


def timeit(fn, n=10):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    out = fn()
  te = time.perf_counter() - t0
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  return 1e3 * dt / n, 1e3 * te / n, out


net, opt, step = build(False)
for _ in range(3):
  step()
ms, enq, l = timeit(step)
print("eager            : %.2f ms/step, host enqueue %.2f ms, loss %.6f" % (ms, enq, float(l)))
losses_e = [float(step()) for _ in range(3)]

net2, opt2, step2 = build(True)
t0 = time.perf_counter()
cs = CapturedStep(step2, warmup=3)
print("capture took %.1f s" % (time.perf_counter() - t0))
ms, enq, l = timeit(cs)
print("graph replay     : %.2f ms/step, host launch %.2f ms, loss %.6f" % (ms, enq, float(l)))
losses_g = [float(cs()) for _ in range(3)]
print("eager losses after 13 steps:", losses_e)
print("graph losses after 13 steps:", losses_g)
opt2._sync_steps_to_host()
print("graph step counter:", sorted(set(st["step"] for st in opt2.state.values())))
# eager use after replay still works (weights epoch)
ms, enq, l = timeit(step2, 3)
print("eager after graph: %.2f ms/step, loss %.6f" % (ms, float(l)))
