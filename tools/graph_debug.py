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
pairs = int(os.environ.get("PAIRS", "240"))

def build(capturable, lr=1e-3):
  torch.manual_seed(0)
  cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
  net = archs.ClusterNet5g(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=lr, capturable=capturable)
  imgs, imgs_tf = make_batch(pairs, 96, dev, seed=0)
  def step():
    net.zero_grad(set_to_none=True)
    xo = net.forward_packed(sobel_process(imgs, False))
    xt = net.forward_packed(sobel_process(imgs_tf, False))
    loss, _ = IID_loss_heads(xo, xt, lamb=1.0)
    loss = loss.mean()
    loss.backward()
    opt.step()
    return loss.detach()
  return net, opt, step

def nanreport(net, tag):
  bad = [n for n, p in net.named_parameters() if not torch.isfinite(p).all()]
  badg = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
  print(tag, "nan params:", bad[:5], len(bad), "nan grads:", badg[:5], len(badg))

# A: eager, host-step Adam vs capturable Adam (same stream) -> same losses?
for cap in (False, True):
  net, opt, step = build(cap)
  print("eager capturable=%s:" % cap, [float(step()) for _ in range(6)])
  nanreport(net, " ")
# B: eager on a side stream
net, opt, step = build(True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
  l = [step() for _ in range(3)]
torch.cuda.current_stream().wait_stream(s)
print("side-stream eager:", [float(x) for x in l]); nanreport(net, " ")
# C: capture
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
  out = step()
nanreport(net, "after capture")
for i in range(4):
  g.replay(); torch.cuda.synchronize()
  print("replay", i, float(out)); nanreport(net, " ")
# launch cost of one replay on an idle GPU
for i in range(3):
  torch.cuda.synchronize(); t0 = time.perf_counter(); c0 = time.thread_time(); g.replay(); t1 = time.perf_counter(); c1 = time.thread_time()
  torch.cuda.synchronize(); t2 = time.perf_counter()
  print("idle-GPU replay: launch call %.2f ms wall / %.2f ms cpu, total %.2f ms" % (1e3*(t1-t0), 1e3*(c1-c0), 1e3*(t2-t0)))
# D: back-to-back replays without sync
for i in range(6):
  g.replay()
torch.cuda.synchronize()
print("after 6 back-to-back replays:", float(out)); nanreport(net, " ")
