import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_batch
from iic_amd import archs
from iic_amd.losses import IID_loss_heads
from iic_amd.optim import Adam
from iic_amd.transforms import sobel_process
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
dev = torch.device("cuda:0")
pairs = int(os.environ.get("PAIRS", "240"))
torch.manual_seed(0)
cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
net = archs.ClusterNet5g(cfg).to(dev).train()
# make the loss non-trivial: larger head weights
with torch.no_grad():
  for h in net.head.heads:
    h[0].weight.normal_(0, 0.3)
imgs, imgs_tf = make_batch(pairs, 96, dev, seed=0)

def fwd():
  xo = net.forward_packed(sobel_process(imgs, False))
  xt = net.forward_packed(sobel_process(imgs_tf, False))
  loss, _ = IID_loss_heads(xo, xt, lamb=1.0)
  return loss.mean()

def fwdbwd():
  net.zero_grad(set_to_none=True)
  l = fwd()
  l.backward()
  return l.detach()

def gradvec():
  return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()

ref = []
for _ in range(3):
  l = fwdbwd(); ref.append((float(l), gradvec()))
print("eager losses", [r[0] for r in ref])
print("eager grad rel diff run-to-run", float((ref[1][1]-ref[0][1]).norm()/ref[0][1].norm()))

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
  fwdbwd()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
# E1: forward only under no_grad
g1 = torch.cuda.CUDAGraph()
with torch.no_grad():
  with torch.cuda.stream(s):
    fwd()
  torch.cuda.synchronize()
  with torch.cuda.graph(g1):
    o1 = fwd()
for i in range(3):
  g1.replay(); torch.cuda.synchronize(); print("E1 fwd-only replay loss", float(o1))
# E2: fwd + bwd
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
  o2 = fwdbwd()
for i in range(3):
  g2.replay(); torch.cuda.synchronize()
  gv = gradvec()
  print("E2 fwd+bwd replay loss", float(o2), "grad rel diff vs eager", float((gv-ref[0][1]).norm()/ref[0][1].norm()))
# per-parameter worst
g2.replay(); torch.cuda.synchronize()
off = 0; worst = []
gv = gradvec()
for n, p in net.named_parameters():
  k = p.numel(); a = gv[off:off+k]; b = ref[0][1][off:off+k]; off += k
  worst.append((float((a-b).norm()/(b.norm()+1e-30)), n))
worst.sort(reverse=True)
print("worst params:", worst[:8])
