"""One-off probe (ROCm graph capture of a view's backward through torch.autograd.grad)."""
import sys, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iic_amd import archs, ops
from iic_amd.archs import cluster as cl
from iic_amd.transforms import sobel_process
dev = torch.device("cuda:0")
mode = sys.argv[1]
cfg = types.SimpleNamespace(in_channels=2, input_sz=32, batchnorm_track=True, num_sub_heads=2, output_k=10)
net = archs.ClusterNet5g(cfg).to(dev).train()
x = sobel_process(torch.rand(24, 1, 32, 32, device=dev), False)
params = [p for p in net.parameters()]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
  for _ in range(2):
    out = net(x)
    torch.autograd.grad(out, params, [torch.ones_like(o) for o in out], allow_unused=True)
torch.cuda.synchronize()
print("warm ok", flush=True)
cap = s if "samestream" in mode else torch.cuda.Stream()
pool = torch.cuda.graph_pool_handle()
sx = x.clone()
cl.bump_weights_epoch()
gf = torch.cuda.CUDAGraph()
with torch.cuda.graph(gf, pool=pool, stream=cap):
  out = net(sx)
print("fwd captured", flush=True)
gouts = [torch.zeros_like(o) for o in out]
gb = torch.cuda.CUDAGraph()
with torch.cuda.graph(gb, pool=pool, stream=cap):
  if "backward" in mode:
    torch.autograd.backward(out, gouts)
  else:
    grads = torch.autograd.grad(out, params, gouts, allow_unused=True)
print("bwd captured", flush=True)
gf.replay(); gb.replay(); torch.cuda.synchronize()
print("replayed ok", mode)
