"""Potential of running the two views as two concurrent graph branches (timing only: shared
scratch buffers race, results are not checked)."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_batch
from iic_amd import archs
from iic_amd.transforms import sobel_process
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
dev = torch.device("cuda:0")
pairs = int(os.environ.get("PAIRS", "660"))
cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
torch.manual_seed(0)
netA = archs.ClusterNet5g(cfg).to(dev).train()
netB = archs.ClusterNet5g(cfg).to(dev).train()
imgs, imgs_tf = make_batch(pairs, 96, dev, seed=0)

def fb(net, x):
  net.zero_grad(set_to_none=True)
  p = net.forward_packed(sobel_process(x, False))
  (p * p).sum().backward()

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
s1.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s1):
  for _ in range(2):
    fb(netA, imgs); fb(netB, imgs_tf)
torch.cuda.synchronize()

def timed(g, n=5):
  ts = []
  for _ in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
  return sorted(ts)[len(ts) // 2]

g_seq = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_seq, stream=s1):
  fb(netA, imgs); fb(netB, imgs_tf)
print("sequential (one branch): %.2f ms" % timed(g_seq))

# warm netB on s2 so its AccumulateGrad nodes / pool buffers live there
s2.wait_stream(s1)
with torch.cuda.stream(s2):
  fb(netB, imgs_tf)
torch.cuda.synchronize()
g_par = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_par, stream=s1):
  s2.wait_stream(s1)
  fb(netA, imgs)
  with torch.cuda.stream(s2):
    fb(netB, imgs_tf)
  s1.wait_stream(s2)
print("two branches            : %.2f ms" % timed(g_par))
