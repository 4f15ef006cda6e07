"""ROCm 7.0 probe: how long does the host wait after a HIP-graph replay, per way of waiting?
(hipDeviceSynchronize vs stream synchronize vs event synchronize vs .item() on a graph output)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
x = torch.randn(4096, 4096, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
  for _ in range(2):
    y = (x @ x).sum()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
  y = (x @ x).sum()


def timed(name, wait):
  ts = []
  for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
      g.replay()
      wait()
    ts.append(1e3 * (time.perf_counter() - t0))
  print("%-34s %s ms" % (name, " ".join("%.2f" % t for t in ts)))


timed("torch.cuda.synchronize()", torch.cuda.synchronize)
timed("stream.synchronize()", lambda: s.synchronize())
def ev():
  e = torch.cuda.Event(); e.record(s); e.synchronize()
timed("event.synchronize()", ev)
timed("y.item()", lambda: y.item())
timed("y.cpu()", lambda: y.cpu())
def evq():
  e = torch.cuda.Event(); e.record(s)
  while not e.query():
    pass
timed("event.query() spin", evq)
