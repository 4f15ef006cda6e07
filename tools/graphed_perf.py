"""Drop-in path timing: the reference script's call sequence (cluster_sobel.py:235-272) at the north-star
shapes, eager vs graphed forwards (iic_amd/graphed.py).  python tools/graphed_perf.py [--steps 10] [--no-item]"""
import argparse, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iic_amd import archs, ops
from iic_amd.losses import IID_loss
from iic_amd.optim import Adam
from iic_amd.transforms import sobel_process

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--no-item", action="store_true")
ap.add_argument("--modes", default="eager1,graphed2,graphed1")
ap.add_argument("--profile", action="store_true", help="cProfile of the timed steps (host side), top 30 by cumulative time")
ap.add_argument("--user-stream", action="store_true", help="run the whole script-side sequence on a non-default stream")
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.user_stream:
  torch.cuda.set_stream(torch.cuda.Stream())
cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
imgs, imgs_tf = bench.make_batch(660, 96, dev)
for mode in a.modes.split(","):
  torch.manual_seed(0)
  net = archs.ClusterNet5g(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=1e-4)
  ops.AUTO_BRANCH[0] = mode.endswith("2")
  ops.GRAPH_FORWARD[0] = mode.startswith("graphed")
  t_host = [0.0]

  def step():
    h0 = time.perf_counter()
    net.zero_grad()
    xo = net(sobel_process(imgs, False))
    xt = net(sobel_process(imgs_tf, False))
    avg = None
    for i in range(5):
      l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
      avg = l if avg is None else avg + l
    avg = avg / 5
    t_host[0] += time.perf_counter() - h0
    v = 0.0 if a.no_item else avg.item()
    h1 = time.perf_counter()
    avg.backward()
    opt.step()
    t_host[0] += time.perf_counter() - h1
    return v
  for _ in range(4):
    step()
  torch.cuda.synchronize()
  t_host[0] = 0.0
  t0 = time.perf_counter()
  if a.profile:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
  for _ in range(a.steps):
    step()
  if a.profile:
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / a.steps
  print("%-10s %.2f ms/step (%.0f pairs/s), host time in python calls %.2f ms/step" % (mode, 1e3 * dt, 660 / dt, 1e3 * t_host[0] / a.steps))
  ops.join()
