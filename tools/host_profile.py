"""Host-side (Python + ctypes) cost of enqueueing one train step: cProfile over a few steps with
the GPU left to run ahead (no sync inside).  python tools/host_profile.py [--steps 5]"""
import argparse, cProfile, pstats, sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iic_amd import archs
from iic_amd.losses import IID_loss_heads
from iic_amd.optim import Adam
from iic_amd.transforms import sobel_process

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--pairs", type=int, default=660)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
net = archs.ClusterNet5g(cfg).to(dev).train()
opt = Adam(net.parameters(), lr=1e-4)
imgs, imgs_tf = bench.make_batch(a.pairs, 96, dev)


def step():
  net.zero_grad(set_to_none=True)
  xo = net.forward_packed(sobel_process(imgs, False))
  xt = net.forward_packed(sobel_process(imgs_tf, False))
  loss, _ = IID_loss_heads(xo, xt, lamb=1.0)
  loss.mean().backward()
  opt.step()


for _ in range(3):
  step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
  step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, wall %.2f ms/step" % (1e3 * (t1 - t0) / a.steps, 1e3 * (t2 - t0) / a.steps))
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
  step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
