"""Yardstick (NOT the product path): the reference-equivalent train step through stock
PyTorch-ROCm / MIOpen on the same MI355X -- the oracle restatement of ClusterNet5g + IID_loss
moved to the GPU, fp32 and bf16-autocast channels_last.  python oracle/torch_gpu_yardstick.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import net_oracle

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 660
params = net_oracle.make_net5g_params(2, 70, 5, True, seed=0)
params = {k: v.to(dev) for k, v in params.items()}
leaves = [v.requires_grad_(True) for k, v in params.items() if v.dtype.is_floating_point and "running" not in k]
opt = torch.optim.Adam(leaves, lr=1e-4)
imgs, imgs_tf = net_oracle.make_paired_batch(N, 96, 3, seed=0)
imgs, imgs_tf = imgs.to(dev), imgs_tf.to(dev)
for mode in ("fp32", "bf16-autocast"):
  def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode != "fp32")):
      loss, _, _, _ = net_oracle.net5g_train_step_loss(params, imgs, imgs_tf, 1.0, 96, 5)
    loss.backward()
    opt.step()
  for _ in range(2): step()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  K = 4
  for _ in range(K): step()
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
  print("stock PyTorch-ROCm %s: %.1f ms/step, %.0f pairs/s" % (mode, dt * 1e3, N / dt))
