"""Reference points on the same GPU: (a) hipBLASLt/rocBLAS bf16 GEMM at the implicit-GEMM
shapes, (b) MIOpen conv2d (torch, bf16 channels_last) fwd / bwd for the ClusterNet5g layers.
Not part of the product path -- a yardstick for the hand-written kernels."""
import sys, os, time
import torch
import torch.nn.functional as F

def timeit(fn, iters=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / iters

dev = torch.device("cuda:0")
N = 660
print("GEMM yardstick (torch.matmul bf16):")
for name, M, Nn, K in [("l1", N*49*49, 64, 576), ("l2", N*25*25, 128, 1152), ("l3", N*13*13, 256, 2304), ("l4", N*7*7, 512, 4608)]:
  a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
  b = torch.randn(K, Nn, device=dev, dtype=torch.bfloat16)
  t = timeit(lambda: a @ b)
  print("  %s M=%d N=%d K=%d: %.1f us  %.0f TF/s" % (name, M, Nn, K, t, 2.0*M*Nn*K/t/1e6))
print("MIOpen conv2d bf16 channels_last (fwd, bwd-data+bwd-weight):")
for name, cin, cout, H, s in [("l1", 64, 64, 49, 1), ("l2", 128, 128, 25, 1), ("l3", 256, 256, 13, 1), ("l4", 512, 512, 7, 1), ("l3.0s2", 128, 256, 25, 2)]:
  x = torch.randn(N, cin, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
  w = (torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
  y = F.conv2d(x, w, stride=s, padding=1)
  dy = torch.randn_like(y)
  fl = 2.0 * y.numel() * cin * 9
  tf = timeit(lambda: F.conv2d(x, w, stride=s, padding=1))
  def bwd():
    yy = F.conv2d(x, w, stride=s, padding=1)
    yy.backward(dy)
    x.grad = None; w.grad = None
  tb = timeit(bwd) - tf
  print("  %s: fwd %.1f us (%.0f TF/s), bwd(data+weight) %.1f us (%.0f TF/s)" % (name, tf, fl/tf/1e6, tb, 2*fl/tb/1e6))
