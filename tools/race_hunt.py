"""Which tensors differ when a two-stream mode of the drop-in path (ops.auto_branch) disagrees with the one-stream run?
("find the 1-in-13" -- found with this tool in round 5: profiles/r05_race_hunt.txt.)

Every run starts from IDENTICAL state (parameters, BatchNorm buffers, inputs); the reference is the one-stream eager run;
each run of the mode under test is compared with it bit for bit -- the views' outputs, the losses, parameter gradients,
parameters, running statistics -- and the first tensors that differ are named.

  python tools/race_hunt.py [runs=200] [n=48] [mode=graph2|graph1|eager2]
    graph2: graph replay + two streams (what `python -m iic_amd.run` does), graph1: graph replay on one stream,
    eager2: auto_branch with eager launches (they stay on the caller's stream since round 5)
  HUNT_STEPS=4      several steps per run with an optimiser (HUNT_OPT=torch|ours), no synchronisation but the loss's .item()
  HUNT_SYNC=1       torch.cuda.synchronize() after every step;  HUNT_SYNC_AT=D|d|m|e: one synchronisation point only
  HUNT_TRACE=1      asynchronous clones of outputs / gradients / parameters per step: WHICH step goes wrong first
"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from iic_amd import archs, ops
from iic_amd.losses import IID_loss
from iic_amd.transforms import sobel_process
from oracle import net_oracle      # (a measurement tool: synthetic parameters / batch from the test fixtures)


def build(sz=32, k=10, heads=2, seed=0):
  cfg = types.SimpleNamespace(in_channels=2, input_sz=sz, batchnorm_track=True, num_sub_heads=heads, output_k=k)
  params = net_oracle.make_net5g_params(2, k, heads, True, seed=seed, randomize_bn=True, head_std=0.3)
  net = archs.ClusterNet5g(cfg)
  net.load_state_dict(params, strict=True)
  return net.cuda().train(), params


def one_run(net, state, imgs, imgs_tf, heads):
  net.load_state_dict(state, strict=True)
  ops_epoch_bump()
  net.zero_grad(set_to_none=True)
  xo = net(sobel_process(imgs, False))
  xt = net(sobel_process(imgs_tf, False))
  tot = None
  for i in range(heads):
    l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
    tot = l if tot is None else tot + l
  tot = tot / heads
  tot.backward()
  torch.cuda.synchronize()
  out = {"loss": tot.detach().clone()}
  for i in range(heads):
    out["xo%d" % i] = xo[i].detach().clone()
    out["xt%d" % i] = xt[i].detach().clone()
  for n_, p in net.named_parameters():
    out["grad:" + n_] = None if p.grad is None else p.grad.detach().clone()
  for n_, b in net.named_buffers():
    out["buf:" + n_] = b.detach().clone()
  return out


def multi_run(net, state, imgs, imgs_tf, heads, steps, sync_each, opt_kind):
  """The unchanged scripts' sequence over several steps (tests/test_gpu_graph.py
  test_auto_branch_reference_call_sequence_is_bit_identical): no synchronisation but the loss's .item()."""
  net.load_state_dict(state, strict=True)
  ops_epoch_bump()
  if opt_kind == "torch":
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
  else:
    from iic_amd.optim import Adam
    opt = Adam(net.parameters(), lr=2e-4)
  out = {}
  for s_ in range(steps):
    net.zero_grad()
    xo = net(sobel_process(imgs, False))
    xt = net(sobel_process(imgs_tf, False))
    tot = None
    for i in range(heads):
      l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
      tot = l if tot is None else tot + l
    tot = tot / heads
    if os.environ.get("HUNT_TRACE", "0") == "1":
      out["s%d:xo0" % s_] = xo[0].detach().clone()
      out["s%d:xt0" % s_] = xt[0].detach().clone()
    out["loss%d" % s_] = torch.tensor(tot.item())
    tot.backward()
    at = os.environ.get("HUNT_SYNC_AT", "")
    if "D" in at:
      torch.cuda.synchronize()                      # after backward, before the optimiser
    if "d" in at:                                   # only the side streams
      for st_ in list(ops._BRANCH_STREAM.values()):
        st_.synchronize()
    if s_ == steps - 1:
      for n_, p in net.named_parameters():
        out["lastgrad:" + n_] = None if p.grad is None else p.grad.detach().clone()
    if os.environ.get("HUNT_TRACE", "0") == "1":    # asynchronous clones on the caller's stream: no host wait added
      for n_, p in net.named_parameters():
        if n_ in ("trunk.conv1.weight", "trunk.layer1.0.conv1.weight", "trunk.layer4.2.conv2.weight", "head.heads.0.0.weight"):
          out["s%d:grad:%s" % (s_, n_)] = None if p.grad is None else p.grad.detach().clone()
    opt.step()
    if os.environ.get("HUNT_TRACE", "0") == "1":
      for n_, p in net.named_parameters():
        if n_ in ("trunk.conv1.weight", "trunk.layer1.0.conv1.weight", "trunk.layer4.2.conv2.weight", "head.heads.0.0.weight"):
          out["s%d:param:%s" % (s_, n_)] = p.detach().clone()
    if "m" in at:                                   # only the caller's stream
      torch.cuda.current_stream().synchronize()
    if "e" in at:
      for st_ in list(ops._BRANCH_STREAM.values()):
        st_.synchronize()
    if sync_each:
      torch.cuda.synchronize()
  torch.cuda.synchronize()
  for n_, p in net.named_parameters():
    out["param:" + n_] = p.detach().clone()
  for n_, b in net.named_buffers():
    out["buf:" + n_] = b.detach().clone()
  return out


def ops_epoch_bump():
  from iic_amd.archs.cluster import bump_weights_epoch
  bump_weights_epoch()        # load_state_dict wrote the parameters in place: the bf16 operands must be re-laid


def main():
  runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
  mode = sys.argv[3] if len(sys.argv) > 3 else "eager2"
  heads = 2
  net, _ = build(heads=heads)
  state = {k_: v.detach().clone() for k_, v in net.state_dict().items()}
  a, b = net_oracle.make_paired_batch(n, 32, 3, seed=5)
  imgs, imgs_tf = a.cuda(), b.cuda()
  ops.AUTO_BRANCH[0] = False
  ops.GRAPH_FORWARD[0] = False
  steps = int(os.environ.get("HUNT_STEPS", "0"))
  sync_each = os.environ.get("HUNT_SYNC", "0") == "1"
  opt_kind = os.environ.get("HUNT_OPT", "torch")
  if steps:
    run = lambda: multi_run(net, state, imgs, imgs_tf, heads, steps, sync_each, opt_kind)
  else:
    run = lambda: one_run(net, state, imgs, imgs_tf, heads)
  ref = run()
  ref2 = run()
  same = all((ref[k_] is None and ref2[k_] is None) or torch.equal(ref[k_], ref2[k_]) for k_ in ref)
  print("one-stream run reproduces itself: %s" % same)
  ops.AUTO_BRANCH[0] = mode != "graph1"
  ops.GRAPH_FORWARD[0] = mode in ("graph1", "graph2")
  bad = 0
  hist = {}
  for r in range(runs):
    got = run()
    diff = []
    for k_ in ref:
      x, y = ref[k_], got[k_]
      if (x is None) != (y is None) or (x is not None and not torch.equal(x, y)):
        d = float("nan") if x is None or y is None else (x.float() - y.float()).abs().max().item()
        diff.append((k_, d))
    if diff:
      bad += 1
      names = [k_ for k_, _ in diff]
      key = ",".join(names[:3]) + ("...(%d)" % len(names) if len(names) > 3 else "")
      hist[key] = hist.get(key, 0) + 1
      if bad <= 6:
        if os.environ.get("HUNT_TRACE", "0") == "1":
          order = sorted([kd for kd in diff if kd[0].startswith("s")], key=lambda kd: (int(kd[0][1]), {"x": 0, "g": 1, "p": 2}[kd[0].split(":")[1][0]]))
          print("run %d: trace of differing tensors in step order: %s" % (r, ", ".join("%s (%.1e)" % kd for kd in order[:10])))
        else:
          print("run %d: %d tensors differ; first: %s" % (r, len(diff), ", ".join("%s (%.2e)" % kd for kd in diff[:8])))
  print("mode %s, n = %d, steps %d (sync each %s, optimiser %s): %d of %d runs differ from the one-stream run" % (
    mode, n, steps, sync_each, opt_kind, bad, runs))
  for k_, c in sorted(hist.items(), key=lambda kv: -kv[1])[:8]:
    print("  %4d x  %s" % (c, k_))


if __name__ == "__main__":
  main()
