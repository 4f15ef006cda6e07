"""ROCm HIP-graph probes that the design decisions in iic_amd/graph.py, iic_amd/graphed.py and DESIGN.md §5 cite,
one sub-command each (formerly seven separate scripts; bodies unchanged).  GPU only.

  python tools/graph_probes.py <probe> [args...]

  graph_probe        eager vs HIP-graph replay of the north-star step: ms/step and host ms  [env PAIRS]
  graph_debug        capturable-Adam / replay-overlap debugging of CapturedStep: NaN report per parameter  [env PAIRS]
  graph_debug2       forward / forward+backward capture: stale-accumulator check across replays  [env PAIRS]
  graph_sync_probe   host wait after a replay per way of waiting (device / stream / event sync, .item())
  dual_branch_probe  potential of running the two views as two concurrent graph branches (timing only)  [env PAIRS]
  graph_cut_probe    cut-graph step with the collectives replaced by a host round trip  [pairs] [host|none]
  graphed_probe      capture of one view's backward through torch.autograd.grad  <mode>
"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def graph_probe():
  """eager vs HIP-graph replay of the north-star step: ms/step and host ms  [env PAIRS]"""
  import torch
  from bench import make_batch
  from iic_amd import archs
  from iic_amd.graph import CapturedStep
  from iic_amd.losses import IID_loss_heads
  from iic_amd.optim import Adam
  from iic_amd.transforms import sobel_process

  dev = torch.device("cuda:0")
  pairs = int(os.environ.get("PAIRS", "660"))


  def build(capturable):
    torch.manual_seed(0)
    cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
    net = archs.ClusterNet5g(cfg).to(dev).train()
    opt = Adam(net.parameters(), lr=1e-4, capturable=capturable)
    imgs, imgs_tf = make_batch(pairs, 96, dev, seed=0)

    def step():
      net.zero_grad(set_to_none=True)
      xo = net.forward_packed(sobel_process(imgs, False))
      xt = net.forward_packed(sobel_process(imgs_tf, False))
      loss, _ = IID_loss_heads(xo, xt, lamb=1.0)
      loss = loss.mean()
      loss.backward()
      opt.step()
      return loss
    return net, opt, step


  def timeit(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      out = fn()
    te = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return 1e3 * dt / n, 1e3 * te / n, out


  net, opt, step = build(False)
  for _ in range(3):
    step()
  ms, enq, l = timeit(step)
  print("eager            : %.2f ms/step, host enqueue %.2f ms, loss %.6f" % (ms, enq, float(l)))
  losses_e = [float(step()) for _ in range(3)]

  net2, opt2, step2 = build(True)
  t0 = time.perf_counter()
  cs = CapturedStep(step2, warmup=3)
  print("capture took %.1f s" % (time.perf_counter() - t0))
  ms, enq, l = timeit(cs)
  print("graph replay     : %.2f ms/step, host launch %.2f ms, loss %.6f" % (ms, enq, float(l)))
  losses_g = [float(cs()) for _ in range(3)]
  print("eager losses after 13 steps:", losses_e)
  print("graph losses after 13 steps:", losses_g)
  opt2._sync_steps_to_host()
  print("graph step counter:", sorted(set(st["step"] for st in opt2.state.values())))
  # eager use after replay still works (weights epoch)
  ms, enq, l = timeit(step2, 3)
  print("eager after graph: %.2f ms/step, loss %.6f" % (ms, float(l)))


def graph_debug():
  """capturable-Adam / replay-overlap debugging of CapturedStep: NaN report per parameter  [env PAIRS]"""
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


def graph_debug2():
  """forward / forward+backward capture: stale-accumulator check across replays  [env PAIRS]"""
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


def graph_sync_probe():
  """host wait after a replay per way of waiting (device / stream / event sync, .item())"""
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


def dual_branch_probe():
  """potential of running the two views as two concurrent graph branches (timing only)  [env PAIRS]"""
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


def graph_cut_probe():
  """cut-graph step with the collectives replaced by a host round trip  [pairs] [host|none]"""
  import torch
  import torch.distributed as tdist
  from iic_amd import archs, ops, dist as idist
  from iic_amd.graph import CapturedPairStep
  from iic_amd.losses import IID_loss_heads
  from iic_amd.optim import Adam
  from iic_amd.transforms import sobel_process

  pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 66
  mode = sys.argv[2] if len(sys.argv) > 2 else "host"
  dev = torch.device("cuda:0")
  torch.manual_seed(0)
  cfg = types.SimpleNamespace(in_channels=2, input_sz=96, batchnorm_track=True, num_sub_heads=5, output_k=70)
  net = archs.ClusterNet5g(cfg).to(dev).train()
  opt = Adam(net.parameters(), lr=1e-4, capturable=True)
  params = list(net.parameters())
  imgs = torch.rand(pairs, 1, 96, 96, device=dev)
  imgs_tf = torch.rand(pairs, 1, 96, 96, device=dev)

  idist.enabled = lambda: True           # pretend: every all_reduce_sum_ becomes a cut


  def fake_all_reduce(t, op=None, group=None):
    if mode == "host":
      h = t.cpu()                        # D2H + stream sync
      t.copy_(h.to(t.device))
    else:
      t.mul_(1.0)                        # device-only stand-in (what RCCL looks like to the host)


  tdist.all_reduce = fake_all_reduce
  idist.dist.all_reduce = fake_all_reduce


  def loss_fn(a, b):
    l, _ = IID_loss_heads(a, b, lamb=1.0)
    return l.mean()


  def finish():
    ops.fold_branch_grads(params)
    idist.all_reduce_grads(params)
    opt.step()


  run = CapturedPairStep(lambda: net.forward_packed(sobel_process(imgs, False)),
                         lambda: net.forward_packed(sobel_process(imgs_tf, False)),
                         loss_fn, finish, lambda: net.zero_grad(set_to_none=True), warmup=1)
  print("segments: loss %d graphs + %d cuts, optimiser %d graphs + %d cuts" % (
    len(run.g_l.items) - run.g_l.cuts, run.g_l.cuts, len(run.g_opt.items) - run.g_opt.cuts, run.g_opt.cuts))
  torch.cuda.synchronize()
  for i in range(6):
    t0 = time.perf_counter()
    out = run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("step %d: enqueue %.2f ms, +sync %.2f ms, loss %.3e" % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t0), float(out)))


def graphed_probe():
  """capture of one view's backward through torch.autograd.grad  <mode>"""
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


PROBES = {"graph_probe": graph_probe, "graph_debug": graph_debug, "graph_debug2": graph_debug2, "graph_sync_probe": graph_sync_probe, "dual_branch_probe": dual_branch_probe, "graph_cut_probe": graph_cut_probe, "graphed_probe": graphed_probe}


if __name__ == "__main__":
  if len(sys.argv) < 2 or sys.argv[1] not in PROBES:
    sys.exit(__doc__)
  probe = PROBES[sys.argv[1]]
  sys.argv = [sys.argv[0] + " " + sys.argv[1]] + sys.argv[2:]     # the bodies read their own argv[1:]
  probe()
