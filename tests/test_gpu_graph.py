"""Captured-step machinery on the GPU: device-step Adam (capturable), HIP-graph replay of a whole
train step, and the two-view branch mode -- each against the plain eager single-stream path."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(seed=0, k=10, heads=2, sz=32):
  from iic_amd import archs
  from oracle import net_oracle
  cfg = types.SimpleNamespace(in_channels=2, input_sz=sz, batchnorm_track=True, num_sub_heads=heads, output_k=k)
  params = net_oracle.make_net5g_params(2, k, heads, True, seed=seed, randomize_bn=True, head_std=0.3)
  net = archs.ClusterNet5g(cfg)
  net.load_state_dict(params, strict=True)
  return net.cuda().train()


def _batch(n=48, sz=32, seed=5):
  from oracle import net_oracle
  a, b = net_oracle.make_paired_batch(n, sz, 3, seed=seed)
  return a.cuda(), b.cuda()


def _make_step(net, opt, imgs, imgs_tf, branch):
  from iic_amd import ops
  from iic_amd.losses import IID_loss_heads
  from iic_amd.transforms import sobel_process

  def step():
    net.zero_grad(set_to_none=True)
    if branch:
      with ops.branch():          # the first view on the side stream
        xo = net.forward_packed(sobel_process(imgs, False))
      xt = net.forward_packed(sobel_process(imgs_tf, False))
      ops.join()
    else:
      xo = net.forward_packed(sobel_process(imgs, False))
      xt = net.forward_packed(sobel_process(imgs_tf, False))
    loss, _ = IID_loss_heads(xo, xt, lamb=1.0)
    loss = loss.mean()
    loss.backward()
    opt.step()
    return loss.detach()
  return step


def test_adam_capturable_matches_torch_and_host_step_variant():
  from iic_amd.optim import Adam
  torch.manual_seed(0)
  shapes = [(64, 3, 3, 3), (64,), (128, 64, 3, 3), (7, 5)]
  ref = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
  a = [torch.nn.Parameter(p.detach().clone()) for p in ref]
  b = [torch.nn.Parameter(p.detach().clone()) for p in ref]
  o_ref = torch.optim.Adam(ref, lr=1e-2)
  o_a, o_b = Adam(a, lr=1e-2), Adam(b, lr=1e-2, capturable=True)
  for it in range(5):
    gs = [torch.randn(s, device="cuda") for s in shapes]
    for ps in (ref, a, b):
      for p, g in zip(ps, gs):
        p.grad = g.clone()
      if it == 2:
        ps[3].grad = None          # a tensor that skips a step keeps its own step count
    o_ref.step(); o_a.step(); o_b.step()
  for r, x, y in zip(ref, a, b):
    assert torch.allclose(r, x, rtol=1e-5, atol=1e-6)
    assert torch.equal(x, y)       # same arithmetic, step count on host vs device
  sd = o_b.state_dict()
  assert [sd["state"][i]["step"] for i in range(4)] == [5, 5, 5, 4]
  # torch.optim.Adam checkpoints (tensor step) load into ours, and ours into torch's
  o_c = Adam([torch.nn.Parameter(p.detach().clone()) for p in ref], lr=1e-2)
  o_c.load_state_dict(o_ref.state_dict())
  assert all(isinstance(st["step"], int) for st in o_c.state.values())
  torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ref], lr=1e-2).load_state_dict(o_a.state_dict())


def test_adam_second_gradient_source():
  from iic_amd import ops
  from iic_amd.optim import Adam
  torch.manual_seed(1)
  p1 = torch.nn.Parameter(torch.randn(1000, device="cuda"))
  p2 = torch.nn.Parameter(p1.detach().clone())
  g1, g2 = torch.randn(1000, device="cuda"), torch.randn(1000, device="cuda")
  o1, o2 = Adam([p1], lr=1e-2), Adam([p2], lr=1e-2)
  p1.grad = g1 + g2
  o1.step()
  ops.BRANCH[0] = 1
  try:
    q = ops.pv(p2)
  finally:
    ops.BRANCH[0] = 0
  assert q is not p2 and q.data_ptr() == p2.data_ptr()
  p2.grad, q.grad = g1.clone(), g2.clone()
  o2.step()
  assert torch.equal(p1, p2)
  ops.clear_branch_grads()


@pytest.mark.parametrize("mode", ["graph", "branch", "graph+branch", "pair"])
def test_captured_and_branched_steps_track_the_eager_step(mode):
  from iic_amd.graph import CapturedStep
  from iic_amd.optim import Adam
  imgs, imgs_tf = _batch()
  runs = {}
  for name in ("eager", mode):
    net = _net()
    graph, branch = "graph" in name, "branch" in name
    opt = Adam(net.parameters(), lr=2e-4, capturable=graph or name == "pair")
    step = _make_step(net, opt, imgs, imgs_tf, branch)
    if name == "pair":       # five linear graphs on two streams
      from iic_amd.graph import CapturedPairStep
      from iic_amd.losses import IID_loss_heads
      from iic_amd.transforms import sobel_process
      run = CapturedPairStep(lambda: net.forward_packed(sobel_process(imgs, False)),
                             lambda: net.forward_packed(sobel_process(imgs_tf, False)),
                             lambda a, b: IID_loss_heads(a, b, lamb=1.0)[0].mean(), opt.step,
                             lambda: net.zero_grad(set_to_none=True), warmup=2)
    else:
      run = CapturedStep(step, warmup=2) if graph else step
    if not graph and name != "pair":
      step(); step()
    losses = [float(run()) for _ in range(6)]
    torch.cuda.synchronize()
    nbt = int(net.trunk.bn1.num_batches_tracked)
    rm = net.trunk.layer3[2].bn2.running_mean.clone()
    runs[name] = (np.array(losses), nbt, rm, [p.detach().clone() for p in net.parameters()])
  le, lm = runs["eager"][0], runs[mode][0]
  assert np.all(np.isfinite(lm))
  # every accumulation on the path is order-independent (exact fixed-point BatchNorm statistics,
  # fixed-order split-K reductions): replaying the captured graph, and running the second view as
  # a concurrent branch, reproduce the eager single-stream step BIT FOR BIT
  assert np.array_equal(lm, le), (le, lm)
  assert runs[mode][1] == runs["eager"][1] == 2 * 8          # two statistic updates per step
  assert torch.equal(runs[mode][2], runs["eager"][2])
  for a, b in zip(runs[mode][3], runs["eager"][3]):
    assert torch.equal(a, b)


@pytest.mark.parametrize("mode", ["pair", "staged"])
def test_two_stream_graph_modes_with_k_split_head_gemm_are_bit_identical_to_eager(mode):
  """Batch large enough (96 x 512 features) for the heads' backward GEMM to take the LDS-tiled kernel with
  its K split over a workspace: the workspace is per branch, so the two views' backward graphs -- replayed
  concurrently on two streams -- must not share it (a shared one made view A's gradients differ from
  replay to replay).  `staged`: the backward captured per layer group (the data-parallel replay order,
  forced here without a process group).  Six steps enqueued back to back, no host synchronisation."""
  from iic_amd.graph import CapturedPairStep
  from iic_amd.losses import IID_loss_heads
  from iic_amd.optim import Adam
  from iic_amd.transforms import sobel_process
  imgs, imgs_tf = _batch(n=96)
  runs = {}
  for name in ("eager", mode):
    net = _net()
    opt = Adam(net.parameters(), lr=2e-4, capturable=name != "eager")
    loss_fn = lambda a, b: IID_loss_heads(a, b, lamb=1.0)[0].mean()     # noqa: E731
    if name == "eager":
      run = _make_step(net, opt, imgs, imgs_tf, False)
      run(); run()
    elif name == "pair":
      run = CapturedPairStep(lambda: net.forward_packed(sobel_process(imgs, False)),
                             lambda: net.forward_packed(sobel_process(imgs_tf, False)),
                             loss_fn, opt.step, lambda: net.zero_grad(set_to_none=True), warmup=2)
    else:
      events = []
      run = CapturedPairStep(lambda: net.forward_packed_taps(sobel_process(imgs, False)),
                             lambda: net.forward_packed_taps(sobel_process(imgs_tf, False)),
                             loss_fn, opt.step, lambda: net.zero_grad(set_to_none=True), warmup=2,
                             grad_groups=net.grad_groups(), opt_step=opt.step, events=events, force_staged=True)
      assert run.staged and len(run.g_ba) == 4 and len(run.buckets) == 4
    losses = [run().clone() for _ in range(6)]       # (a replay returns the same static tensor every time)
    torch.cuda.synchronize()
    runs[name] = ([float(l) for l in losses], [p.detach().clone() for p in net.parameters()])
    if name.startswith("staged"):
      assert events[:9] == [("bwd", 0), ("reduce", 0), ("bwd", 1), ("reduce", 1), ("bwd", 2), ("reduce", 2),
                            ("bwd", 3), ("reduce", 3), ("opt",)]
      # .grad of every parameter is a view into its group's flat bucket
      for grp, flat in zip(net.grad_groups(), run.buckets):
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        assert all(lo <= p.grad.data_ptr() < hi for p in grp if p.grad is not None)
  assert runs[mode][0] == runs["eager"][0], (runs[mode][0], runs["eager"][0])
  for a, b in zip(runs[mode][1], runs["eager"][1]):
    assert torch.equal(a, b)


def test_update_lr_between_replays_changes_the_captured_optimiser_step():
  """update_lr (/root/reference/code/utils/cluster/general.py:12-23) scales param_groups[i]["lr"] in place.
  The captured optimiser step reads its rate from device memory, pushed before every replay: a replayed
  run with a rate change in the middle must equal the eager run with the same change, bit for bit -- and
  differ from a replayed run without it.  The checkpoint stays torch.optim.Adam-compatible."""
  from iic_amd.graph import CapturedPairStep
  from iic_amd.losses import IID_loss_heads
  from iic_amd.optim import Adam
  from iic_amd.transforms import sobel_process
  imgs, imgs_tf = _batch()
  out = {}
  for name in ("eager", "pair", "pair-constant"):
    net = _net()
    opt = Adam(net.parameters(), lr=2e-4, capturable=name != "eager")
    if name == "eager":
      run = _make_step(net, opt, imgs, imgs_tf, False)
      run(); run()
    else:
      run = CapturedPairStep(lambda: net.forward_packed(sobel_process(imgs, False)),
                             lambda: net.forward_packed(sobel_process(imgs_tf, False)),
                             lambda a, b: IID_loss_heads(a, b, lamb=1.0)[0].mean(), opt.step,
                             lambda: net.zero_grad(set_to_none=True), warmup=2)
    for i in range(6):
      if i == 3 and name != "pair-constant":
        for g in opt.param_groups:          # what update_lr does (general.py:20-23)
          g["lr"] *= 0.1
      run()
    torch.cuda.synchronize()
    out[name] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    sd = opt.state_dict()
    assert all(set(g) == {"lr", "betas", "eps", "params"} for g in sd["param_groups"]), sd["param_groups"][0].keys()
  assert torch.equal(out["pair"], out["eager"])
  assert not torch.equal(out["pair"], out["pair-constant"])


def test_replay_back_to_back_equals_replay_with_syncs():
  from iic_amd.graph import CapturedStep
  from iic_amd.optim import Adam
  imgs, imgs_tf = _batch()
  out = []
  for sync in (True, False):
    net = _net()
    opt = Adam(net.parameters(), lr=2e-4, capturable=True)
    cs = CapturedStep(_make_step(net, opt, imgs, imgs_tf, True), warmup=2)
    for _ in range(5):
      l = cs()
      if sync:
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    out.append(float(l))
    assert all(torch.isfinite(p).all() for p in net.parameters())
  assert out[0] == out[1]


def test_training_is_bit_reproducible_run_to_run():
  """Same initial state, same batch, three independent runs of 4 eager steps: identical losses and
  parameters (round 1: BatchNorm statistics went through float atomics and the loss of this very
  fixture took several discrete values run to run)."""
  from iic_amd.optim import Adam
  imgs, imgs_tf = _batch()
  ref = None
  for _ in range(3):
    net = _net()
    opt = Adam(net.parameters(), lr=2e-4)
    step = _make_step(net, opt, imgs, imgs_tf, False)
    losses = [float(step()) for _ in range(4)]
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    if ref is None:
      ref = (losses, flat)
    else:
      assert losses == ref[0], (losses, ref[0])
      assert torch.equal(flat, ref[1])


def test_auto_branch_reference_call_sequence_is_bit_identical():
  """iic_amd.ops.auto_branch (what `python -m iic_amd.run` switches on) with EAGER launches: the unchanged scripts'
  call sequence -- net(x), net(x_tf), IID_loss per sub-head, stock torch.optim.Adam -- against the same sequence with
  the switch off: identical bits.  Eager forwards of the pair stay on the caller's stream (the side stream is for
  captured / replayed views: tests/test_gpu_graphed.py)."""
  from iic_amd import ops
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  imgs, imgs_tf = _batch()
  res = []
  for auto in (False, True):
    net = _net()
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    ops.AUTO_BRANCH[0] = auto
    try:
      losses = []
      for _ in range(4):
        net.zero_grad()
        xo = net(sobel_process(imgs, False))
        assert not ops._PENDING_JOIN
        assert ops._SOLO_FIRST[0] == (1 if auto else 0)
        xt = net(sobel_process(imgs_tf, False))
        tot = None
        for i in range(2):
          l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
          tot = l if tot is None else tot + l
        assert not ops._PENDING_JOIN and not ops._SOLO_FIRST[0]          # the loss joined
        tot /= 2
        losses.append(tot.item())
        tot.backward()
        opt.step()
      net.eval()
      with torch.no_grad():
        ev = net(sobel_process(imgs, False))[0].clone()     # evaluation never branches
      assert not ops._PENDING_JOIN
    finally:
      ops.AUTO_BRANCH[0] = False
    torch.cuda.synchronize()
    res.append((losses, ev, [p.detach().clone() for p in net.parameters()],
                net.trunk.bn1.running_mean.clone(), int(net.trunk.bn1.num_batches_tracked)))
  assert res[0][0] == res[1][0], (res[0][0], res[1][0])
  assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][3], res[1][3]) and res[0][4] == res[1][4] == 8
  assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))


def test_captured_pair_steps_share_one_stream_pair_that_really_overlaps():
  """HIP multiplexes streams onto a few hardware queues in creation order: two streams on one queue run their work one
  after the other (the second CapturedPairStep of a process used to end up there: LAB.md section R5.6).  Every captured pair
  step shares ONE pair per device, chosen by a concurrency probe; the probe itself can tell the difference -- a stream
  never runs beside itself."""
  from iic_amd.graph import _pair_streams, _streams_overlap
  a, b = _pair_streams(), _pair_streams()
  assert a[0] is b[0] and a[1] is b[1] and a[0] is not a[1]
  assert _streams_overlap(a[0], a[1])
  assert not _streams_overlap(a[0], a[0])
