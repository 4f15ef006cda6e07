"""Host-side bookkeeping that needs no GPU: the PT buffer pool's pin / unpin (deferred weight gradients, iic_amd/ops.py)
and the graph-replay planner of the drop-in path (iic_amd/graphed.py::plan: which occurrence of a training forward runs
eagerly, which one is captured, which ones replay; positions within a step; resource namespaces of one-stream runs)."""
import types

import torch

from iic_amd import graphed, ops


def test_pool_keeps_pinned_buffers_out_of_circulation_until_unpin():
  pool = ops.PTPool()
  dev = torch.device("cpu")
  a = pool.alloc((2, 4, 4, 8), dev, 1)
  b = pool.alloc((2, 4, 4, 8), dev, 1)
  pool.pin(a)
  pool.release(a)                      # a recorded launch still reads it: must not be handed out again
  pool.release(b)
  c = pool.alloc((2, 4, 4, 8), dev, 1)
  assert c.data_ptr() == b.data_ptr()
  d = pool.alloc((2, 4, 4, 8), dev, 1)
  assert d.data_ptr() not in (a.data_ptr(), b.data_ptr())      # a fresh buffer, not the pinned one
  pool.release(a)                      # a second release while pinned changes nothing
  assert len(pool.held) == 1
  pool.unpin_all()
  assert not pool.pinned and not pool.held
  e = pool.alloc((2, 4, 4, 8), dev, 1)
  assert e.data_ptr() == a.data_ptr()  # back in circulation
  # pinning something the pool does not own is ignored
  pool.pin(torch.zeros(3))
  assert not pool.pinned


def test_deferring_wgrads_records_instead_of_launching(monkeypatch):
  calls = []
  monkeypatch.setattr(ops, "_conv_wgrad_launch", lambda *a: calls.append(a))
  monkeypatch.setattr(ops, "lib", lambda: types.SimpleNamespace(iic_conv_wgrad_nsplit=lambda g: 7))
  monkeypatch.setattr(ops.ctypes, "byref", lambda g: g)
  g = types.SimpleNamespace(Cout=64, Cin=64, ntaps=9)
  x = torch.zeros(2, 6, 6, 64, dtype=torch.bfloat16)
  dy = torch.zeros(2, 6, 6, 64, dtype=torch.bfloat16)
  with ops.deferring_wgrads() as dw:
    out = ops.conv_wgrad(g, x, dy, 9)
  assert not calls and len(dw.items) == 1 and dw.items[0][5] is out and dw.items[0][7] == 7
  ops.run_deferred_wgrads(dw.items, 1)
  assert len(calls) == 1 and calls[0][-1][1:] == (1, "side") and not dw.items
  out2 = ops.conv_wgrad(g, x, dy, 9)                       # outside the context: launched at once, on its branch's scratch
  assert len(calls) == 2 and calls[1][5] is out2 and calls[1][-1][1:] == (ops.BRANCH[0],)


class _Net(torch.nn.Module):
  def __init__(self):
    super(_Net, self).__init__()
    self.fc = torch.nn.Linear(4, 3)


def test_plan_warms_up_twice_then_captures_then_replays_per_position():
  net = _Net()
  x = torch.zeros(5, 4)
  modes = []
  for step in range(4):
    # an optimiser step between the iterations: the planner sees a new weights epoch and restarts the position count
    graphed._cl.bump_weights_epoch()
    for view in range(2):
      pl = graphed.plan(net, x, {}, 1 if view == 0 else 0)      # first forward of a step on the side branch
      modes.append((step, view, pl.key[4], pl.key[5], pl.mode))
      if pl.mode == "capture":
        pl.st["graphs"][pl.key] = types.SimpleNamespace(sig=graphed._storage_sig(net))   # stands for a _ViewGraph
  assert [m[4] for m in modes] == ["eager", "eager", "eager", "eager", "capture", "capture", "replay", "replay"]
  assert [m[2] for m in modes] == [0, 1] * 4 and [m[3] for m in modes] == [1, 0] * 4


def test_plan_gives_one_stream_positions_their_own_namespace_and_notices_moved_parameters():
  net = _Net()
  x = torch.zeros(5, 4)
  graphed._cl.bump_weights_epoch()
  p0 = graphed.plan(net, x, {}, 0)
  p1 = graphed.plan(net, x, {}, 0)          # second forward of the step on the SAME branch
  assert p0.res == 0 and p1.res >= 100 and p1.res in ops._NO_PROXY_BRANCHES
  # a captured graph whose parameters were re-allocated (.cpu() / .cuda() round trip of a checkpoint) is dropped
  st = p0.st
  st["graphs"][p0.key] = types.SimpleNamespace(sig=graphed._storage_sig(net))
  graphed._cl.bump_weights_epoch()
  assert graphed.plan(net, x, {}, 0).mode == "replay"
  net.fc.weight.data = net.fc.weight.data.clone()
  graphed._cl.bump_weights_epoch()
  pl = graphed.plan(net, x, {}, 0)
  assert pl.mode == "eager" and not st["graphs"]
  # a different batch shape (the ragged last batch) is a key of its own: eager until it has been seen twice
  graphed._cl.bump_weights_epoch()
  assert graphed.plan(net, torch.zeros(3, 4), {}, 0).mode == "eager"


class _FakeDeviceTensor(torch.Tensor):
  """A CPU tensor that claims to live on the device: lets the host-side decisions of ops.auto_branch run here."""
  is_cuda = property(lambda self: True)


def test_auto_branch_keeps_eager_pair_forwards_on_the_callers_stream(monkeypatch):
  """Default since the end of round 4: a forward of the pair that runs eagerly does not fork (ops.AUTO_BRANCH_EAGER
  is the opt-in); the first one marks the pair (_SOLO_FIRST) so that the second one does not fork in its place, both
  postpone their running-statistic updates to the join, and the join clears the mark.  With graph replay on, the
  two warm-up occurrences of a position run the same way inside the position's resource namespace, without
  parameter aliases."""
  calls = []

  class Net(torch.nn.Module):
    def __init__(self):
      super(Net, self).__init__()
      self.fc = torch.nn.Linear(4, 3)

    @ops.auto_branch
    def forward(self, x, head="B"):
      calls.append((ops.BRANCH[0], ops._SOLO_FIRST[0], ops.pv(self.fc.weight) is self.fc.weight))
      return [torch.as_tensor(x).as_subclass(torch.Tensor) @ self.fc.weight.t()]

  monkeypatch.setattr(ops, "flush_deferred_running", lambda: None)
  net = Net().train()
  x = torch.zeros(5, 4).as_subclass(_FakeDeviceTensor)
  prev = ops.AUTO_BRANCH[0], ops.GRAPH_FORWARD[0], ops.AUTO_BRANCH_EAGER[0]
  try:
    ops.AUTO_BRANCH[0], ops.GRAPH_FORWARD[0], ops.AUTO_BRANCH_EAGER[0] = True, False, False
    net(x)
    assert ops._SOLO_FIRST[0] and not ops._PENDING_JOIN and calls[-1] == (0, True, True)
    net(x)
    assert calls[-1] == (0, True, True) and not ops._PENDING_JOIN
    ops.join()
    assert not ops._SOLO_FIRST[0]
    # graph replay on: warm-up occurrences of (position 0, branch 1) run on the caller's stream in namespace 1
    ops.GRAPH_FORWARD[0] = True
    monkeypatch.setattr(graphed, "eligible", lambda mod, x_, a, k: True)
    for step in range(2):
      graphed._cl.bump_weights_epoch()
      net(x)
      assert calls[-1] == (1, True, True) and 1 in ops._NO_PROXY_BRANCHES and 1 not in ops._AUTO_FOLD
      net(x)
      assert calls[-1] == (0, True, True)
      ops.join()
    st = graphed._state(net)
    assert sorted((k[4], k[5]) for k in st["warm"]) == [(0, 1), (1, 0)] and all(v == 2 for v in st["warm"].values())
    # the third occurrence would capture -- and, captured, fork onto the side stream
    graphed._cl.bump_weights_epoch()
    assert graphed.plan(net, x, {}, 1).mode == "capture"
  finally:
    ops.AUTO_BRANCH[0], ops.GRAPH_FORWARD[0], ops.AUTO_BRANCH_EAGER[0] = prev
    ops._SOLO_FIRST[0] = False
