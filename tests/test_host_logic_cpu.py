"""Host-side bookkeeping that needs no GPU: the graph-replay planner of the drop-in path (iic_amd/graphed.py::plan: which occurrence of a training forward runs
eagerly, which one is captured, which ones replay; positions within a step; resource namespaces of one-stream runs)."""
import types

import torch

from iic_amd import graphed, ops


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
  """A forward of the pair that runs eagerly does not fork; the first one marks the pair (_SOLO_FIRST = 1) so that the
  second one does not fork in its place; both postpone their running-statistic updates, and the second one ends the
  pair by applying them itself (both ran on the caller's stream: nothing to wait for -- a checkpoint taken right after
  a step with a foreign loss and a foreign optimiser sees them).
  A first forward that raises clears it too.  With graph replay on, the two warm-up occurrences of a position run the
  same way inside the position's resource namespace, without parameter aliases."""
  calls = []

  class Net(torch.nn.Module):
    def __init__(self):
      super(Net, self).__init__()
      self.fc = torch.nn.Linear(4, 3)

    @ops.auto_branch
    def forward(self, x, head="B"):
      calls.append((ops.BRANCH[0], ops._SOLO_FIRST[0], ops.pv(self.fc.weight) is self.fc.weight))
      if getattr(self, "boom", False):
        raise RuntimeError("boom")
      return [torch.as_tensor(x).as_subclass(torch.Tensor) @ self.fc.weight.t()]

  flushes = []
  monkeypatch.setattr(ops, "flush_deferred_running", lambda: flushes.append(1))
  net = Net().train()
  x = torch.zeros(5, 4).as_subclass(_FakeDeviceTensor)
  prev = ops.AUTO_BRANCH[0], ops.GRAPH_FORWARD[0]
  try:
    ops.AUTO_BRANCH[0], ops.GRAPH_FORWARD[0] = True, False
    net(x)
    assert ops._SOLO_FIRST[0] == 1 and not ops._PENDING_JOIN and calls[-1] == (0, 1, True)
    n_fl = len(flushes)
    net(x)
    # the second view ends the pair: mark cleared, postponed running-statistic updates flushed -- with no join of ours
    assert calls[-1] == (0, 1, True) and not ops._PENDING_JOIN and ops._SOLO_FIRST[0] == 0 and len(flushes) == n_fl + 1
    ops.join()
    assert not ops._SOLO_FIRST[0]
    # nothing of ours joins (a foreign loss and a foreign optimiser): every pair still flushes once, at its second view
    n_fl = len(flushes)
    net(x); net(x)
    assert ops._SOLO_FIRST[0] == 0 and len(flushes) == n_fl + 1
    net(x)
    assert ops._SOLO_FIRST[0] == 1 and calls[-1] == (0, 1, True)
    net(x)
    ops.join()
    # a first forward that raises does not leave the mark behind
    net.boom = True
    try:
      net(x)
    except RuntimeError:
      pass
    net.boom = False
    assert ops._SOLO_FIRST[0] == 0
    # graph replay on: warm-up occurrences of (position 0, branch 1) run on the caller's stream in namespace 1
    ops.GRAPH_FORWARD[0] = True
    monkeypatch.setattr(graphed, "eligible", lambda mod, x_, a, k: True)
    for step in range(2):
      graphed._cl.bump_weights_epoch()
      net(x)
      assert calls[-1] == (1, 1, True) and 1 in ops._NO_PROXY_BRANCHES
      net(x)
      assert calls[-1] == (0, 1, True)
      ops.join()
    st = graphed._state(net)
    assert sorted((k[4], k[5]) for k in st["warm"]) == [(0, 1), (1, 0)] and all(v == 2 for v in st["warm"].values())
    # the third occurrence would capture -- and, captured, fork onto the side stream
    graphed._cl.bump_weights_epoch()
    assert graphed.plan(net, x, {}, 1).mode == "capture"
  finally:
    ops.AUTO_BRANCH[0], ops.GRAPH_FORWARD[0] = prev
    ops._SOLO_FIRST[0] = 0


def test_sig_changed_notices_a_buffer_replaced_through_a_submodule():
  """nn.Module._apply on a SUBMODULE (net.trunk.cuda(), bn.to(...)) replaces buffer objects without passing the
  architecture's _apply counter: the cached tensor list would keep comparing the OLD buffers' unchanged addresses and a
  captured graph would go on writing running statistics into them.  The buffer dictionaries are compared
  by identity on every call."""
  net = torch.nn.Sequential(torch.nn.Conv2d(1, 2, 3), torch.nn.BatchNorm2d(2))
  st = {}
  vg = types.SimpleNamespace(sig=graphed._storage_sig(net))
  assert not graphed._sig_changed(net, st, vg)
  assert not graphed._sig_changed(net, st, vg)
  old = net[1].running_mean
  net[1]._apply(lambda t: t.clone())           # what .cuda() / .to() do to a submodule: new buffer objects, new addresses
  assert net[1].running_mean is not old
  assert graphed._sig_changed(net, st, vg)      # (the old object is still alive in the cached list: addresses alone would say "unchanged")
  vg2 = types.SimpleNamespace(sig=graphed._storage_sig(net))
  assert not graphed._sig_changed(net, st, vg2)


def test_branch_state_is_one_context_per_thread():
  """Round 6: the fork / join state of the two-stream execution -- current branch, pending joins,
  postponed running-statistic updates, leaf aliases, the solo-pair mark -- lives in ONE BranchContext per thread; the
  module-level names are views of the calling thread's context, so a second thread (a DataParallel-style worker, the
  autograd engine's backward thread) starts clean and cannot disturb the first."""
  import threading
  main = ops.context()
  assert ops.context() is main
  prev = ops.BRANCH[0]
  seen = {}
  try:
    ops.BRANCH[0] = 2
    ops._PENDING_JOIN.append(("main", "side"))
    ops._DEFERRED_RUNNING.append("update")
    ops._NO_PROXY_BRANCHES.add(7)
    ops._SOLO_FIRST[0] = 1

    def worker():
      seen["ctx_is_other"] = ops.context() is not main
      seen["state"] = (ops.BRANCH[0], len(ops._PENDING_JOIN), len(ops._DEFERRED_RUNNING), 7 in ops._NO_PROXY_BRANCHES,
                       ops._SOLO_FIRST[0], ops._BRANCH_MAIN[0], ops._CAPTURE_PROXIES[0])
      ops.BRANCH[0] = 5
      ops._PENDING_JOIN.append(("w", "w"))
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen["ctx_is_other"] and seen["state"] == (0, 0, 0, False, 0, None, None)
    assert ops.BRANCH[0] == 2 and list(ops._PENDING_JOIN) == [("main", "side")] and main.branch == 2
    assert main.pending_join is ops._PENDING_JOIN._o() and main.solo_first == 1
  finally:
    ops.BRANCH[0] = prev
    del ops._PENDING_JOIN[:]
    del ops._DEFERRED_RUNNING[:]
    ops._NO_PROXY_BRANCHES.discard(7)
    ops._SOLO_FIRST[0] = 0
