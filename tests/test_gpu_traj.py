"""SURVEY.md section 8c tier T3, end-to-end clause: "loss trajectory over N identical synthetic
steps [bf16] tracks the fp32 reference (same trend, final gap stated)".

30 Adam steps of the reference's train step (cluster_sobel.py:235-272: sobel x2 -> net(all_imgs), net(all_imgs_tf)
-> IID_loss per sub-head -> mean -> backward -> Adam) on ONE fixed batch, from identical initial weights:
  * the production bf16 path (MFMA convolutions, bf16 activations),
  * the same host code on the exact-fp32 kernels (ops.fp32_mode()),
  * the REFERENCE's own modules on the CPU (tests/golden/traj_*.json, oracle/gen_golden_traj.py) at 1 / 2 / 8 BLAS
    threads -- a 30-step fp32 run does not reproduce itself across summation orders, so the fixture is a band.
Gates: the first step to the per-tier tolerances; every 5-step window mean inside the reference band widened by the
stated gap; the windows' trend (sign of consecutive differences) equal to the reference's wherever the reference's own
runs agree on it; the final-loss gap printed and held to what was measured + margin."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _windows(v, n=5):
  v = np.asarray(v, dtype=np.float64)
  return v.reshape(-1, n).mean(axis=1)


def _band(fix):
  runs = np.array([fix["threads"][k] for k in sorted(fix["threads"])], dtype=np.float64)      # [threads, steps]
  w = np.stack([_windows(r) for r in runs])
  return runs, w.min(axis=0), w.max(axis=0), w


def _train(net, opt, a, b, heads, steps, sobel):
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  losses = []
  for _ in range(steps):
    net.zero_grad()
    xo = net(sobel_process(a, False) if sobel else a)
    xt = net(sobel_process(b, False) if sobel else b)
    tot = None
    for i in range(heads):
      l, _ = IID_loss(xo[i], xt[i], lamb=1.0)
      tot = l if tot is None else tot + l
    tot = tot / heads
    tot.backward()
    opt.step()
    losses.append(float(tot.detach()))
  return losses


def _report(name, fix, bf16, fp32):
  runs, lo, hi, w = _band(fix)
  wb, wf = _windows(bf16), _windows(fp32)
  os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
  lines = ["%s: %d identical steps; 5-step window means" % (name, len(bf16)),
           "  reference (CPU, its own modules), band over 1 / 2 / 8 BLAS threads: " +
           " ".join("[%.4f, %.4f]" % (a, b) for a, b in zip(lo, hi)),
           "  HIP fp32 mode : " + " ".join("%.4f" % v for v in wf),
           "  HIP bf16 path : " + " ".join("%.4f" % v for v in wb),
           "  first step    : reference %.6f, fp32 mode %.6f, bf16 %.6f" % (runs[0][0], fp32[0], bf16[0]),
           "  final step    : reference %s, fp32 mode %.6f, bf16 %.6f" % (
             " / ".join("%.6f" % r[-1] for r in runs), fp32[-1], bf16[-1]),
           "  final gap of the bf16 path to the reference band's centre: %.2e (relative %.2e); fp32 mode: %.2e; the band's "
           "own half-width: %.2e" % (abs(bf16[-1] - runs[:, -1].mean()), abs(bf16[-1] - runs[:, -1].mean()) / abs(runs[:, -1].mean()),
                                     abs(fp32[-1] - runs[:, -1].mean()), 0.5 * (runs[:, -1].max() - runs[:, -1].min()))]
  txt = "\n".join(lines)
  with open(os.path.join(ROOT, "gpurun_out", "traj_%s.txt" % name), "w") as f:
    f.write(txt + "\n")
  print(txt)
  return runs, lo, hi, w, wb, wf


def _gates(runs, lo, hi, w, wb, wf, bf16, fp32, first_tol_fp32, first_tol_bf16, gap_bf16, gap_fp32):
  ref0 = runs[0][0]
  assert np.isfinite(bf16).all() and np.isfinite(fp32).all()
  assert abs(fp32[0] - ref0) <= first_tol_fp32 * abs(ref0), (fp32[0], ref0)
  assert abs(bf16[0] - ref0) <= first_tol_bf16 * abs(ref0), (bf16[0], ref0)
  # windows inside the reference's own band, widened by the stated gap (relative to the band centre)
  cen = 0.5 * (lo + hi)
  assert (wf >= lo - gap_fp32 * np.abs(cen)).all() and (wf <= hi + gap_fp32 * np.abs(cen)).all(), (wf, lo, hi)
  assert (wb >= lo - gap_bf16 * np.abs(cen)).all() and (wb <= hi + gap_bf16 * np.abs(cen)).all(), (wb, lo, hi)
  # same trend: wherever all reference runs agree on the sign of a window-to-window change (and it is not a plateau),
  # both HIP paths show that sign
  d = np.sign(np.diff(w, axis=1))
  agree = (d == d[0]).all(axis=0) & (np.abs(np.diff(w, axis=1)).min(axis=0) > 2e-3 * np.abs(cen[1:]))
  assert agree.sum() >= 2, "fixture too flat to say anything about the trend"
  assert (np.sign(np.diff(wb))[agree] == d[0][agree]).all(), (np.diff(wb), d[0], agree)
  assert (np.sign(np.diff(wf))[agree] == d[0][agree]).all(), (np.diff(wf), d[0], agree)
  # the run trains: the last window is below the first by about what the reference's is
  assert wb[-1] < wb[0] and wf[-1] < wf[0]
  ref_drop = (w[:, 0] - w[:, -1]).mean()
  assert abs((wb[0] - wb[-1]) - ref_drop) <= 0.25 * abs(ref_drop) and abs((wf[0] - wf[-1]) - ref_drop) <= 0.15 * abs(ref_drop), \
      (wb[0] - wb[-1], wf[0] - wf[-1], ref_drop)


def test_net5g_bf16_and_fp32_trajectories_track_the_reference():
  from iic_amd import archs, ops
  from iic_amd.optim import Adam
  from oracle import net_oracle
  from oracle.gen_golden_traj import NET5G, STEPS, net5g_init
  fix = json.load(open(os.path.join(G, "traj_net5g.json")))
  assert fix["steps"] == STEPS and fix["config"] == NET5G
  dev = torch.device("cuda", 0)
  imgs, imgs_tf = net_oracle.make_mild_pair(NET5G["n_pairs"], NET5G["input_sz"], 3, seed=NET5G["pair_seed"])
  a, b = imgs.to(dev), imgs_tf.to(dev)
  cfg = types.SimpleNamespace(in_channels=2, input_sz=NET5G["input_sz"], batchnorm_track=True, num_sub_heads=NET5G["heads"],
                              output_k=NET5G["k"])

  def run(fp32):
    net = archs.ClusterNet5g(cfg)
    net.load_state_dict({k: v.clone() for k, v in net5g_init().items()}, strict=True)
    net.to(dev).train()
    opt = Adam(net.parameters(), lr=NET5G["lr"])
    if fp32:
      with ops.fp32_mode():
        return _train(net, opt, a, b, NET5G["heads"], STEPS, True)
    return _train(net, opt, a, b, NET5G["heads"], STEPS, True)
  bf16, fp32 = run(False), run(True)
  runs, lo, hi, w, wb, wf = _report("net5g", fix, bf16, fp32)
  _gates(runs, lo, hi, w, wb, wf, bf16, fp32, first_tol_fp32=2e-4, first_tol_bf16=1e-2, gap_bf16=GAP5G_BF16, gap_fp32=GAP5G_FP32)


def test_net6c_bf16_and_fp32_trajectories_track_the_reference():
  from iic_amd import archs, ops
  from iic_amd.optim import Adam
  from oracle import net_oracle
  from oracle.gen_golden_traj import NET6C, STEPS, net6c_init
  fix = json.load(open(os.path.join(G, "traj_net6c.json")))
  assert fix["steps"] == STEPS and fix["config"] == NET6C
  dev = torch.device("cuda", 0)
  imgs, imgs_tf = net_oracle.make_mild_pair(NET6C["n_pairs"], NET6C["input_sz"], 3, seed=NET6C["pair_seed"])
  a, b = imgs.to(dev), imgs_tf.to(dev)
  cfg = types.SimpleNamespace(in_channels=1, input_sz=NET6C["input_sz"], batchnorm_track=True, num_sub_heads=NET6C["heads"],
                              output_k=NET6C["k"])

  def run(fp32):
    net = archs.ClusterNet6c(cfg)
    net.load_state_dict({k: v.clone() for k, v in net6c_init().items()}, strict=True)
    net.to(dev).train()
    opt = Adam(net.parameters(), lr=NET6C["lr"])
    if fp32:
      with ops.fp32_mode():
        return _train(net, opt, a, b, NET6C["heads"], STEPS, False)
    return _train(net, opt, a, b, NET6C["heads"], STEPS, False)
  bf16, fp32 = run(False), run(True)
  runs, lo, hi, w, wb, wf = _report("net6c", fix, bf16, fp32)
  _gates(runs, lo, hi, w, wb, wf, bf16, fp32, first_tol_fp32=1e-3, first_tol_bf16=5e-2, gap_bf16=GAP6C_BF16, gap_fp32=GAP6C_FP32)


# The stated gaps: how far outside the reference's own band a 5-step window mean may lie, relative to the band's
# centre.  Measured on the MI355X (profiles/r06_traj_net5g.txt, r06_traj_net6c.txt):
#   ClusterNet5g  fp32 mode 2e-4 in every window (it IS the reference's trajectory: -2.1736 vs -2.1710 ... -2.1726 after
#                 30 steps), bf16 2.9 % in the first window (-0.984 vs -1.013: the steps where the loss falls fastest),
#                 0.2 % from the third window on, final-loss gap 2.0e-3 relative;
#   ClusterNet6c  the reference's own runs spread by 4 % after 30 steps (-1.068 / -1.077 / -1.114 at 1 / 2 / 8 BLAS
#                 threads: MI ~ 0 at the start, lr 1e-3); fp32 mode 1.1 % outside that band, bf16 5.9 % (final loss
#                 -1.167: it trains slightly FASTER than any fp32 run, same shape).
GAP5G_BF16, GAP5G_FP32 = 0.05, 0.005
GAP6C_BF16, GAP6C_FP32 = 0.10, 0.03
