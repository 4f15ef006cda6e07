#!/usr/bin/env python
"""Benchmark of the IIC training hot path on MI355X (BASELINE.json metric).

One "step" = the reference's train step on a resident synthetic batch
(/root/reference/code/scripts/cluster/cluster_sobel.py:235-272):
  sobel_process x2 -> net(all_imgs), net(all_imgs_tf) (train-mode BN, two statistic groups)
  -> IID_loss x5 sub-heads -> mean -> backward -> (grad all-reduce when N>1) -> Adam step.
Workload: STL10-shaped 96x96, ClusterNet5g, output_k 70, 5 sub-heads, 660 pairs per GPU
(BASELINE.json configs[1]).  Multi-GPU: one process per GPU (torchrun), batch sharded by
pair, weak scaling (660 pairs per rank), RCCL all-reduce of the raw joint P (inside
IID_loss) and of the parameter gradients.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time
import types

T_PROCESS_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import torch  # noqa: E402

PAIRS_PER_GPU = 660
INPUT_SZ = 96
OUTPUT_K = 70
SUB_HEADS = 5
FLOP_PER_PAIR = 36.35e9      # BASELINE.md §4 (algorithmic, fwd + bwd-data + bwd-weight, 2 images)
BF16_PEAK_TFLOPS = 2500.0    # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0        # HBM3E (same guide; 6.3 TB/s is what a streaming kernel measures)


def make_batch(n_pairs, sz, device, seed=0):
  """Synthetic (all_imgs, all_imgs_tf) in [0,1], fp32 [n,1,sz,sz]: smoothed noise base images
  replicated 3x (the reference's num_dataloaders), tf = flip * gain + noise."""
  g = torch.Generator(device="cpu").manual_seed(seed)
  nb = n_pairs // 3
  base = torch.rand(nb, 1, sz, sz, generator=g)
  base = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(base, (2, 2, 2, 2), mode="replicate"), 5, 1)
  imgs = base.repeat(3, 1, 1, 1)
  gain = torch.rand(n_pairs, 1, 1, 1, generator=g) * 0.8 + 0.6
  noise = torch.randn(imgs.shape, generator=g) * 0.05
  imgs_tf = torch.clamp(torch.flip(imgs, dims=[3]) * gain + noise, 0.0, 1.0)
  return imgs.to(device).contiguous(), imgs_tf.to(device).contiguous()


def _free_port():
  import socket
  sk = socket.socket()
  sk.bind(("127.0.0.1", 0))
  p = sk.getsockname()[1]
  sk.close()
  return p


def self_launch(args):
  """`python bench.py --gpus N` (N > 1) outside a launcher: become N ranks -- the reference's counterpart is one line
  that just works (torch.nn.DataParallel over the visible devices: cluster_sobel.py:146, segmentation_twohead.py:173).
  Re-executes this command under `python -m torch.distributed.run --nproc-per-node N` (one process per GPU, RCCL); exits
  non-zero with a message when the box has fewer than N devices -- never a silent 1-rank run."""
  backend = os.environ.get("IIC_DIST_BACKEND", "nccl")
  ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
  if ndev < args.gpus and backend == "nccl":
    sys.stderr.write("bench.py: --gpus %d needs %d MI355X devices, this machine has %d (RCCL wants one device per rank; "
                     "IIC_DIST_BACKEND=gloo runs the ranks on shared devices as a functional check only)\n"
                     % (args.gpus, args.gpus, ndev))
    sys.exit(2)
  if ndev == 0:
    sys.stderr.write("bench.py: no GPU visible\n")
    sys.exit(2)
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  sys.stderr.write("bench.py: --gpus %d without a launcher environment: %s\n" % (args.gpus, " ".join(cmd)))
  sys.stderr.flush()
  os.execv(sys.executable, cmd)


class Ranks(object):
  """This process's place in the job: WORLD_SIZE / RANK / LOCAL_RANK from the launcher, the device, the process group
  (RCCL unless IIC_DIST_BACKEND says otherwise).  `on` = the collectives run: world > 1, or IIC_DIST_FORCE=1 at world
  size 1 (a single MI355X then executes the whole N > 1 path through real RCCL calls: tests/test_gpu_rccl.py)."""

  def __init__(self, args):
    self.world = int(os.environ.get("WORLD_SIZE", "1"))
    self.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
      sys.stderr.write("bench.py needs an MI355X\n")
      sys.exit(2)
    if args.gpus != self.world:
      sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s) (torchrun --nproc-per-node must equal "
                       "--gpus)\n" % (args.gpus, self.world))
      sys.exit(2)
    ndev = torch.cuda.device_count()
    self.backend = os.environ.get("IIC_DIST_BACKEND", "nccl")   # "gloo": functional check of the N>1
    if local_rank >= ndev:                                      # path on a 1-GPU box (ranks share cuda:0)
      if self.backend == "nccl" and os.environ.get("IIC_RCCL_SHARED_DEVICE", "0") != "1":
        sys.stderr.write("bench.py: rank %d has no device of its own (%d visible) and RCCL wants one per rank\n"
                         % (self.rank, ndev))
        sys.exit(2)
      local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    self.dev = torch.device("cuda", local_rank)
    self.on = self.world > 1 or os.environ.get("IIC_DIST_FORCE", "0") == "1"
    if self.on:
      import torch.distributed as dist
      os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
      os.environ.setdefault("MASTER_PORT", str(_free_port()))
      kw = {"device_id": self.dev} if (self.backend == "nccl" and os.environ.get("IIC_RCCL_EAGER_INIT", "0") == "1") else {}
      dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, **kw)
      from iic_amd import dist as idist
      idist.enable()

  def broadcast(self, net):
    if self.on:      # identical weights (and BatchNorm buffers) on every rank
      from iic_amd import dist as idist
      idist.broadcast_module_state(net, 0)

  def fence(self):
    torch.cuda.synchronize()
    if self.on:
      torch.distributed.barrier()
    torch.cuda.synchronize()

  def max_over_ranks(self, dt):
    if not self.on:
      return dt
    t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t)

  def report(self):
    """What the record says about the data-parallel side of the run (collective: every rank calls it)."""
    if not self.on:
      return None
    from iic_amd import dist as idist
    from iic_amd import graph as igraph
    # every rank must have issued the SAME sequence of collectives (the stream probes and the capture fall-back are
    # the places where ranks could diverge): the per-rank totals, gathered
    totals = [None] * self.world
    torch.distributed.all_gather_object(totals, {k: int(v) for k, v in idist.CALLS.items()})
    return {"backend": "%s%s" % (self.backend, " (= RCCL)" if self.backend == "nccl" else ""), "world_size": self.world,
            "forced_at_world_size_1": self.world == 1,
            "collectives_issued_by_rank0": dict(idist.CALLS),
            "collectives_issued_per_rank": totals,
            "stream_probes": igraph.PROBE_LOG[-24:]}

  def close(self):
    if self.on:
      torch.distributed.destroy_process_group()



_HOLD = {}


def gpu_hold(ms):
  """Park the current stream for ~ms milliseconds (a spinning one-wave kernel) so that the host can enqueue the whole
  instrumented step behind it.  Without this the eager instrumented steps are host-paced on a slow host: the stream
  reaches an event record before the kernel that follows it has been submitted, and the event pair around a launch then
  also measures the host's gap (seen as roofline.frac 0.25-0.26 on some boxes against 0.32-0.33 on others for the same
  kernels and the same rocprofv3 durations)."""
  if "cycles_per_ms" not in _HOLD:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(4000000)
    e1.record()
    torch.cuda.synchronize()
    _HOLD["cycles_per_ms"] = 4000000.0 / max(e0.elapsed_time(e1), 1e-3)
  torch.cuda._sleep(int(ms * _HOLD["cycles_per_ms"]))


class ConvTimer(object):
  """HIP-event timing of the three kernel families of the step, on the launch stream (torch's current stream == the
  stream handed to the C ABI), around every call of the wrappers the architectures go through:
    conv     ops.conv_igemm                       forward + backward-data implicit GEMM      (MFMA roof, bf16)
    wgrad    ops.conv_wgrad                       weight gradient incl. its split-K reduce   (MFMA roof, bf16)
    bn       ops.bn_apply / bn_bwd_reduce / bn_bwd_apply   BatchNorm / ReLU / residual passes (HBM roof)
  Algorithmic work per call from the geometry descriptor / the tensors handed over."""

  def __init__(self):
    self.records = []          # conv family (kept under this name: summary() is the bench record's `roofline`)
    self.fam = {"wgrad": [], "bn": []}
    self._orig = {}

  def _wrap(self, ops, name, fam, work):
    orig = getattr(ops, name)
    self._orig[name] = orig
    dst = self.records if fam == "conv" else self.fam[fam]

    def timed(*a, **k):
      e0 = torch.cuda.Event(enable_timing=True)
      e1 = torch.cuda.Event(enable_timing=True)
      e0.record()
      r = orig(*a, **k)
      e1.record()
      fl, by = work(*a, **k)
      dst.append((e0, e1, fl, by))
      return r
    setattr(ops, name, timed)

  def install(self):
    from iic_amd import ops

    def conv_work(g, x_pt, w_t, out_pt, stats=None, res_grad=None, res_act=None, accumulate=False, premask=False, red=None):
      flops = 2.0 * g.N * g.MY * g.MX * g.Cout * g.Cin * g.ntaps
      # algorithmic HBM bytes: read the input rows once, write the output rows, read the weight
      # slice once, plus every tensor a fused epilogue must read once (previous contents of an
      # accumulating launch, residual gradient, mask activation, the y / y2 tiles of a fused
      # BatchNorm-backward reduction) -- all bf16
      rows = float(g.N * g.MY * g.MX)
      extra = sum(1 for t in (res_grad, res_act) if t is not None) + (1 if accumulate else 0)
      if red is not None:
        extra += 1 + (1 if red[3] is not None else 0)
      return flops, 2.0 * (rows * g.Cin + rows * g.Cout * (1 + extra) + g.ntaps * g.Cout * g.Cin)

    def wgrad_work(g, x_pt, dy_pt, wtaps, *a, **k):
      rows = float(g.N * g.MY * g.MX)
      # read the input rows and the gradient rows once (bf16), write the fp32 gradient once
      return 2.0 * rows * g.Cout * g.Cin * g.ntaps, 2.0 * rows * (g.Cin + g.Cout) + 4.0 * g.ntaps * g.Cout * g.Cin

    def bn_bytes(n_tensors, N, H, W, C):
      return 0.0, 2.0 * n_tensors * float(N) * H * W * C          # interiors only, bf16

    def bn_apply_work(y, coef, out, N, H, W, P, C, res=None, y2=None, coef2=None, relu=True):
      return bn_bytes(2 + (res is not None) + (y2 is not None), N, H, W, C)

    def bn_red_work(dout, act, y, sums, N, H, W, P, C, y2=None, sums2=None, mask_coef=None):
      return bn_bytes(2 + (act is not None) + (y2 is not None), N, H, W, C)

    def bn_bapply_work(dout, act, y, bcoef, dy, N, H, W, P, C, y2=None, bcoef2=None, dy2=None, mask_coef=None):
      return bn_bytes(3 + (act is not None) + (y2 is not None) + (dy2 is not None), N, H, W, C)
    self._wrap(ops, "conv_igemm", "conv", conv_work)
    self._wrap(ops, "conv_wgrad", "wgrad", wgrad_work)
    self._wrap(ops, "bn_apply", "bn", bn_apply_work)
    self._wrap(ops, "bn_bwd_reduce", "bn", bn_red_work)
    self._wrap(ops, "bn_bwd_apply", "bn", bn_bapply_work)

  def uninstall(self):
    from iic_amd import ops
    for name, orig in self._orig.items():
      setattr(ops, name, orig)
    self._orig = {}

  @staticmethod
  def _sum(recs):
    if not recs:
      return None
    tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs)
    n = len(recs)
    return {"launches": n, "avg_us": 1e3 * tot_ms / n, "tflops": sum(f for _, _, f, _ in recs) / (tot_ms * 1e-3) / 1e12,
            "total_ms": tot_ms, "alg_bytes": sum(b for _, _, _, b in recs) / n,
            "tbytes_per_s": sum(b for _, _, _, b in recs) / (tot_ms * 1e-3) / 1e12}

  def summary(self):
    return self._sum(self.records)

  def families(self, n_steps):
    """[{family, bound, achieved, peak, unit, frac, kernel_ms_per_step, launches_per_step}] -- SURVEY 8d: "balanced:
    report both fractions"."""
    out = []
    for name, recs, bound, what in (
        ("conv_fwd_bwd_data", self.records, "mfma", "conv_igemm_{bd,pw,p64} + conv_igemm (forward + backward-data)"),
        ("weight_gradient", self.fam["wgrad"], "mfma", "conv_wgrad_dma / conv_wgrad + conv_wgrad_reduce"),
        ("bn_hbm", self.fam["bn"], "hbm", "bn_apply + bn_bwd_reduce + bn_bwd_apply (BatchNorm / ReLU / residual passes)")):
      s = self._sum(recs)
      if not s:
        continue
      ach, peak, unit = (s["tflops"], BF16_PEAK_TFLOPS, "TFLOP/s") if bound == "mfma" else (s["tbytes_per_s"] * 1e3, HBM_PEAK_GBS, "GB/s")
      out.append({"family": name, "kernels": what, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                  "frac": ach / peak, "kernel_ms_per_step": s["total_ms"] / n_steps,
                  "launches_per_step": s["launches"] / float(n_steps), "algorithmic_bytes_per_launch": s["alg_bytes"]})
    return out


def pmc_step_hbm():
  """(HBM bytes of one whole step, file) from the newest committed PMC passes (tools/pmc_traffic.py), or (None, None)."""
  import glob
  for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True):
    try:
      d = json.load(open(p))
      if d.get("step_hbm_bytes"):
        return float(d["step_hbm_bytes"]), os.path.relpath(p, ROOT)
      # (records written before round 6 carry no step total: every launch of the 2-step PMC command, halved)
      tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in d["kernels"].values())
      return tot / 2.0, os.path.relpath(p, ROOT)
    except Exception:
      continue
  return None, None


def pmc_traffic():
  """(HBM bytes per conv_igemm launch, file it came from) -- from the newest committed rocprofv3 PMC passes
  of this same command (tools/pmc_traffic.py; PMC cannot be collected from inside the process), or
  (None, None)."""
  import glob
  for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True):
    try:
      return float(json.load(open(p))["conv_igemm_hbm_bytes_per_launch"]), os.path.relpath(p, ROOT)
    except Exception:
      continue
  return None, None


def _cpu_step_fn(n_pairs):
  """One train step of the CPU baseline as a callable, and its kind: the reference's OWN modules (ClusterNet5g,
  IID_loss, torch Adam, through the Python-2 import hook) when its tree is on this machine ($IIC_REFERENCE or
  /root/reference -- never the case on the driver's GPU box), else the oracle restatement ("port")."""
  ref = os.environ.get("IIC_REFERENCE", "/root/reference")
  if os.path.exists(os.path.join(ref, "code", "archs", "cluster", "net5g.py")) and not os.environ.get("IIC_CPU_BASELINE_PORT"):
    try:
      os.environ["IIC_REFERENCE"] = ref
      from oracle import net_oracle, ref_import
      ref_net5g = ref_import.ref_cluster_archs()["net5g"]
      ref_loss = ref_import.ref_cluster_losses().IID_loss
      ref_sobel = ref_import.ref_sobel_process()
      cfg = types.SimpleNamespace(in_channels=2, input_sz=INPUT_SZ, output_k=OUTPUT_K, num_sub_heads=SUB_HEADS,
                                  batchnorm_track=True)
      net = ref_net5g.ClusterNet5g(cfg).train()
      opt = torch.optim.Adam(net.parameters(), lr=1e-4)
      imgs, imgs_tf = net_oracle.make_paired_batch(n_pairs, INPUT_SZ, 3, seed=0)

      def step():
        opt.zero_grad()
        xo, xt = net(ref_sobel(imgs, False)), net(ref_sobel(imgs_tf, False))
        tot = None
        # (IID_losses.py:18-19 clamps in place into an expand()-ed view, which torch >= 1.x refuses to back-propagate:
        #  the unmodified function runs with Tensor.expand materialised, as in oracle/gen_golden.py -- same values)
        orig_expand = torch.Tensor.expand
        torch.Tensor.expand = lambda self, *a, **k: orig_expand(self, *a, **k).clone()
        try:
          for i in range(SUB_HEADS):
            l, _ = ref_loss(xo[i], xt[i], lamb=1.0)
            tot = l if tot is None else tot + l
        finally:
          torch.Tensor.expand = orig_expand
        (tot / SUB_HEADS).backward()
        opt.step()
      return step, "reference"
    except Exception as e:      # noqa: BLE001  (fall back to the port; say why)
      sys.stderr.write("cpu_baseline: reference modules not usable (%s: %s): oracle port\n" % (type(e).__name__, e))
  from oracle import net_oracle
  params = net_oracle.make_net5g_params(2, OUTPUT_K, SUB_HEADS, True, seed=0)
  leaves = []
  for k, v in params.items():
    if v.dtype.is_floating_point and "running" not in k:
      v.requires_grad_(True)
      leaves.append(v)
  opt = torch.optim.Adam(leaves, lr=1e-4)
  imgs, imgs_tf = net_oracle.make_paired_batch(n_pairs, INPUT_SZ, 3, seed=0)

  def step():
    opt.zero_grad()
    loss, _, _, _ = net_oracle.net5g_train_step_loss(params, imgs, imgs_tf, 1.0, INPUT_SZ, SUB_HEADS)
    loss.backward()
    opt.step()
  return step, "port"


def cpu_baseline(n_pairs=96, budget_s=45.0):
  """Reference-equivalent CPU path on the host cores, bounded sample of the same workload (96 pairs per step).
  torch's intra-op pool is swept over 32 / 64 / 128 threads (one warm-up + timed steps each inside the budget) and
  the best is reported with its thread count: on the 256-core GPU hosts the default (all cores) is >50x SLOWER for
  these small convolutions (measured 0.2 pairs/s at 256 threads, round 1)."""
  step, kind = _cpu_step_fn(n_pairs)
  ncpu = os.cpu_count() or 1
  sweep, t_start = {}, time.time()
  for nt in [n for n in (32, 64, 128) if n <= ncpu] or [ncpu]:
    if sweep and time.time() - t_start > budget_s:
      break
    torch.set_num_threads(nt)
    times = []
    for s_i in range(3):
      t0 = time.time()
      step()
      times.append(time.time() - t0)
      if s_i >= 1 and time.time() - t_start > budget_s:
        break
    sweep[nt] = n_pairs / min(times[1:] or times)
  best = max(sweep, key=sweep.get)
  torch.set_num_threads(min(ncpu, 32))
  return {"value": sweep[best], "unit": "paired-images/sec", "cores": best, "kind": kind,
          "threads_sweep_pairs_per_s": {str(k): round(v, 2) for k, v in sweep.items()},
          "sample": "%d pairs/step, 1 warm-up + up to 2 timed steps per thread count, fp32 torch-CPU %s of the "
                    "ClusterNet5g+IID_loss+Adam step, 96x96, k=70, 5 sub-heads"
                    "%s" % (n_pairs, "run of the reference's own modules" if kind == "reference" else "restatement (oracle/)",
                            "" if kind == "reference" else "; the restatement runs at 0.9-1.3 x the rate of the reference's own "
                            "modules on the same host (profiles/r05_cpu_reference_vs_port.txt: 18.5 vs 19.7 pairs/s on 8 cores)")}


def cpu_baseline_mnist(batch=700, budget_s=40.0):
  """BASELINE.json configs[0] -- the reference's own CPU-runnable case: MNIST 24x24,
  ClusterNet6cTwoHead (k_A 50 overclustering / k_B 10), batch 700, 5 sub-heads
  (cluster_greyscale_twohead.py; examples/commands.txt:30) -- one head-A step + one head-B step
  of the oracle restatement on the host cores, FULL batch.  Reported beside the north-star CPU
  baseline."""
  from oracle import iid_oracle, net_oracle
  torch.set_num_threads(min(os.cpu_count() or 1, 32))
  params = net_oracle.make_net6c_params(1, 24, num_sub_heads=SUB_HEADS, batchnorm_track=True, seed=0,
                                        heads=("head_A", "head_B"), output_ks=[50, 10])
  leaves = []
  for k, v in params.items():
    if v.dtype.is_floating_point and "running" not in k:
      v.requires_grad_(True)
      leaves.append(v)
  opt = torch.optim.Adam(leaves, lr=1e-4)
  imgs, imgs_tf = net_oracle.make_paired_batch(batch - batch % 5, 24, 5, seed=0)

  def one(head):
    opt.zero_grad()
    xo = net_oracle.net6c_forward(params, imgs, True, head, SUB_HEADS)
    xt = net_oracle.net6c_forward(params, imgs_tf, True, head, SUB_HEADS)
    tot = None
    for i in range(SUB_HEADS):
      l, _ = iid_oracle.IID_loss(xo[i], xt[i], lamb=1.0)
      tot = l if tot is None else tot + l
    (tot / SUB_HEADS).backward()
    opt.step()
  times, t_start = [], time.time()
  for s in range(4):
    if s >= 2 and time.time() - t_start > budget_s:
      break
    t0 = time.time()
    one("head_A")
    one("head_B")
    times.append(time.time() - t0)
  t = sorted(times[1:])[len(times[1:]) // 2]
  return {"value": imgs.size(0) / t, "unit": "paired-images/sec (one head-A + one head-B step per batch)",
          "cores": torch.get_num_threads(), "kind": "port",
          "sample": "BASELINE configs[0]: MNIST 24x24 ClusterNet6cTwoHead k_A 50 / k_B 10, batch %d, 5 "
                    "sub-heads, %d timed A+B step pairs (+1 warm-up), fp32 torch-CPU restatement"
                    % (imgs.size(0), len(times) - 1)}


def reference_api_rate(cfg, dev, imgs, imgs_tf, pairs, steps, auto_branch=False, graph_forward=False):
  """The same step through the reference's OWN call sequence (cluster_sobel.py:235-272 as the
  unchanged script issues it): net(x) -> python list of sub-head tensors, IID_loss once per
  sub-head, `+=` / `/=` averaging, stock torch.optim.Adam, `.item()` reads of the loss
  (cluster_sobel.py:255-266), eager launches.  Reported next to the headline number so that the
  cost of the drop-in boundary is visible."""
  from iic_amd import archs, ops
  from iic_amd.losses import IID_loss
  from iic_amd.transforms import sobel_process
  torch.manual_seed(0)
  net = archs.ClusterNet5g(cfg).to(dev).train()
  if graph_forward:      # what `python -m iic_amd.run` sets up: get_opt("Adam") is rebound to the fused HIP Adam
    from iic_amd.optim import Adam
    opt = Adam(net.parameters(), lr=1e-4)
  else:
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
  prev_auto, prev_graph = ops.AUTO_BRANCH[0], ops.GRAPH_FORWARD[0]
  ops.AUTO_BRANCH[0] = bool(auto_branch)
  ops.GRAPH_FORWARD[0] = bool(graph_forward)

  def step():
    net.zero_grad()
    xo = net(sobel_process(imgs, False))
    xt = net(sobel_process(imgs_tf, False))
    avg, avg_nl = None, None
    for i in range(cfg.num_sub_heads):
      loss, loss_nl = IID_loss(xo[i], xt[i], lamb=1.0)
      if avg is None:
        avg, avg_nl = loss, loss_nl
      else:
        avg += loss
        avg_nl += loss_nl
    avg /= cfg.num_sub_heads
    avg_nl /= cfg.num_sub_heads
    v = avg.item() + avg_nl.item()        # the script reads both every step (:262-266)
    avg.backward()
    opt.step()
    return v
  for _ in range(4 if graph_forward else 2):      # (a shape is captured after it has been seen twice)
    step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    v = step()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  ops.AUTO_BRANCH[0], ops.GRAPH_FORWARD[0] = prev_auto, prev_graph
  what = ("list-returning net(x), IID_loss per sub-head, %s, loss .item() every step -- the unchanged script's "
          "call sequence" % ("the fused HIP Adam behind get_opt" if graph_forward else "torch.optim.Adam"))
  if graph_forward:
    what += (", its two forwards on two streams and each forward / backward replayed as a captured HIP graph "
             "(iic_amd/graphed.py: what `python -m iic_amd.run` does by default)")
  else:
    what += ", eager launches, one stream"
  return {"paired_images_per_sec": pairs / dt, "ms_per_step": 1e3 * dt, "final_loss": v, "what": what}


SEG_CONFIGS = {
  # BASELINE.json configs[3]: Potsdam-3 200x200 SegmentationNet10aTwoHead, batch 75, in_ch 4,
  # k_A 24 / k_B 3 (commands.txt:83); BASELINE asks T = 1, the reference's own setting is T = 10
  "potsdam3": dict(bn=75, sz=200, in_ch=4, k_A=24, k_B=3, T=1, mask_p=1.0),
  # configs[4]: COCO-Stuff-3 128x128, RGB + Sobel = 5 channels, batch 120, k_A 15 / k_B 3, T = 10,
  # stuff mask (commands.txt:74)
  "coco3": dict(bn=120, sz=128, in_ch=5, k_A=15, k_B=3, T=10, mask_p=0.6),
}
FP32_MFMA_PEAK_TFLOPS = 157.3


def bench_segmentation(args):
  """`--config potsdam3|coco3`: the two-head segmentation train step
  (segmentation_twohead.py:300-345: head A step + head B step = forward x2,
  IID_segmentation_loss_uncollapsed, backward, Adam each), one GPU, synthetic resident inputs.
  `roofline` is the P-matrix contraction (seg_joint_kernel forward + the two seg_grad_kernel
  launches of its backward, fp32 MFMA 16x16x4) of head A, timed with HIP events on the launch
  stream in the timed steps themselves; `roofline_conv` is the implicit-GEMM conv family."""
  from iic_amd import archs, ops
  from iic_amd.optim import Adam
  from iic_amd.seg_losses import IID_segmentation_loss_uncollapsed
  c = dict(SEG_CONFIGS[args.config])
  if args.T is not None:
    c["T"] = args.T
  rk = Ranks(args)
  dev, world = rk.dev, rk.world
  from iic_amd import dist as idist
  torch.manual_seed(0)
  cfg = types.SimpleNamespace(in_channels=c["in_ch"], input_sz=c["sz"], batchnorm_track=True, num_sub_heads=1,
                              output_k_A=c["k_A"], output_k_B=c["k_B"])
  net = archs.SegmentationNet10aTwoHead(cfg).to(dev).train()
  rk.broadcast(net)
  params = list(net.parameters())
  opt = Adam(params, lr=1e-4)
  # N > 1 (BASELINE configs[3] / [4] are 4 / 8 GPUs by name; the reference runs them under nn.DataParallel,
  # segmentation_twohead.py:173): every rank owns `bn` pairs of its own (weak scaling), the per-shift raw joints are
  # all-reduced inside the loss (iic_amd/seg_losses.py; additive over samples exactly like the reference's conv over
  # the batch, segmentation/IID_losses.py:125), the parameter gradients after backward
  g = torch.Generator().manual_seed(rk.rank)
  bn, sz, T = c["bn"], c["sz"], c["T"]
  x = torch.rand(bn, c["in_ch"], sz, sz, generator=g).to(dev)
  xt = (torch.flip(x, dims=[3]) * 0.9 + 0.05).contiguous()
  aff = torch.zeros(bn, 2, 3, device=dev)
  aff[:, 0, 0] = -1.0
  aff[:, 1, 1] = 1.0
  mask = (torch.rand(bn, sz, sz, generator=g) < c["mask_p"]).float().to(dev)
  ev = []

  # the two views of a step are independent until the loss: by default the first runs on a side
  # stream (iic_amd.ops.branch; --no-branch = one stream); same kernels, same arithmetic
  two_streams = [not args.no_branch]

  def step(timed=False):
    for head in ("A", "B"):
      net.zero_grad(set_to_none=True)
      ops.clear_branch_grads()
      if two_streams[0]:
        with ops.branch():
          a = net(x, head=head)[0]
      else:
        a = net(x, head=head)[0]
      b = net(xt, head=head)[0]
      a_d, b_d = a.detach().requires_grad_(True), b.detach().requires_grad_(True)
      e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
      e[0].record()
      loss, _ = IID_segmentation_loss_uncollapsed(a_d, b_d, all_affine2_to_1=aff, all_mask_img1=mask, lamb=1.0,
                                                  half_T_side_dense=T, half_T_side_sparse_min=0,
                                                  half_T_side_sparse_max=0)
      e[1].record()
      loss.backward()
      e[2].record()
      if timed and head == "A":
        ev.append(e)
      torch.autograd.backward([a, b], [a_d.grad, b_d.grad])
      if rk.on:
        ops.fold_branch_grads(params)              # .grad += the side view's gradients
        idist.all_reduce_grad_groups([params])     # SUM over ranks, one flat bucket (4.5 M parameters)
      opt.step()
    return loss
  for _ in range(args.warmup):
    step()
  rk.fence()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    last = step()
  rk.fence()
  dt = rk.max_over_ranks(time.perf_counter() - t0) / args.steps
  # kernel-level rooflines: HIP events around the contraction and around every conv launch, in extra
  # ONE-stream steps after the timed region (with two streams an event pair also spans whatever the
  # other view has in flight, and the event records cost GPU bubbles)
  n_inst = 0 if args.no_roofline else min(args.steps, 2)
  streams_timed = 2 if two_streams[0] else 1
  two_streams[0] = False
  conv = ConvTimer()
  conv.install()
  for _ in range(n_inst):
    gpu_hold(80.0)
    step(True)
  torch.cuda.synchronize()
  conv.uninstall()
  dp_report = rk.report()          # (collective: every rank)
  if n_inst == 0:
    if rk.rank == 0:
      print(json.dumps({"metric": "paired-images/sec, %s SegmentationNet10aTwoHead + IID_segmentation_loss_uncollapsed" % args.config,
                        "value": bn * world / dt, "unit": "paired-images/sec", "n_gpus": world, "steps": args.steps,
                        "warmup": args.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak",
                        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                        "config": {"workload": args.config, "streams": streams_timed, "final_loss": float(last.detach()),
                                   "parallelism": "dp%d" % world, "global_batch_pairs": bn * world,
                                   "data_parallel": dp_report}}))
    rk.close()
    return
  kA = c["k_A"]
  f_joint = 2.0 * kA * kA * (2 * T + 1) ** 2 * bn * sz * sz          # 2 k^2 (2T+1)^2 bn h w
  ms_f = sum(e[0].elapsed_time(e[1]) for e in ev) / len(ev)
  ms_b = sum(e[1].elapsed_time(e[2]) for e in ev) / len(ev)
  tf = 3.0 * f_joint / ((ms_f + ms_b) * 1e-3) / 1e12
  # read-once algorithmic bytes of the contraction: the two fp32 softmax maps, forward + 2 backward
  # passes, + the two gradient maps written
  abytes = (3 * 2 + 2) * 4.0 * bn * kA * sz * sz
  cs = conv.summary()
  out = {
    "metric": "paired-images/sec, %s SegmentationNet10aTwoHead + IID_segmentation_loss_uncollapsed" % args.config,
    "value": bn * world / dt, "unit": "paired-images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
    "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
    "dtype": "bf16", "data": "synthetic",
    "config": {"workload": "%s %dx%dx%d SegmentationNet10aTwoHead k_A %d / k_B %d, batch %d, T=%d, "
                           "mask density %.1f, one head-A step + one head-B step per batch pair "
                           "(segmentation_twohead.py train step), bf16 MFMA convs / fp32 head + loss"
                           % (args.config, sz, sz, c["in_ch"], kA, c["k_B"], bn, T, c["mask_p"]),
               "launch": "eager (python/ctypes)", "streams": streams_timed,
               "parallelism": "dp%d" % world, "global_batch_pairs": bn * world, "data_parallel": dp_report,
               "final_loss": float(last.detach())},
    "roofline": {"bound": "mfma", "kernel": "seg_joint_kernel + 2x seg_grad_kernel (P = sum x1(u+t) x2(u)^T over "
                                            "(2T+1)^2 shifts and its gradient; fp32 MFMA 16x16x4)",
                 "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS,
                 "traffic": None, "algorithmic_flops_per_launch": f_joint,
                 "algorithmic_bytes": abytes, "arithmetic_intensity_flop_per_byte": 3.0 * f_joint / abytes,
                 "joint_fwd_ms": ms_f, "grad_bwd_ms": ms_b,
                 "timed_in": "%d instrumented one-stream steps after the timed region" % n_inst,
                 "hbm_floor_ms": abytes / 8e12 * 1e3, "mfma_floor_ms": 3.0 * f_joint / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3},
  }
  if cs:
    out["roofline_conv"] = {"bound": "mfma", "kernel": "conv_igemm family (fwd + bwd-data), bf16 MFMA",
                            "achieved": cs["tflops"], "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": cs["tflops"] / BF16_PEAK_TFLOPS, "launches_timed": cs["launches"],
                            "kernel_ms_per_step": cs["total_ms"] / n_inst,
                            "timed_in": "%d instrumented one-stream steps after the timed region" % n_inst}
  if rk.rank == 0:
    print(json.dumps(out))
  rk.close()


C6_CONFIGS = {
  # BASELINE.json configs[0] on the GPU: MNIST 24x24 ClusterNet6cTwoHead, k_A 50 / k_B 10, batch 700, 5 sub-heads
  # (cluster_greyscale_twohead.py, commands.txt:30): one head-A step + one head-B step per batch
  "mnist6c": dict(bn=700, sz=24, in_ch=1, sobel=False, heads=("A", "B"), k=(50, 10)),
  # configs[2] per-GPU shape: CIFAR 24x24, --include_rgb (RGB + Sobel = 5 channels), ClusterNet6c, output_k 280
  # (commands.txt:41 trains with batch 2800 over 5 dataloaders; 700 here, as in round 1's measurement)
  "cifar6c": dict(bn=700, sz=24, in_ch=5, sobel=True, heads=(None,), k=(280,)),
}
# ClusterNet6c forward FLOPs per image at 24x24 (net6c.py:10-88, vgg.py:8-35: 5x5 convs 1/5->64 @24, 64->128
# @12, 128->256 @6, 256->512 @3): 2 * (576*25*c*64 + 144*25*64*128 + 36*25*128*256 + 9*25*256*512)
def _c6_flops(in_ch):
  return 2.0 * (576 * 25 * in_ch * 64 + 144 * 25 * 64 * 128 + 36 * 25 * 128 * 256 + 9 * 25 * 256 * 512)


def bench_6c(args):
  """`--config mnist6c|cifar6c`: the ClusterNet6c configs (VERDICT r2 missing #2).  The train step of the
  reference script (forward x2 -> IID_loss x 5 sub-heads -> backward -> Adam; per head for the two-head net)
  replayed as captured HIP graphs with the two views on two streams (iic_amd.graph.CapturedPairStep), as at
  the north star: at 24 x 24 the step is ~600 small launches and eager Python is launch-bound (round 1:
  10.1 ms per MNIST A+B step).  `roofline`: the conv family (HIP events, one-stream eager steps after the
  timed region) and the step's algorithmic rate (3 passes x 2 views x forward FLOPs)."""
  from iic_amd import archs
  from iic_amd.graph import CapturedPairStep
  from iic_amd.losses import IID_loss_heads
  from iic_amd.optim import Adam
  from iic_amd.transforms import sobel_process
  from iic_amd import dist as idist, ops
  c = C6_CONFIGS[args.config]
  rk = Ranks(args)
  dev, world = rk.dev, rk.world
  torch.manual_seed(0)
  two = len(c["heads"]) == 2
  if two:
    cfg = types.SimpleNamespace(in_channels=c["in_ch"], input_sz=c["sz"], batchnorm_track=True, num_sub_heads=SUB_HEADS,
                                output_k_A=c["k"][0], output_k_B=c["k"][1])
    net = archs.ClusterNet6cTwoHead(cfg).to(dev).train()
  else:
    cfg = types.SimpleNamespace(in_channels=c["in_ch"], input_sz=c["sz"], batchnorm_track=True, num_sub_heads=SUB_HEADS,
                                output_k=c["k"][0])
    net = archs.ClusterNet6c(cfg).to(dev).train()
  rk.broadcast(net)
  params = list(net.parameters())
  use_graph = not args.no_graph
  opt = Adam(params, lr=args.lr, capturable=use_graph)
  # N > 1 (BASELINE configs[2] is 8 GPUs by name; cluster_sobel.py:146 wraps the net in nn.DataParallel): every rank
  # owns `bn` pairs of its own (weak scaling), the raw joints of all sub-heads are all-reduced inside the loss, the
  # parameter gradients (5.5 M: one flat bucket) before the optimiser -- in the captured step both are capture cuts
  g = torch.Generator().manual_seed(rk.rank)
  bn, sz = c["bn"], c["sz"]
  raw_ch = c["in_ch"] - 1 if c["sobel"] else c["in_ch"]          # RGB + grey as the loaders emit it
  x = torch.rand(bn, raw_ch, sz, sz, generator=g).to(dev)
  xt = torch.clamp(torch.flip(x, dims=[3]) * 0.9 + 0.05, 0, 1).contiguous()

  def inp(t):
    return sobel_process(t, True) if c["sobel"] else t

  def fwd(t, head):
    return net.forward_packed(inp(t), head=head) if two else net.forward_packed(inp(t))

  def loss_fn(a, b):
    return IID_loss_heads(a, b, lamb=1.0)[0].mean()

  def finish():
    if rk.on:
      ops.fold_branch_grads(params)
      idist.all_reduce_grad_groups([params])
    opt.step()

  def eager_head(head):
    net.zero_grad(set_to_none=True)
    ops.clear_branch_grads()
    loss = loss_fn(fwd(x, head), fwd(xt, head))
    loss.backward()
    finish()
    return loss

  def eager_step():
    for h in c["heads"]:
      last = eager_head(h)
    return last
  if use_graph:
    caps = [CapturedPairStep((lambda h=h: fwd(x, h)), (lambda h=h: fwd(xt, h)), loss_fn, finish,
                             lambda: net.zero_grad(set_to_none=True), warmup=max(1, args.warmup))
            for h in c["heads"]]

    def run():
      for cap in caps:
        last = cap()
      return last
  else:
    run = eager_step
    for _ in range(args.warmup):
      eager_step()
  rk.fence()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    last = run()
  t_enq = time.perf_counter() - t0
  rk.fence()
  dt = rk.max_over_ranks(time.perf_counter() - t0) / args.steps
  n_inst = 0 if args.no_roofline else min(args.steps, 3)
  conv = ConvTimer()
  conv.install()
  for _ in range(n_inst):
    gpu_hold(40.0)
    eager_step()
  torch.cuda.synchronize()
  conv.uninstall()
  fl_step = len(c["heads"]) * 3 * 2 * bn * _c6_flops(c["in_ch"])
  out = {
    "metric": "paired-images/sec, %s" % ("MNIST 24x24 ClusterNet6cTwoHead+IID_loss (head-A step + head-B step)"
                                         if two else "CIFAR 24x24x5 ClusterNet6c+IID_loss"),
    "value": bn * world / dt, "unit": "paired-images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
    "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
    "data": "synthetic",
    "config": {"workload": "%s: %dx%dx%d, batch %d, 5 sub-heads, k %s, bf16 MFMA convs / fp32 first layer + heads + loss, "
                           "fused HIP Adam" % (args.config, sz, sz, c["in_ch"], bn, "/".join(str(k) for k in c["k"])),
               "launch": ("hip-graph replay: %d captured pair steps (%d linear graph segments each%s), the two views on two "
                          "streams" % (len(c["heads"]), 4 + sum(len(sg.items) - sg.cuts for sg in (caps[0].g_l, caps[0].g_opt)),
                                       ", %d collectives issued eagerly between them" % (caps[0].g_l.cuts + caps[0].g_opt.cuts)
                                       if rk.on else ""))
                         if use_graph else "eager (python/ctypes)",
               "parallelism": "dp%d" % world, "global_batch_pairs": bn * world, "data_parallel": rk.report(),
               "host_enqueue_ms_per_step": 1e3 * t_enq / args.steps, "final_loss": float(last.detach())},
  }
  cs = conv.summary() if n_inst else None
  if cs:
    out["roofline"] = {"bound": "mfma", "kernel": "conv_igemm family (5x5 fwd + bwd-data implicit GEMM, bf16 MFMA)",
                       "achieved": cs["tflops"], "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": cs["tflops"] / BF16_PEAK_TFLOPS, "traffic": None, "launches_timed": cs["launches"],
                       "avg_launch_us": cs["avg_us"], "kernel_ms_per_step": cs["total_ms"] / n_inst,
                       "timed_in": "%d instrumented one-stream eager steps after the timed region" % n_inst,
                       "step_algorithmic_tflops": fl_step / dt / 1e12,
                       "step_frac_of_peak": fl_step / dt / 1e12 / BF16_PEAK_TFLOPS}
  if rk.rank == 0:
    print(json.dumps(out))
  rk.close()


SECONDARY = [("mnist6c", []), ("cifar6c", []), ("potsdam3", ["--T", "1"]), ("coco3", [])]


def secondary_configs():
  """The other BASELINE.json configs at full size on this GPU, one sub-process each (`python bench.py --config ...`,
  a fresh device state per config), condensed to value / ms_per_step / roofline fractions -- so that the driver's
  record carries them.  The full lines are what `--config X` prints."""
  import subprocess
  res = {}
  for name, extra in SECONDARY:
    cmd = [sys.executable, os.path.abspath(__file__), "--config", name] + extra
    try:
      r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
      d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
      e = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
           "workload": d["config"]["workload"], "command": "python bench.py --config %s %s" % (name, " ".join(extra))}
      for k in ("roofline", "roofline_conv"):
        if k in d:
          e[k] = {q: d[k].get(q) for q in ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms_per_step",
                                           "step_frac_of_peak") if d[k].get(q) is not None}
      res[name] = e
    except Exception as e:      # noqa: BLE001  (a secondary measurement never takes the headline down)
      res[name] = {"error": "%s: %s" % (type(e).__name__, e)}
  return res


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--config", default="stl10_5g", choices=["stl10_5g"] + sorted(SEG_CONFIGS) + sorted(C6_CONFIGS),
                  help="stl10_5g = the BASELINE metric (default); potsdam3 / coco3 = secondary "
                       "measurements of the segmentation configs with their own roofline object")
  ap.add_argument("--T", type=int, default=None, help="half_T_side_dense override for the segmentation configs")
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="pairs per GPU (default 660)")
  ap.add_argument("--lr", type=float, default=1e-4,
                  help="Adam learning rate (timing ablations that produce wrong gradients use 0 to keep the weights sane)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-roofline", action="store_true")
  ap.add_argument("--no-graph", action="store_true",
                  help="issue every launch of the timed steps from Python (eager) instead of replaying "
                       "the step as one captured HIP graph (iic_amd.graph.CapturedStep; N=1 default)")
  ap.add_argument("--no-branch", action="store_true",
                  help="run the two views one after the other on one stream instead of as two "
                       "concurrent branches of the step graph (iic_amd.ops.branch)")
  ap.add_argument("--eager-branch", action="store_true",
                  help="(with --no-graph) eager launches, but the second view on a side stream "
                       "(iic_amd.ops.branch): what the two-stream overlap gives without graphs")
  ap.add_argument("--no-reference-api", action="store_true",
                  help="skip the second measurement through the reference's own call sequence "
                       "(net(x) -> list, IID_loss per sub-head, torch.optim.Adam; reported in config)")
  ap.add_argument("--strong", action="store_true",
                  help="N > 1: keep the GLOBAL batch at --pairs (default 660, the reference's batch_sz) and give "
                       "every rank pairs/N of it, instead of --pairs per rank (weak scaling, the default)")
  ap.add_argument("--no-secondary", action="store_true",
                  help="skip the other BASELINE.json configs (mnist6c, cifar6c, potsdam3 T=1, coco3), which the default "
                       "N=1 run measures in sub-processes after the headline and reports under `secondary`")
  ap.add_argument("--with-augment", action="store_true",
                  help="also build every step's batch inside the timed region with the GPU paired "
                       "augmentation (iic_amd.augment, SURVEY 8f rank 1) from a resident uint8 "
                       "dataset, as cluster_sobel.py:205-232 does from its dataloaders; the default "
                       "(off) is the metric's own timed region, which excludes data loading")
  args = ap.parse_args()
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    self_launch(args)          # (does not return)
  if args.config in C6_CONFIGS:
    return bench_6c(args)
  if args.config != "stl10_5g":
    if args.steps == 10 and args.warmup == 3:
      args.steps, args.warmup = 3, 1
    return bench_segmentation(args)

  rk = Ranks(args)
  world, rank, dev, dist_on = rk.world, rk.rank, rk.dev, rk.on
  if args.strong and world > 1:
    assert args.pairs % world == 0, "--strong: --pairs must divide by the number of ranks"
    args.pairs //= world

  from iic_amd import archs, dist as idist, ops
  from iic_amd.losses import IID_loss_heads
  from iic_amd.optim import Adam
  from iic_amd.transforms import sobel_process

  torch.manual_seed(0)
  cfg = types.SimpleNamespace(in_channels=2, input_sz=INPUT_SZ, batchnorm_track=True,
                              num_sub_heads=SUB_HEADS, output_k=OUTPUT_K)
  net = archs.ClusterNet5g(cfg).to(dev).train()
  rk.broadcast(net)   # identical weights on every rank
  # N = 1: the whole step (sobel -> 2 forwards -> loss -> backward -> Adam) is captured once in a
  # HIP graph and replayed -- same kernels, same arithmetic, one launch call per step instead of
  # ~1100 from Python.
  # N > 1: same graphs, cut at the collectives (issued eagerly between the segments); IIC_DIST_GRAPH=0
  # or IIC_DIST_OVERLAP=1 select the eager launch modes.
  dist_graph = dist_on and os.environ.get("IIC_DIST_GRAPH", "1") != "0" and os.environ.get("IIC_DIST_OVERLAP", "0") != "1"
  use_graph = (not dist_on or dist_graph) and not args.no_graph and not args.with_augment
  opt = Adam(net.parameters(), lr=args.lr, capturable=use_graph)
  # the second view (net(all_imgs_tf)) as a parallel graph branch: same kernels and arithmetic,
  # the tail of one view's launch is filled by the other view's next launch
  use_branch = use_graph and not args.no_branch
  # weak scaling: every rank owns `pairs` pairs (its shard of the global batch of pairs*world)
  imgs, imgs_tf = make_batch(args.pairs, INPUT_SZ, dev, seed=rank)
  params = list(net.parameters())
  groups = net.grad_groups()       # gradient buckets of the data-parallel step, in backward order
  assert sorted(id(p) for g in groups for p in g) == sorted(id(p) for p in params)
  # N > 1 with eager launches (IIC_DIST_GRAPH=0, or after a failed capture / self-check): by default
  # the two views run on two streams (measured at N = 1:
  # 42.0 -> 38.0 ms without graphs); the side view's gradients are folded into .grad after backward
  # and ONE bucketed SUM all-reduce follows.  IIC_DIST_OVERLAP=1 selects the round-1 mode instead:
  # one stream, gradient all-reduce overlapped with backward through post-accumulate hooks.
  reducer = None
  two_stream = args.eager_branch or (dist_on and os.environ.get("IIC_DIST_OVERLAP", "0") != "1")
  if dist_on and not two_stream:
    reducer = idist.GradReducer(params)

  aug = None
  if args.with_augment:
    import numpy as np
    from iic_amd.augment import PairedAugmenter
    g = torch.Generator().manual_seed(100 + rank)
    dataset = torch.randint(0, 256, (8192, INPUT_SZ, INPUT_SZ, 3), dtype=torch.uint8, generator=g).to(dev)
    aug = PairedAugmenter(dataset, 84, INPUT_SZ, False, seed=rank)     # cluster_sobel.py:52,76-77 defaults
    aug_state = {"pos": 0}

  def next_batch():
    # one iteration of the zipped loaders: dataloader_batch_sz = pairs / 3 base images, tf1 once
    # (replicated into the 3 slots) and 3 independent tf2 draws (cluster_sobel.py:215-226)
    nb = args.pairs // 3
    idx = (aug_state["pos"] + np.arange(nb)) % 8192
    aug_state["pos"] = int((aug_state["pos"] + nb) % 8192)
    base, tfs = aug.paired_batch(idx, 3)
    return base.repeat(3, 1, 1, 1), torch.cat(tfs, 0)

  def loss_fn(xo, xt):
    loss, _ = IID_loss_heads(xo, xt, lamb=1.0)
    return loss.mean()

  def step():
    # one stream, one view after the other (the eager path; N > 1; the instrumented steps)
    net.zero_grad(set_to_none=True)
    ops.clear_branch_grads()
    bi, bt = next_batch() if aug is not None else (imgs, imgs_tf)
    if two_stream:
      with ops.branch():          # the view that comes first goes to the side stream
        xo = net.forward_packed(sobel_process(bi, False))
      xt = net.forward_packed(sobel_process(bt, False))
      ops.join()
    else:
      xo = net.forward_packed(sobel_process(bi, False))
      xt = net.forward_packed(sobel_process(bt, False))
    loss = loss_fn(xo, xt)
    loss.backward()
    if reducer is not None:
      reducer.finish()
    elif dist_on:
      ops.fold_branch_grads(params)          # .grad += the side view's gradients (one foreach add)
      idist.all_reduce_grad_groups(groups)   # one flat bucket per layer group, as the staged graphs do
    opt.step()
    return loss

  fence = rk.fence

  def finish():
    if dist_on:
      ops.fold_branch_grads(params)          # .grad += the side view's gradients (one foreach add)
      idist.all_reduce_grad_groups(groups)
    opt.step()

  run = step
  launch_mode = None
  replay_events = []          # staged N > 1 replay: host issue order of backward groups / bucket all-reduces
  if use_branch:
    from iic_amd.graph import CapturedPairStep
    try:
      force_staged = os.environ.get("IIC_FORCE_STAGED", "0") == "1"
      staged = (dist_on and os.environ.get("IIC_DIST_STAGED", "1") != "0") or force_staged
      if staged:
        # backward captured per layer group: a group's gradient bucket is all-reduced (third stream, async)
        # while the groups below it still run backward
        run = CapturedPairStep(lambda: net.forward_packed_taps(sobel_process(imgs, False)),
                               lambda: net.forward_packed_taps(sobel_process(imgs_tf, False)),
                               loss_fn, finish, lambda: net.zero_grad(set_to_none=True),
                               warmup=max(1, args.warmup), grad_groups=groups, opt_step=opt.step,
                               events=replay_events, force_staged=force_staged)
        nseg = 2 + 3 * len(groups) + len(run.g_l.items) - run.g_l.cuts + len(run.g_opt.items) - run.g_opt.cuts
        ncoll = run.g_l.cuts + run.g_opt.cuts + len(groups)
        launch_mode = ("hip-graph replay: %d linear graph segments, the two views on two streams, backward staged "
                       "in %d layer groups whose gradient buckets are all-reduced under the remaining backward, "
                       "%d collectives issued eagerly between the segments" % (nseg, len(groups), ncoll))
      else:
        run = CapturedPairStep(lambda: net.forward_packed(sobel_process(imgs, False)),
                               lambda: net.forward_packed(sobel_process(imgs_tf, False)),
                               loss_fn, finish, lambda: net.zero_grad(set_to_none=True),
                               warmup=max(1, args.warmup))       # warm-up steps are real steps
        launch_mode = "hip-graph replay: %d linear graph segments, the two views on two streams%s" % (
          4 + len(run.g_l.items) - run.g_l.cuts + len(run.g_opt.items) - run.g_opt.cuts,
          ", %d collectives issued eagerly between them" % (run.g_l.cuts + run.g_opt.cuts) if dist_on else "")
    except Exception as e:      # (N > 1 only: every rank issues the same collectives in either mode)
      if world == 1:
        raise
      sys.stderr.write("rank %d: graph capture failed (%r), eager launches instead\n" % (rank, e))
      run, use_graph, use_branch = step, False, False
      for _ in range(args.warmup):
        last = step()
    if world > 1 and use_branch and os.environ.get("IIC_DIST_GRAPH", "1") != "force":
      # self-check before the timed region: a replayed step must not be slower than an eager one
      # (seen only with two gloo ranks time-slicing ONE GPU, where the host-side collectives between
      # graph segments stall for seconds); all ranks take the same decision
      def timed2(fn):
        fence()
        t = time.perf_counter()
        for _ in range(2):
          fn()
        fence()
        return time.perf_counter() - t
      t_graph, t_eager = timed2(run), timed2(step)
      verdict = torch.tensor([1.0 if t_graph > 1.5 * t_eager else 0.0], device=dev)
      torch.distributed.all_reduce(verdict, op=torch.distributed.ReduceOp.MAX)
      if float(verdict) > 0:
        if rank == 0:
          sys.stderr.write("graph replay slower than eager launches here (%.1f vs %.1f ms per step): "
                           "eager launches\n" % (500 * t_graph, 500 * t_eager))
        run, use_graph, use_branch = step, False, False
  elif use_graph:
    from iic_amd.graph import CapturedStep
    run = CapturedStep(step, warmup=max(1, args.warmup))
  else:
    for _ in range(args.warmup):
      last = step()
  fence()
  t0 = time.perf_counter()
  c0 = time.thread_time()
  for _ in range(args.steps):
    last = run()
  t_enq = time.perf_counter() - t0      # host time to enqueue the K steps (no sync inside)
  c_enq = time.thread_time() - c0
  fence()
  dt = time.perf_counter() - t0
  # Roofline of the conv kernels: HIP events around every conv launch, on the launch stream, in
  # extra steps run right AFTER the timed region.  (Inside the timed region the 316 event pairs
  # per step cost ~6 ms/step of GPU bubbles -- measured -- and would distort `value`.)
  timer = None
  if not args.no_roofline:          # every rank runs the extra steps (they contain collectives)
    if rank == 0:
      timer = ConvTimer()
      timer.install()
    # The stream is parked in front of every instrumented step so that the host can enqueue the whole step behind it
    # (gpu_hold).  How long that takes depends on the host (20-40 ms on most boxes, > 80 ms on slow ones -- where a
    # fixed 80 ms hold read frac 0.25 for kernels whose rocprofv3 durations had not changed): a probe step measures it,
    # its records are dropped, the measured steps are parked for 1.5 x that.
    gpu_hold(80.0)
    t_h = time.perf_counter()
    step()
    hold_ms = min(400.0, max(80.0, 1.5e3 * (time.perf_counter() - t_h) + 20.0))
    fence()
    if timer is not None:      # the probe step's records are dropped
      del timer.records[:]
      for recs in timer.fam.values():
        del recs[:]
    for _ in range(min(args.steps, 3)):
      gpu_hold(hold_ms)
      step()
    fence()
    if timer is not None:
      timer.uninstall()
  # Secondary measurement, NOT the headline: exact replica de-duplication.  all_imgs is 3 copies of
  # pairs/3 base images (cluster_sobel.py:215-226; make_batch builds it that way), so the first view's trunk can run on
  # the unique third -- forward AND backward (features repeated, autograd sums the replicas' gradients; BatchNorm batch
  # statistics are replication-invariant).  Same losses and gradients (tests: test_replica_dedup_fp32_mode_vs_reference_
  # golden), a third of that view's conv work.  The metric's `value` keeps the reference's redundant forward.
  dedup = None
  if not dist_on and use_branch and use_graph and aug is None and not args.no_reference_api and args.pairs % 3 == 0:
    try:
      from iic_amd.archs import cluster as _cl
      from iic_amd.graph import CapturedPairStep

      def fwd_a_dedup():
        with _cl.replicated(3):
          return net.forward_packed(sobel_process(imgs, False))
      run_d = CapturedPairStep(fwd_a_dedup, lambda: net.forward_packed(sobel_process(imgs_tf, False)),
                               loss_fn, finish, lambda: net.zero_grad(set_to_none=True), warmup=2)
      fence()
      td = time.perf_counter()
      for _ in range(args.steps):
        last_d = run_d()
      fence()
      td = (time.perf_counter() - td) / args.steps
      dedup = {"paired_images_per_sec": args.pairs / td, "ms_per_step": 1e3 * td, "final_loss": float(last_d.detach()),
               "what": "opt-in `with iic_amd.archs.cluster.replicated(3): net(all_imgs)`: the first view's trunk runs once "
                       "on the %d unique images of the 3x replicated batch (forward and backward); exact, not the "
                       "metric's configuration" % (args.pairs // 3)}
    except Exception as e:      # noqa: BLE001  (a secondary measurement never takes the headline down)
      dedup = {"error": "%s: %s" % (type(e).__name__, e)}
  ref_api = None
  if not dist_on and not args.no_reference_api:
    ref_api = reference_api_rate(cfg, dev, imgs, imgs_tf, args.pairs, args.steps)
    # (rounds 3-4 also measured an eager two-stream variant; round 5 removed that mode: iic_amd/ops.py)
    ref_api["graphed"] = reference_api_rate(cfg, dev, imgs, imgs_tf, args.pairs, args.steps, auto_branch=True,
                                            graph_forward=True)
  loss_val = float(last.detach())
  # the two views' streams after the process group exists and has issued collectives: do they still run side by side,
  # and does a pending collective hold either of them up?  (HIP maps streams onto 4 hardware queues in creation order;
  # RCCL's stream is one more tenant: iic_amd.graph._collective_blocks)
  stream_check = None
  probe_all = use_branch
  if dist_on:      # (the probes issue collectives: either every rank runs them or none does -- a rank whose capture failed has no pair)
    flag = torch.tensor([1.0 if use_branch else 0.0], device=dev)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    probe_all = bool(flag.item() > 0.5)
  if dist_on and probe_all:
    from iic_amd import graph as igraph
    stream_check = {"pair_overlaps": igraph._streams_overlap(run.s1, run.s2),
                    "collective_blocks_stream_1": igraph._collective_blocks(run.s1),
                    "collective_blocks_stream_2": igraph._collective_blocks(run.s2)}
    if getattr(run, "s3", None) is not None:
      stream_check["fold_stream_beside_both"] = (igraph._streams_overlap(run.s1, run.s3) and
                                                 igraph._streams_overlap(run.s2, run.s3))
  dt = rk.max_over_ranks(dt)
  ms_per_step = 1e3 * dt / args.steps
  value = args.pairs * world / (dt / args.steps)

  dp_report = rk.report()          # (collective: every rank)
  if rank == 0:
    out = {
      "metric": "paired-images/sec, STL10 96x96 ClusterNet5g+IID_loss",
      "value": value, "unit": "paired-images/sec", "n_gpus": world, "steps": args.steps,
      "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
      "scaling": "strong" if (args.strong and world > 1) else "weak", "vs_baseline": None, "dtype": "bf16",
      "data": "synthetic" if aug is None else "synthetic uint8 dataset, GPU paired augmentation in the timed region",
      "config": {"workload": "STL10 96x96 ClusterNet5g IID+ (cluster_sobel.py train step), "
                             "batch %d pairs/GPU, 5 sub-heads, k=70, bf16 MFMA convs / fp32 "
                             "stem+heads+loss, fused HIP Adam" % args.pairs,
                 "global_batch_pairs": args.pairs * world, "input": "96x96x1 grey -> sobel 2ch",
                 "parallelism": "dp%d" % world, "final_loss": loss_val,
                 "streams": 2 if (use_branch or two_stream) else 1,
                 "launch": (launch_mode if use_branch else "hip-graph replay") if use_graph else "eager (python/ctypes)",
                 "replay_issue_order": ["%s%s" % (e[0], "" if len(e) == 1 else e[1])
                                        for e in replay_events[:2 * len(groups) + 1]] if (use_branch and replay_events) else None,
                 "host_enqueue_ms_per_step": 1e3 * t_enq / args.steps,
                 "host_cpu_ms_per_step": 1e3 * c_enq / args.steps},
    }
    if timer is not None:
      s = timer.summary()
      if s:
        out["roofline"] = {
          "bound": "mfma", "kernel": "conv_igemm_kernel + conv_igemm_bd_kernel (fwd + bwd-data implicit GEMM, bf16 MFMA)",
          "achieved": s["tflops"], "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": s["tflops"] / BF16_PEAK_TFLOPS, "traffic": pmc_traffic()[0],
          "traffic_source": pmc_traffic()[1],
          "traffic_unit": "HBM bytes per launch (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/pmc_traffic.py)",
          "algorithmic_bytes_per_launch": s["alg_bytes"],
          "launches_timed": s["launches"], "avg_launch_us": s["avg_us"],
          "timed_in": "%d instrumented steps run after the timed region (same batch, same state)" % min(args.steps, 3),
          "kernel_ms_per_step": s["total_ms"] / min(args.steps, 3),
          "step_algorithmic_tflops": value / world * FLOP_PER_PAIR / 1e12,
          "step_frac_of_peak": value / world * FLOP_PER_PAIR / 1e12 / BF16_PEAK_TFLOPS,
          "families": timer.families(min(args.steps, 3)),
        }
        hb, hsrc = pmc_step_hbm()
        if hb:
          # the step is balanced between the matrix pipe and HBM -- both fractions.  Bytes: PMC (every kernel
          # of a step, separate FETCH_SIZE / WRITE_SIZE passes); time: this run's step
          out["roofline"]["hbm"] = {"bytes_per_step": hb, "source": hsrc, "TB_per_s": hb / (ms_per_step * 1e-3) / 1e12,
                                    "frac": hb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "algorithmic_bytes_per_step": 0.14e9 * args.pairs,
                                    "algorithmic_frac": 0.14e9 * args.pairs / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if ref_api is not None:
      out["config"]["reference_api"] = ref_api
    if dedup is not None:
      out["config"]["replica_dedup_opt_in"] = dedup
    if dist_on:
      out["config"]["data_parallel"] = dp_report
      out["config"]["data_parallel"]["streams_after_collectives"] = stream_check
    if not dist_on and not args.no_secondary and args.pairs == PAIRS_PER_GPU:
      out["secondary"] = secondary_configs()
    if not dist_on and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline()
      out["cpu_baseline"]["configs0_mnist"] = cpu_baseline_mnist()
    # wall time of this whole invocation (imports, captures, the timed region, the reference-API / secondary / CPU legs)
    out["elapsed_s"] = round(time.time() - T_PROCESS_START, 1)
    print(json.dumps(out))
  rk.close()


if __name__ == "__main__":
  main()
