"""python -m iic_amd.run code.scripts.cluster.cluster_sobel --model_ind ... : run an UNCHANGED
reference training script (Python-2 source, read where it lies) on the HIP hot path.

  1. iic_amd.py2compat.enable(root): import hook that translates the reference's Python-2
     modules in memory (root = $IIC_REFERENCE or the sys.path / PYTHONPATH entry holding code/);
  2. iic_amd.install.install(strict=True): rebind the hot-path names (losses, sobel_process,
     architectures, evaluation matching, Adam) -- raises if any of them cannot be rebound;
  3. N > 1 (launched under torchrun, one process per GPU): `setup_distributed()`;
  4. execute the script module as __main__.

The scripts' two training forwards per step run on two HIP streams (iic_amd.ops.auto_branch:
net(all_imgs) on a side stream while net(all_imgs_tf) is enqueued; the loss joins) unless
IIC_AUTO_BRANCH=0.

Multi-GPU semantics for the unchanged scripts: every rank's loaders produce
the full batch; each architecture's TRAINING forward keeps this rank's contiguous rows (pairs
stay together: both views are sliced identically), evaluation forwards run the whole batch on
every rank; the losses all-reduce the raw joint; parameter gradients are SUM-all-reduced right
before every optimiser step; every rank constructs identical initial weights (the scripts seed
nothing: construction runs under a fixed forked RNG seed, $IIC_INIT_SEED).  The scripts write
config.pickle / config.txt / figures / checkpoints into <out_root>/<model_ind> with plain open() /
savefig / torch.save on every rank: ranks > 0 get `--out_root <out_root>/.rank<r>` (their own scratch
copy; with --restart it is first filled from rank 0's directory), and torch.save is a no-op there, so the
directory a user looks at is written by rank 0 alone.  python / numpy / torch RNGs are seeded identically
on every rank ($IIC_RUN_SEED, default 0), so the (unseeded) loaders shuffle and augment identically and the
row slices of all ranks together are exactly one global batch, and segmentation's per-step shift
(_draw_sparse_shift) is the same on every rank before the joints are all-reduced.  The script's
torch.nn.DataParallel degenerates to a plain call with one visible device.
"""
import os
import sys

torch_save_orig = None     # torch.save before setup_distributed() replaced it on ranks > 0

ARCH_NAMES = ("ClusterNet5g", "ClusterNet5gTwoHead", "ClusterNet6c", "ClusterNet6cTwoHead",
              "SegmentationNet10a", "SegmentationNet10aTwoHead")


def _seeded_constructor(cls, seed):
  """Subclass whose construction (weight init) runs under a fixed RNG seed, restoring the
  global RNG afterwards: identical initial parameters on every rank without communication
  (parameters are still on the CPU when the script constructs the net, cluster_sobel.py:140)."""
  import torch

  class Seeded(cls):
    def __init__(self, *a, **k):
      with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        super(Seeded, self).__init__(*a, **k)
  Seeded.__name__ = cls.__name__
  Seeded.__qualname__ = cls.__qualname__
  Seeded.__module__ = cls.__module__
  return Seeded


def setup_distributed(backend="nccl", ref_modules=("code.archs", "code.archs.cluster",
                                                   "code.archs.segmentation")):
  """Everything `main` does for WORLD_SIZE > 1, callable on its own (the world-size-2 gloo test
  drives it on CPU stand-ins).  torch.distributed must be initialised; install() must have run
  if reference modules are to be re-pointed at the seeded constructors."""
  import torch
  import torch.distributed as dist
  from torch.optim.optimizer import register_optimizer_step_pre_hook

  from . import dist as idist
  assert dist.is_initialized()
  idist.enable()
  idist.SHARD_INPUTS[0] = True

  def _allreduce_hook(opt, args, kwargs):
    idist.all_reduce_grads([p for g in opt.param_groups for p in g["params"]])
  handle = register_optimizer_step_pre_hook(_allreduce_hook)

  seed = int(os.environ.get("IIC_INIT_SEED", "0"))
  for mname in ref_modules:
    m = sys.modules.get(mname)
    if m is None:
      continue
    for n in ARCH_NAMES:
      cls = getattr(m, n, None)
      if isinstance(cls, type) and not getattr(cls, "_iic_seeded", False):
        s = _seeded_constructor(cls, seed)
        s._iic_seeded = True
        setattr(m, n, s)
  global torch_save_orig
  if torch_save_orig is None:
    torch_save_orig = torch.save
  if dist.get_rank() != 0:
    # identical replicas: one writer is enough (and concurrent writers would corrupt the files)
    torch.save = lambda *a, **k: None
  return handle


def per_rank_out_root(argv, rank, default_root=None):
  """argv with `--out_root` re-pointed at <out_root>/.rank<rank> for rank > 0 (returns (argv, rank-0 root,
  this rank's root)); an absent --out_root is left alone when no default is known."""
  argv = list(argv)
  root = default_root
  idx = None
  for i, a in enumerate(argv):
    if a == "--out_root" and i + 1 < len(argv):
      idx, root = i + 1, argv[i + 1]
    elif a.startswith("--out_root="):
      idx, root = i, a.split("=", 1)[1]
  if rank == 0 or root is None:
    return argv, root, root
  mine = os.path.join(root, ".rank%d" % rank)
  if idx is None:
    argv += ["--out_root", mine]
  elif argv[idx].startswith("--out_root="):
    argv[idx] = "--out_root=" + mine
  else:
    argv[idx] = mine
  return argv, root, mine


def _model_ind(argv):
  for i, a in enumerate(argv):
    if a == "--model_ind" and i + 1 < len(argv):
      return argv[i + 1]
    if a.startswith("--model_ind="):
      return a.split("=", 1)[1]
  return None


def _restart_epoch(argv):
  """Epoch a `--restart` run continues from (config.pickle of <out_root>/<model_ind>, written by the script every
  epoch), 0 otherwise: the per-run RNG seed is offset by it so that a restarted multi-rank run does not replay the
  shuffle / augmentation sequence of epoch 0 (every rank derives the same value)."""
  if "--restart" not in argv:
    return 0
  import pickle
  root = None
  for i, a in enumerate(argv):
    if a == "--out_root" and i + 1 < len(argv):
      root = argv[i + 1]
    elif a.startswith("--out_root="):
      root = a.split("=", 1)[1]
  ind = _model_ind(argv)
  if root is None or ind is None:
    return 0
  try:
    with open(os.path.join(root, ind, "config.pickle"), "rb") as f:
      return int(getattr(pickle.load(f), "last_epoch", 0)) + 1
  except Exception:      # noqa: BLE001  (no checkpoint yet: the script itself will complain)
    return 0


def main(argv=None):
  argv = list(sys.argv[1:] if argv is None else argv)
  if not argv:
    sys.exit("usage: python -m iic_amd.run <reference.script.module> [script args...]")
  from . import ops, py2compat
  from .install import install
  install(strict=True)
  # the two forwards of the scripts' train step on two streams (iic_amd.ops.auto_branch);
  # IIC_AUTO_BRANCH=0 keeps everything on one stream
  ops.AUTO_BRANCH[0] = os.environ.get("IIC_AUTO_BRANCH", "1") != "0"
  # ... and each of them (with its backward) replayed as a captured HIP graph once its shape has been seen
  # twice (iic_amd/graphed.py); IIC_GRAPH_FORWARD=0 keeps eager launches
  ops.GRAPH_FORWARD[0] = os.environ.get("IIC_GRAPH_FORWARD", "1") != "0"
  if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(os.environ.get("IIC_DIST_BACKEND", "nccl"))
    setup_distributed()
    import random
    import shutil
    import numpy as np
    seed = int(os.environ.get("IIC_RUN_SEED", "0")) + 1000003 * _restart_epoch(argv)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    rank = dist.get_rank()
    # (the scripts' default --out_root is the authors' cluster path: only an explicit one is re-pointed)
    argv, root0, mine = per_rank_out_root(argv, rank)
    if rank > 0 and root0 is not None:
      os.makedirs(mine, exist_ok=True)
      ind = _model_ind(argv)
      if "--restart" in argv and ind is not None and os.path.isdir(os.path.join(root0, ind)):
        shutil.copytree(os.path.join(root0, ind), os.path.join(mine, ind), dirs_exist_ok=True)
    dist.barrier()
  target = argv[0]
  sys.argv = argv
  py2compat.run_script(target)


if __name__ == "__main__":
  main()
