"""python -m iic_amd.run code.scripts.cluster.cluster_sobel --model_ind ... : run an
UNCHANGED reference training script on the HIP hot path (one process per GPU; for N GPUs
launch under torchrun -- the script's torch.nn.DataParallel degenerates to a plain call with
one visible device and iic_amd.dist shards the batch by pair)."""
import os
import runpy
import sys


def main():
  if len(sys.argv) < 2:
    sys.exit("usage: python -m iic_amd.run <reference.script.module> [script args...]")
  from .install import install, py2_shims
  py2_shims()
  if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl")
    from . import dist as idist
    from .archs import cluster as _cl
    idist.enable()
    # every rank's data loaders produce the full batch (the scripts are unchanged): keep this
    # rank's contiguous rows (pairs stay together: both views are sliced identically) ...
    _cl.SHARD_INPUTS[0] = True
    # ... and SUM-all-reduce the parameter gradients right before any optimiser step.
    from torch.optim.optimizer import register_optimizer_step_pre_hook

    def _allreduce_hook(opt, args, kwargs):
      idist.all_reduce_grads([p for g in opt.param_groups for p in g["params"]])
    register_optimizer_step_pre_hook(_allreduce_hook)
  install()
  target = sys.argv[1]
  sys.argv = sys.argv[1:]
  runpy.run_module(target, run_name="__main__", alter_sys=True)


if __name__ == "__main__":
  main()
