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
    idist.enable()
  install()
  target = sys.argv[1]
  sys.argv = sys.argv[1:]
  runpy.run_module(target, run_name="__main__", alter_sys=True)


if __name__ == "__main__":
  main()
