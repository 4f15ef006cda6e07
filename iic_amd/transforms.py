"""Drop-in sobel_process (/root/reference/code/utils/cluster/transforms.py:47-96) on the
HIP kernel (iic_amd/csrc/stem.hip::sobel_kernel)."""
from . import ops


def sobel_process(imgs, include_rgb, using_IR=False):
  assert imgs.is_cuda, "sobel_process (HIP): device tensor required -- no CPU fallback"
  return ops.sobel(imgs.float(), include_rgb, using_IR)
