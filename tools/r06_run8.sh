cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
IIC_HIP_LIB=dbg timeout 900 python -m pytest tests/test_gpu_seg_loss.py -m "gpu and hooks" -x -q 2>&1 | tail -6
