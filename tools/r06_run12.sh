cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -x -q -k "start_two_ranks" 2>&1 | tail -8
