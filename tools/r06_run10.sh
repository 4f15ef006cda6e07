bash tools/r06_rccl_flaky.sh 2>&1 | tail -30
bash tools/r06_full_gpu.sh 2>&1 | tail -30
