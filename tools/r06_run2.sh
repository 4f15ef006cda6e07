cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_dist.py -m gpu -q --durations=20 > gpurun_out/r06_rccl_tests.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r06_rccl_tests.log
timeout 600 python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err; echo "bench rc=$?"; tail -c 6000 gpurun_out/r06_bench_a.json
