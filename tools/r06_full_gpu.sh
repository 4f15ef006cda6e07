cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r06_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -32 gpurun_out/r06_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
