# One gpurun call: full GPU suite, the default bench line, and the rocprofv3 kernel statistics of
# the same command (copied into profiles/ by hand afterwards).
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final_gpu_tests.log
timeout 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; cat gpurun_out/final_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
if [ "$1" = "--others" ]; then timeout 400 python tools/other_configs.py > gpurun_out/other_configs.txt 2>&1; tail -12 gpurun_out/other_configs.txt; fi
