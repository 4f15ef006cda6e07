cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vgg.py -m gpu -x -q -k "firstconv" > gpurun_out/r06_fc_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06_fc_tests.log
timeout 600 python tools/firstconv_perf.py > gpurun_out/r06_firstconv_perf.txt 2>&1; cat gpurun_out/r06_firstconv_perf.txt | tail -8
