cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_vgg.py tests/test_gpu_seg_net.py tests/test_gpu_graph.py tests/test_gpu_graphed.py -m gpu -x -q > gpurun_out/r06_tests_b.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r06_tests_b.log
for c in "potsdam3 --T 1" "coco3" "mnist6c" "cifar6c"; do timeout 300 python bench.py --config $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], round(d['ms_per_step'],3), 'ms', round(d['value'],1))"; done
