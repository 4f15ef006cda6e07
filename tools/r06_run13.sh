cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_vgg.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/gemm_x3_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gemm_x3_ab.txt
for c in cifar6c mnist6c; do timeout 300 python bench.py --config $c --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], round(d['ms_per_step'],3), 'ms', round(d['value'],1))"; done
