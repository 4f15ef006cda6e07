# usage (inside gpurun): bash tools/roofline_check.sh  -> the live roofline objects of the bench configs (HIP events around the
# conv launches in the instrumented steps), twice for the north-star config to show the spread
cd $GRAFT_REPO_ROOT
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d.get("roofline",{}); print("%-10s ms/step %8.3f  frac %.4f  avg_launch_us %s  kernel_ms_per_step %s" % (sys.argv[1], d["ms_per_step"], r.get("frac", float("nan")), r.get("avg_launch_us"), r.get("kernel_ms_per_step")))'
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-reference-api --no-secondary 2>/dev/null | python -c "$P" north-star; done
timeout 300 python bench.py --config mnist6c 2>/dev/null | python -c "$P" mnist6c
timeout 300 python bench.py --config cifar6c 2>/dev/null | python -c "$P" cifar6c
timeout 300 python bench.py --config potsdam3 --T 1 2>/dev/null | python -c "$P" potsdam3
timeout 300 python bench.py --config coco3 2>/dev/null | python -c "$P" coco3
