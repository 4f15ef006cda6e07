# usage (inside gpurun): bash tools/queue_balance.sh <tag> [bench.py args...]  -> gpurun_out/<tag>_queue_balance.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qb_$tag
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/qb_$tag -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-api --no-secondary "$@" > $R/gpurun_out/${tag}_qb.log 2>&1
cd $R
python tools/queue_balance.py /tmp/qb_$tag 2 7 | tee gpurun_out/${tag}_queue_balance.txt
