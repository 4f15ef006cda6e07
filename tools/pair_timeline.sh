# usage (inside gpurun): bash tools/pair_timeline.sh <tag> [bench.py args...]  -> gpurun_out/<tag>_pair_timeline.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$tag
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-api --no-secondary "$@" > $R/gpurun_out/${tag}_tl.log 2>&1
echo "prof rc=$?"; tail -1 $R/gpurun_out/${tag}_tl.log | cut -c1-200
cd $R
python tools/pair_timeline.py /tmp/tl_$tag 3 > gpurun_out/${tag}_pair_timeline.txt
cat gpurun_out/${tag}_pair_timeline.txt
