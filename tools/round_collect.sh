# One gpurun call: round-end measurement artefacts of round <tag> (default r04): kernel statistics of the default
# (two streams, graphs) and the sequential eager launch modes, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in
# separate passes), one SQ pass, the default bench line and the 6c lines.  Results land in gpurun_out/; copy the
# summaries into profiles/ afterwards.   usage: bash tools/round_collect.sh [tag]
T=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 400 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "bench rc=$?"; tail -1 gpurun_out/${T}_bench_default.json | cut -c1-260
timeout 200 python bench.py --config mnist6c > gpurun_out/${T}_bench_mnist6c.json 2>/dev/null; tail -1 gpurun_out/${T}_bench_mnist6c.json | cut -c1-200
timeout 200 python bench.py --config cifar6c > gpurun_out/${T}_bench_cifar6c.json 2>/dev/null; tail -1 gpurun_out/${T}_bench_cifar6c.json | cut -c1-200
bash $R/tools/prof_step.sh ${T}_final_pair | tail -3
bash $R/tools/prof_step.sh ${T}_final_seq --no-branch --no-graph | tail -3
timeout 300 python bench.py --config potsdam3 --T 1 > gpurun_out/${T}_bench_seg_potsdam3.json 2>/dev/null; tail -1 gpurun_out/${T}_bench_seg_potsdam3.json | cut -c1-160
timeout 300 python bench.py --config potsdam3 --T 10 > gpurun_out/${T}_bench_seg_potsdam3T10.json 2>/dev/null; tail -1 gpurun_out/${T}_bench_seg_potsdam3T10.json | cut -c1-160
timeout 300 python bench.py --config coco3 > gpurun_out/${T}_bench_seg_coco3.json 2>/dev/null; tail -1 gpurun_out/${T}_bench_seg_coco3.json | cut -c1-160
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference-api --no-secondary --no-branch --no-graph"
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_sq
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o p -- $B > $R/gpurun_out/pmc_f.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o p -- $B > $R/gpurun_out/pmc_w.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_sq -o p -- $B > $R/gpurun_out/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $R
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w gpurun_out/${T}_pmc_traffic.json | tail -12
python tools/pmc_stalls.py /tmp/pmc_sq gpurun_out/${T}_final_seq_kernel_stats.txt gpurun_out/${T}_pmc_stalls.txt | head -24
