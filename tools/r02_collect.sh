# One gpurun call: round-2 measurement artefacts (kernel stats of the default and the sequential
# launch modes, PMC traffic passes, SQ pass, segmentation stats).  Results land in gpurun_out/.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
bash $R/tools/prof_step.sh r02_final_pair
bash $R/tools/prof_step.sh r02_final_seq --no-branch --no-graph
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference-api --no-branch --no-graph"
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_sq
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o p -- $B > $R/gpurun_out/pmc_f.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o p -- $B > $R/gpurun_out/pmc_w.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_sq -o p -- $B > $R/gpurun_out/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $R
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w gpurun_out/r02_pmc_traffic.json | tail -14
python tools/pmc_stalls.py /tmp/pmc_sq gpurun_out/r02_final_seq_kernel_stats.txt gpurun_out/r02_pmc_stalls.txt | head -24
