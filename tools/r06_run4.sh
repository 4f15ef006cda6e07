cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/pmc_cmd.sh r06_fc 1 $GRAFT_REPO_ROOT/tools/firstconv_perf.py 2>&1 | tail -14
