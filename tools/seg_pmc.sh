#!/bin/bash
# SQ counter pass over the segmentation-loss kernels alone (tools/seg_kernel_perf.py): $1 = config
mkdir -p gpurun_out
cfg=${1:-potsdam}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_seg_$cfg
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_seg_$cfg -o p -- python $GRAFT_REPO_ROOT/tools/seg_kernel_perf.py $cfg 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_seg_$cfg.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, re
f = glob.glob("gpurun_out/pmc_seg_$cfg/**/*counter_collection.csv", recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
  k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
  per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in per.items():
  wc = c["SQ_WAVE_CYCLES"] or 1
  print("%-60s n=%d  wait_any %.3f  wait_inst %.3f (lds %.3f)  active %.3f  lds_conflict %.3f  mfma_busy_cycles/launch %.3e" % (
    k, len(n[k]), c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_INST_LDS"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc,
    c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_VALU_MFMA_BUSY_CYCLES"] / len(n[k])))
PY
