#!/bin/bash
# per-dispatch durations / grids of kernels whose name matches $1, in one profiled bench run ($2.. = bench args)
pat=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trk
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trk -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-reference-api "$@" > /tmp/trk.log 2>&1
python - "$pat" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/trk/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[1] in r["Kernel_Name"]]
for r in rows[:40]:
  d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
  print("%-70s %9.1f us grid %s x %s x %s  wg %s lds %s" % (r["Kernel_Name"][:70], d, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?"))))
PY
