# usage (inside gpurun): bash tools/wgrad_trace.sh <layer substring> [abl codes]  -> per-kernel average durations by ablation code
# (rocprofv3 --kernel-trace of tools/wgrad_pl_ab.py, planar asm variant only; the tool times 23 launches per code in order)
R=$GRAFT_REPO_ROOT; L="$1"; A="${2:-0,1,2,3,4}"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/wtr
WGRAD_ONLY="$L" WGRAD_VARIANTS=${WGRAD_TRACE_VARIANT:-2} WGRAD_ABL_VARIANT=${WGRAD_TRACE_VARIANT:-2} WGRAD_ABL=$A timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/wtr -o p -- python $R/tools/wgrad_pl_ab.py > /tmp/wtr.log 2>&1
tail -3 /tmp/wtr.log
python3 - "$A" <<'P'
import csv, glob, sys
f = glob.glob('/tmp/wtr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
main = [r for r in rows if 'conv_wgrad_pl' in r['Kernel_Name']]
red = [r for r in rows if 'conv_wgrad_reduce' in r['Kernel_Name']]
codes = ['plain'] * 2 + sys.argv[1].split(',')
# the tool: 2 reps x (23 timed + 1 result) launches plain, then 23 per ablation code
n = len(main)
groups = [('plain', main[:48])] + [(c, main[48 + i * 23: 48 + (i + 1) * 23]) for i, c in enumerate(sys.argv[1].split(','))]
for name, g in groups:
  d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in g]
  if d:
    print("main kernel  abl %-6s n=%3d  avg %.1f us  min %.1f" % (name, len(d), sum(d) / len(d), min(d)))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in red]
print("reduce kernel n=%d avg %.1f us min %.1f" % (len(d), sum(d) / len(d), min(d)))
P
