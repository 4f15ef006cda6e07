"""When does each hardware queue finish its share of a replayed pair step?  From a rocprofv3 --kernel-trace CSV:
per step, the time the last kernel of each queue ends (optimiser kernels apart) and each queue's busy time.
    python tools/queue_balance.py <dir with *kernel_trace.csv> [first_step last_step]"""
import csv, glob, os, sys
d = sys.argv[1]
fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(fs[0]))]
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam_dev_kernel" in r[2] or r[2].startswith("adam_kernel")]
b = [i for j, i in enumerate(ends) if j + 1 == len(ends) or ends[j + 1] - i > 50]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(b) - 1
for k in range(lo, min(hi, len(b) - 1)):
  step = rows[b[k] + 1:b[k + 1] + 1]
  t0 = step[0][0]
  out = []
  for q in sorted(set(r[3] for r in step)):
    qs = [r for r in step if r[3] == q and "adam" not in r[2]]
    if len(qs) < 20:
      continue
    out.append("queue %s: %d dispatches, busy %.2f ms, done at %.2f ms" % (q, len(qs), sum(r[1] - r[0] for r in qs) / 1e6, (max(r[1] for r in qs) - t0) / 1e6))
  print("step %d span %.2f ms | %s" % (k, (step[-1][1] - t0) / 1e6, " | ".join(out)))
