"""Per hardware queue, when its matrix-bound kernels run inside the last step of a rocprofv3 --kernel-trace CSV: start / end of
its forward-phase and backward-phase work (ms from the step's first kernel), to see whether two views really overlap.
    python tools/queue_spans.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(fs[0])))
ends = [i for i, r in enumerate(rows) if "adam_dev_kernel" in r[2] or r[2].startswith("adam_kernel")]
bounds = [i for j, i in enumerate(ends) if j + 1 == len(ends) or ends[j + 1] - i > 50]
ks = rows[bounds[-2] + 1:bounds[-1] + 1]
t0 = ks[0][0]
tj = min(k[0] for k in ks if "iid_joint" in k[2])
print("step: %.2f ms, loss starts at %.2f ms" % ((max(k[1] for k in ks) - t0) / 1e6, (tj - t0) / 1e6))
for q in sorted(set(k[3] for k in ks)):
  for name, sel in (("forward", lambda k: k[0] < tj), ("backward", lambda k: k[0] >= tj)):
    m = [k for k in ks if k[3] == q and sel(k) and ("conv_" in k[2] or "bn_" in k[2])]
    if m:
      busy = sum(k[1] - k[0] for k in m)
      print("  queue %s %-8s conv/bn kernels %4d: first starts %7.2f ms, last ends %7.2f ms, busy %.2f ms" % (
        q, name, len(m), (m[0][0] - t0) / 1e6, (max(k[1] for k in m) - t0) / 1e6, busy / 1e6))
if len(sys.argv) > 2:      # the first kernels of every queue in the step (start ms, duration us, name)
  for q in sorted(set(k[3] for k in ks)):
    print("queue %s:" % q)
    for k in [k for k in ks if k[3] == q][:int(sys.argv[2])]:
      print("   %8.3f ms  %7.1f us  %s" % ((k[0] - t0) / 1e6, (k[1] - k[0]) / 1e3, k[2][:70]))
