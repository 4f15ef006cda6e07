"""rocprofv3 --kernel-trace --stats CSV -> the per-kernel summary format kept under profiles/."""
import csv, glob, os, sys
d, steps, note = sys.argv[1], int(sys.argv[2]), " ".join(sys.argv[3:])
fs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
assert fs, "no kernel_stats.csv under " + d
rows = list(csv.DictReader(open(fs[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# " + note)
print("# total kernel time %.2f ms/step over %d steps (sum of kernel durations; concurrent kernels count twice)" % (tot / 1e6 / steps, steps))
print("%-100s %8s %12s %6s %9s" % ("kernel", "calls", "total_us", "pct", "avg_us"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
  t = float(r["TotalDurationNs"])
  print("%-100s %8d %12.1f %6.2f %9.1f" % (r["Name"][:100], int(r["Calls"]), t / 1e3, 100 * t / tot, t / 1e3 / int(r["Calls"])))
