"""Where the waves of each kernel spend their time: one rocprofv3 SQ counter pass (8 SQ slots)
of one train step, summarised per kernel.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
      SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o p -- \
      python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
  python tools/pmc_stalls.py gpurun_out/pmc_sq profiles/r01_v7_kernel_stats.txt profiles/r01_pmc_stalls.txt

Reading (MI355X_MICROARCH.md "rocprofv3 PMC slots"): WAIT_ANY (parked on s_waitcnt / barrier) +
WAIT_INST_ANY (issue stalls: MFMA dependencies, busy pipes; WAIT_INST_LDS is a sub-bucket) +
ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES
is in cycles summed over SIMDs, so MFMA pipe utilisation = busy / (1024 SIMDs x kernel time x
clock) -- the kernel time is taken from the un-instrumented kernel statistics (second argument),
PMC passes perturb it.  LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = share of LDS cycles lost to conflicts.
"""
import collections
import csv
import glob
import os
import re
import sys

CLOCK_HZ = 2.4e9
SIMDS = 256 * 4


def norm(name):
  return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def main():
  d, stats_txt, out = sys.argv[1:4]
  f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
  assert f, "no counter_collection.csv under %s" % d
  per = collections.defaultdict(lambda: collections.defaultdict(float))
  launches = collections.defaultdict(set)
  for r in csv.DictReader(open(f[0])):
    k = norm(r["Kernel_Name"])
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
  avg_us = {}
  for line in open(stats_txt):
    if line.startswith("#") or line.startswith("kernel"):
      continue
    m = re.match(r"(.{100}) +(\d+) +([\d.]+) +([\d.]+) +([\d.]+)", line)
    if m:
      avg_us.setdefault(norm(m.group(1)), float(m.group(5)))
  rows = []
  for k, c in per.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
      continue
    n = len(launches[k])
    # the statistics file truncates names to 100 characters: match on the common prefix
    us = next((v for kk, v in avg_us.items() if kk[:60] == k[:60]), None)
    util = None
    if us:
      util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n / (SIMDS * us * 1e-6 * CLOCK_HZ)
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    rows.append((wc, k, n, us, util, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
                 c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                 (c.get("SQ_LDS_BANK_CONFLICT", 0) / lds) if lds > 0 else 0.0))
  rows.sort(reverse=True)
  with open(out, "w") as o:
    o.write("# %s\n# counters: one SQ pass of bench.py --steps 1 --warmup 1 (2 steps); kernel time from %s\n"
            % (__doc__.strip().splitlines()[0], os.path.basename(stats_txt)))
    o.write("# shares are of SQ_WAVE_CYCLES; mfma_util = MFMA busy cycles / (1024 SIMDs x un-instrumented kernel time x 2.4 GHz)\n")
    o.write("%-62s %6s %8s %9s %9s %9s %9s %9s %9s\n" % ("kernel", "calls", "avg_us", "mfma_util", "wait_any",
                                                        "wait_inst", "(lds)", "active", "lds_confl"))
    for (_, k, n, us, util, wa, wi, wl, ac, lc) in rows[:24]:
      o.write("%-62s %6d %8s %9s %9.3f %9.3f %9.3f %9.3f %9.3f\n"
              % (k[:62], n, "%.1f" % us if us else "-", "%.3f" % util if util is not None else "-", wa, wi, wl, ac, lc))
  print(open(out).read())


if __name__ == "__main__":
  main()
