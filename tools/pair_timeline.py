"""What the chip does during one replayed pair step (two views on two streams): from a rocprofv3 --kernel-trace CSV,
for the last `steps` steps (a step ends with the last Adam kernel):
  * step span, time with 0 / 1 / >= 2 kernels in flight, the longest idle gaps with the kernels on either side;
  * which kernel classes overlap (matrix-bound: conv / wgrad; HBM-bound: BatchNorm passes; small: finalisers,
    reduces, loss, optimiser) -- time with (class on one queue, class on the other);
  * per class: sum of durations alone on the chip vs. sharing it.
    python tools/pair_timeline.py <dir with *kernel_trace.csv> [steps=3]  > profiles/rNN_pair_timeline.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
assert fs, "no kernel_trace.csv under " + d
rows = []
for r in csv.DictReader(open(fs[0])):
  rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()


def klass(name):
  n = name
  if "conv_igemm" in n or "conv_wgrad_dma" in n or "conv_wgrad_kernel" in n:
    return "mfma"
  if "stem_" in n and "patch_sums" not in n and "combine" not in n:
    return "stem"
  if n.startswith("bn_apply") or "bn_bwd_apply" in n or "bn_bwd_reduce" in n or "maxpool" in n or "avgpool" in n:
    return "hbm"
  return "small"


ends = [i for i, r in enumerate(rows) if "adam_dev_kernel" in r[2] or r[2].startswith("adam_kernel")]
# a step's optimiser = a run of adam launches: step boundaries = last adam of each run
bounds = [i for j, i in enumerate(ends) if j + 1 == len(ends) or ends[j + 1] - i > 50]
assert len(bounds) > nsteps, "trace holds %d steps" % len(bounds)
print("# %s: %d kernel dispatches, %d steps found, analysing the last %d" % (os.path.basename(fs[0]), len(rows), len(bounds), nsteps))
tot = defaultdict(float)
gaps_all = []
pair = defaultdict(float)
phase = defaultdict(float)
nomfma = defaultdict(float)
alone = defaultdict(float)
shared = defaultdict(float)
for s in range(len(bounds) - nsteps, len(bounds)):
  lo, hi = bounds[s - 1] + 1, bounds[s]
  ks = rows[lo:hi + 1]
  t0, t1 = ks[0][0], max(k[1] for k in ks)
  ev = []
  for i, k in enumerate(ks):
    ev.append((k[0], 1, i))
    ev.append((k[1], 0, i))
  ev.sort()
  active = set()
  prev = t0
  last_ended = None
  for t, kind, i in ev:
    dt = t - prev
    if dt > 0:
      n = len(active)
      tot["0" if n == 0 else "1" if n == 1 else "2+"] += dt
      if n == 0 and last_ended is not None:
        gaps_all.append((dt, ks[last_ended][2][:60], None, s))
      cl = sorted(klass(ks[j][2]) for j in active)
      if "mfma" not in cl:
        for j in active:
          nomfma[ks[j][2].split("(")[0][:48]] += dt / n
        if n == 0:
          nomfma["(nothing in flight)"] += dt
      if n == 1:
        alone[cl[0]] += dt
      elif n >= 2:
        pair[" + ".join(cl[:3])] += dt
        for c in set(cl):
          shared[c] += dt
    if kind == 1:
      if not active and gaps_all and gaps_all[-1][2] is None and gaps_all[-1][3] == s:
        g = gaps_all[-1]
        gaps_all[-1] = (g[0], g[1], ks[i][2][:60], s)
      active.add(i)
    else:
      active.discard(i)
      last_ended = i
    prev = t
  # phases: forward (to the first loss kernel), loss, backward (to the first optimiser kernel), optimiser
  tj = min([k[0] for k in ks if "iid_joint" in k[2]] or [t1])
  tg = max([k[1] for k in ks if "iid_grad" in k[2]] or [t1])
  ta = min([k[0] for k in ks if "adam" in k[2] and k[0] > tg] or [t1])
  for name, a, b in (("forward", t0, tj), ("loss", tj, tg), ("backward", tg, ta), ("optimiser", ta, t1)):
    phase[name, "span"] += b - a
    act = 0
    prevt = a
    m = 0
    evs = sorted([(max(k[0], a), 1, klass(k[2]) == "mfma") for k in ks if k[1] > a and k[0] < b] +
                 [(min(k[1], b), 0, klass(k[2]) == "mfma") for k in ks if k[1] > a and k[0] < b])
    for t, kind, is_m in evs:
      if t > prevt:
        phase[name, "mfma" if m > 0 else ("other" if act > 0 else "idle")] += t - prevt
        if m > 1:
          phase[name, "mfma2"] += t - prevt
      prevt = t
      act += 1 if kind else -1
      m += (1 if kind else -1) if is_m else 0
  tot["span"] += t1 - t0
  tot["kernel_sum"] += sum(k[1] - k[0] for k in ks)
  tot["n"] += len(ks)

qstat = defaultdict(lambda: defaultdict(int))
for s in range(len(bounds) - nsteps, len(bounds)):
  for k in rows[bounds[s - 1] + 1:bounds[s] + 1]:
    qstat[k[3]][klass(k[2]) + ("/wgrad" if "wgrad" in k[2] else "")] += 1
print("hardware queues (dispatches per step by class): " + " | ".join(
  "queue %s: %s" % (q, ", ".join("%s %d" % (c, n / nsteps) for c, n in sorted(d.items()))) for q, d in sorted(qstat.items())))
ms = lambda v: v / 1e6 / nsteps
print("per step: span %.2f ms | %d dispatches | sum of kernel durations %.2f ms" % (ms(tot["span"]), tot["n"] / nsteps, ms(tot["kernel_sum"])))
print("  nothing in flight %.2f ms | exactly one kernel %.2f ms | two or more %.2f ms" % (ms(tot["0"]), ms(tot["1"]), ms(tot["2+"])))
print("phases (ms/step): span | a matrix-bound kernel in flight (of which two or more) | only other kernels | idle")
for name in ("forward", "loss", "backward", "optimiser"):
  print("  %-10s %6.2f | %6.2f (%5.2f) | %5.2f | %5.2f" % (name, ms(phase[name, "span"]), ms(phase[name, "mfma"]), ms(phase[name, "mfma2"]),
                                                  ms(phase[name, "other"]), ms(phase[name, "idle"])))
print("one kernel alone on the chip, by class (ms/step): " + ", ".join("%s %.2f" % (k, ms(v)) for k, v in sorted(alone.items(), key=lambda x: -x[1])))
print("kernels sharing the chip, by class (ms/step):      " + ", ".join("%s %.2f" % (k, ms(v)) for k, v in sorted(shared.items(), key=lambda x: -x[1])))
print("overlap combinations (ms/step):")
for k, v in sorted(pair.items(), key=lambda x: -x[1])[:12]:
  print("  %-28s %.2f" % (k, ms(v)))
print("time with NO matrix-bound kernel in flight, by what runs instead (ms/step, shared intervals split evenly):")
for k, v in sorted(nomfma.items(), key=lambda x: -x[1])[:16]:
  print("  %-50s %.2f" % (k, ms(v)))
print("idle gaps: %d per step, %.2f ms per step; by size: <5us %.2f ms, 5-20us %.2f ms, >20us %.2f ms" % (
  len(gaps_all) / nsteps, ms(sum(g[0] for g in gaps_all)), ms(sum(g[0] for g in gaps_all if g[0] < 5000)),
  ms(sum(g[0] for g in gaps_all if 5000 <= g[0] < 20000)), ms(sum(g[0] for g in gaps_all if g[0] >= 20000))))
print("longest idle gaps (us, kernel before -> kernel after):")
for g in sorted(gaps_all, key=lambda x: -x[0])[:14]:
  print("  %7.1f  %s -> %s" % (g[0] / 1e3, g[1], g[2]))
