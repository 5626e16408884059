"""The TIMED region of `bench.py --steps K` from a rocprofv3 --kernel-trace CSV: the LAST K optimisation steps of the FIRST K + ...
steps of the process (the no-prefetch leg that follows re-runs K steps: with `--steps 200` the trace holds conditioning + warm-up +
200 timed + 1 + 200 no-prefetch steps), delimited by the optimizer launch (adam_all_kernel) that ends each step.  Prints the step
time on the GPU timeline and the per-kernel launch averages inside the region -- what bench.py's own HIP-event averages must agree
with (the prefetched march runs on the side stream and overlaps the others, so the durations sum to more than the step).
usage: python profiles/timed_region_r03.py <dir with *kernel_trace.csv | csv> [K=200] [condition=1024] [warmup=20]"""
import collections
import csv
import glob
import os
import sys


def main(path, count, condition, warmup):
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if "adam_all_kernel" in r["Kernel_Name"]]
    first = condition + warmup
    t0, t1 = ends[first - 1], ends[first + count - 1]
    t0 = min(int(r["Start_Timestamp"]) for r in rows if int(r["Start_Timestamp"]) >= t0)     # (the host pauses before the timed region)
    per = collections.defaultdict(list)
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t0 and e <= t1:
            per[r["Kernel_Name"]].append((e - s) / 1e3)
    total = sum(sum(v) for v in per.values())
    print("timed region (%d steps after the first %d; %d optimizer launches in the trace): %.3f ms per step on the GPU timeline; sum of "
          "kernel durations %.1f us per step" % (count, first, len(ends), (t1 - t0) / 1e6 / count, total / count))
    for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        short = name.split("(")[0].replace("void ", "")
        print("%-72s %5d launches  avg %8.2f us  per step %8.2f us" % (short[-72:], len(d), sum(d) / len(d), sum(d) / count))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 200, int(a[3]) if len(a) > 3 else 1024, int(a[4]) if len(a) > 4 else 20)
