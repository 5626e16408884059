"""The TIMED region of bench.py from a rocprofv3 --kernel-trace CSV (round 2): the `count` optimisation steps that follow the
first `first` ones (conditioning + warm-up), delimited by the optimizer launch that ends each step.  Prints the step time on the
GPU timeline and the per-kernel launch averages inside the region -- what bench.py's own HIP-event averages must agree with
(kernels of the side stream -- the prefetched march -- overlap the others, so the durations sum to more than the step).
usage: python profiles/timed_region_r02.py <kernel_trace.csv> [first=1044] [count=200]"""
import collections
import csv
import sys


def main(path, first, count):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if "adam_all_kernel" in r["Kernel_Name"]]
    t0, t1 = ends[first - 1], ends[first + count - 1]
    t0 = min(int(r["Start_Timestamp"]) for r in rows if int(r["Start_Timestamp"]) >= t0)     # (the host pauses before the timed region)
    per = collections.defaultdict(list)
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t0 and e <= t1:
            per[r["Kernel_Name"]].append((e - s) / 1e3)
    total = sum(sum(v) for v in per.values())
    print("timed region (%d steps after the first %d): %.3f ms per step on the GPU timeline; sum of kernel "
          "durations %.1f us per step" % (count, first, (t1 - t0) / 1e6 / count, total / count))
    for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        short = name.split("(")[0].replace("void ", "")
        print("%-72s %5d launches  avg %8.2f us  per step %8.2f us" % (short[-72:], len(d), sum(d) / len(d), sum(d) / count))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 1044, int(a[3]) if len(a) > 3 else 200)
