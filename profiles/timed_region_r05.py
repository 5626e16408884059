"""The TIMED region of `bench.py --steps K` from a rocprofv3 --kernel-trace CSV: the LAST K optimisation steps of the FIRST K + ...
steps of the process (the no-prefetch leg that follows re-runs K steps: with `--steps 200` the trace holds conditioning + warm-up +
200 timed + 1 + 200 no-prefetch steps), delimited by the optimizer launch that ends each step (adam_all_kernel; adam_mlp_pack_kernel during the deterministic conditioning of round 5, whose table optimizer runs entirely in the scatter-add).  Prints the step
time on the GPU timeline and the per-kernel launch averages inside the region -- what bench.py's own HIP-event averages must agree
with (the prefetched march runs on the side stream and overlaps the others, so the durations sum to more than the step).
usage: python profiles/timed_region_r05.py <dir with *kernel_trace.csv | csv> [K=200] [condition=1024] [warmup=20]"""
import collections
import csv
import glob
import os
import sys


def main(path, count, condition, warmup):
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if ("adam_all_kernel" in r["Kernel_Name"] or "adam_mlp_pack_kernel" in r["Kernel_Name"])]
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
    # one ordinary step of the region on the timeline (the 6th: no occupancy update in it): start relative to the previous step's
    # optimizer end, duration, queue, and the idle time of that queue in front of the launch
    k0, k1 = ends[first + 4], ends[first + 5]
    print("\ntimeline of timed step 6 (us after the previous optimizer launch ended; %.1f us long):" % ((k1 - k0) / 1e3))
    last_end = {}
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Queue_Id", "?")
        if e > k0 and s < k1 + 1:
            short = r["Kernel_Name"].split("(")[0].replace("void ", "")
            gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
            print("  %8.1f  +%7.1f us  queue %-4s gap %6.1f  %s" % ((s - k0) / 1e3, (e - s) / 1e3, q, gap, short[-60:]))
        last_end[q] = max(last_end.get(q, 0), e)


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 200, int(a[3]) if len(a) > 3 else 1024, int(a[4]) if len(a) > 4 else 20)
