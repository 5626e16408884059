"""The reference-shaped loop (`bench.py --path modules`) on the GPU timeline, from a rocprofv3 --kernel-trace CSV: the last K steps, delimited by
the optimizer launch that ends each step (adam_multi_kernel of compat/apex).  Prints the step time on the GPU timeline, the per-kernel
launch averages, the GPU's idle time per step (no kernel of any queue running) and one step's timeline.
usage: python profiles/timed_region_modules_r06.py <dir with *kernel_trace.csv | csv> [K=40]"""
import collections
import csv
import glob
import os
import sys


def main(path, count):
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if "adam_multi_kernel" in r["Kernel_Name"]]
    t0, t1 = ends[-count - 1], ends[-1]
    per = collections.defaultdict(list)
    busy, cur_end, idle = 0, t0, 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t0 and e <= t1:
            per[r["Kernel_Name"]].append((e - s) / 1e3)
            if s > cur_end:
                idle += s - cur_end
            cur_end = max(cur_end, e)
    total = sum(sum(v) for v in per.values())
    print("last %d steps (%d optimizer launches in the trace): %.3f ms per step on the GPU timeline; sum of kernel durations %.1f us per step; "
          "GPU idle (no kernel running) %.1f us per step; %d launches per step" % (count, len(ends), (t1 - t0) / 1e6 / count, total / count, idle / 1e3 / count,
                                                                                  sum(len(v) for v in per.values()) // count))
    for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        short = name.split("(")[0].replace("void ", "")
        print("%-72s %5d launches  avg %8.2f us  per step %8.2f us" % (short[-72:], len(d), sum(d) / len(d), sum(d) / count))
    k0, k1 = ends[-6], ends[-5]
    print("\ntimeline of one step (us after the previous optimizer launch ended; %.1f us long):" % ((k1 - k0) / 1e3))
    last_end = {}
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Queue_Id", "?")
        if e > k0 and s <= k1 and s >= k0 - 200000:
            gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
            if s >= k0:
                print("%9.1f  + %6.1f us  queue %s  gap %7.1f  %s" % ((s - k0) / 1e3, (e - s) / 1e3, q, gap, r["Kernel_Name"].split("(")[0].replace("void ", "")[-64:]))
        last_end[q] = e


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
