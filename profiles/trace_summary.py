"""Per-kernel launch statistics of the TIMED region from a rocprofv3 --kernel-trace CSV: the first `skip` launches of every kernel
(bench.py's warm-up steps: a fresh model's rays do not terminate early, so the first scatter-adds run 3-5x longer) are left out,
which is what bench.py's own HIP-event averages cover.    usage: python profiles/trace_summary.py <kernel_trace.csv> [skip=16]"""
import collections
import csv
import sys


def main(path, skip):
    per = collections.defaultdict(list)
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%-64s %7s %10s %10s %10s   (first %d launches of each kernel skipped)" % ("kernel", "calls", "avg us", "min us", "max us", skip))
    for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1][skip:])):
        d = d[skip:] if len(d) > 2 * skip else d
        if name.startswith("ngp::") or "ngp" in name:
            short = name.split("(")[0].replace("void ", "")
            print("%-64s %7d %10.2f %10.2f %10.2f" % (short[-64:], len(d), sum(d) / len(d), min(d), max(d)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16)
