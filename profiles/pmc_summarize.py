"""Per-kernel summary of rocprofv3 --pmc counters from `*_counter_collection.csv` files (one file per pass; several passes may be
given).  For every kernel whose name contains <name-filter> the LAST `--tail` dispatches are kept (the timed region of a bench run:
conditioning and warm-up launches come first) and each counter is reported as the mean over them.
usage: pmc_summarize.py [--tail N] <name-filter> <csv> [<csv> ...]   -> JSON {kernel: {counter: mean, "dispatches": n}}"""
import csv
import json
import sys
from collections import defaultdict


def short_name(name):
    s = name.split("(")[0].replace("void ", "").replace("ngp::", "")
    return s.strip()


def summarize(paths, flt, tail=None):
    out = {}
    for p in paths:
        per = defaultdict(lambda: defaultdict(list))            # kernel -> counter -> [(dispatch id, value)]
        for r in csv.DictReader(open(p)):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            if flt not in name:
                continue
            per[short_name(name)][r["Counter_Name"]].append((int(r.get("Dispatch_Id") or 0), float(r["Counter_Value"])))
        for k, cs in per.items():
            for c, vals in cs.items():
                vals = [v for _, v in sorted(vals)]
                if tail:
                    vals = vals[-tail:]
                d = out.setdefault(k, {})
                d[c] = sum(vals) / len(vals)
                d["dispatches"] = max(d.get("dispatches", 0), len(vals))
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    tail = None
    if args[0] == "--tail":
        tail = int(args[1]); args = args[2:]
    print(json.dumps(summarize(args[1:], args[0], tail), indent=1))
