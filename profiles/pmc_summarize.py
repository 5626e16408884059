"""Average rocprofv3 --pmc counters per kernel from `*_counter_collection.csv` files (one or several passes).
usage: pmc_summarize.py <name-filter> <csv> [<csv> ...]   -> JSON {kernel: {counter: mean per dispatch, "dispatches": n}}"""
import csv
import json
import sys
from collections import defaultdict


def summarize(paths, flt):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in paths:
        for r in csv.DictReader(open(p)):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            if flt not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace("ngp::", "")
            a = acc[short][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: dict({c: v[0] / v[1] for c, v in cs.items()}, dispatches=max(v[1] for v in cs.values())) for k, cs in acc.items()}


if __name__ == "__main__":
    print(json.dumps(summarize(sys.argv[2:], sys.argv[1]), indent=1))
