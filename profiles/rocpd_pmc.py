"""Per-kernel average of one PMC counter from a rocprofv3 (rocpd sqlite) run.  usage: rocpd_pmc.py <db> [name-filter]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. it
reads HALF of a wide coalesced stream (MI355X_MICROARCH.md, HBM section) -- both raw and x2 are printed."""
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                      "group by name, counter_name order by 3*4 desc").fetchall()
    return rows


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else "ngp"
    print("%-70s %-12s %7s %14s %10s" % ("kernel", "counter", "calls", "avg KiB", "avg us"))
    for name, cname, n, avg, dur in per_kernel(sys.argv[1]):
        if flt in name:
            print("%-70s %-12s %7d %14.1f %10.1f" % (name[:70], cname, n, avg, dur / 1e3))
