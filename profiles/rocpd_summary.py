"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / % -- the `--stats` table.
usage: python profiles/rocpd_summary.py <results.db> [skip_first_n_dispatches_fraction]"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = cur.execute("select min(start), max(end) from kernels").fetchone()
    print("kernel-busy total %.3f ms over a %.3f ms span; %d distinct kernels, %d dispatches" % (
        total / 1e6, (span[1] - span[0]) / 1e6, len(rows), sum(r[1] for r in rows)))
    print("%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:top]:
        print("%-90s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                                                            100.0 * r[2] / total))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
