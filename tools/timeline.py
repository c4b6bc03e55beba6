"""Print the kernel timeline of the last full step found in a rocprofv3 rocpd database (kernel trace).
usage: python tools/timeline.py <results.db> [anchor-kernel-substring]"""
import sqlite3
import sys


def main(db_path, anchor="pm_embed"):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 3:
        print("anchor not found"); return
    a, b = idx[-3], idx[-2]
    t0 = rows[a][1]
    prev_end = t0
    print("step period: %.1f us, %d kernels" % ((rows[b][1] - t0) / 1e3, b - a))
    for name, s, e in rows[a:b]:
        short = name.split("(")[0][-60:]
        print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main(*sys.argv[1:])
