"""Print the kernel timeline of the fastest step (= a hipGraph replay) found in a rocprofv3 rocpd database.
usage: python tools/timeline.py <results.db> [anchor-kernel-substring]
  rocprofv3 --kernel-trace -d out -o kt -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline"""
import sqlite3
import sys


def main(db_path, anchor="pm_forward_fused"):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 3:
        print("anchor not found")
        return
    period, k = min((rows[idx[j + 1]][1] - rows[idx[j]][1], j) for j in range(len(idx) - 1))
    a, b = idx[k], idx[k + 1]
    t0 = prev_end = rows[a][1]
    print("step period: %.1f us, %d kernels" % (period / 1e3, b - a))
    for name, s, e in rows[a:b]:
        print("%8.1f  dur %6.1f  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name.split("(")[0][-55:]))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main(*sys.argv[1:])
