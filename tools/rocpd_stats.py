"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a CSV of per-kernel stats.
usage: python tools/rocpd_stats.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                      "max(vgpr_count), max(lds_size), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPRs", "LDS", "WG"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), "%.1f" % r[3], int(r[4]), int(r[5]), "%.2f" % (100.0 * r[2] / tot), r[6], r[7], r[8]])
    print("wrote", out_path, "kernels:", len(rows), "total ms: %.3f" % (tot / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
