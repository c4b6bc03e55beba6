"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected separately, as the
TCC slots require).  Counter values are KB per dispatch (summed over the XCDs' TCCs by rocprofv3); the read side
gets the x2 correction MI355X_MICROARCH.md prescribes for gfx950 wide coalesced streams (FETCH_SIZE under-counts
them by half), so treat it as an upper bound for gather-heavy kernels.
usage: python tools/hbm_traffic.py <fetch_dir> <write_dir> <out.csv> <out.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main(fetch_dir, write_dir, out_csv, out_json):
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, (0, 0))[0] + wr.get(k, (0, 0))[0])):
        if not k.startswith("riggs::"):
            continue
        f, n = fe.get(k, (0.0, 0))
        w, _ = wr.get(k, (0.0, 0))
        rows.append((k, n, f, 2 * f / 1024.0, w, w / 1024.0))
    with open(out_csv, "w", newline="") as fh:
        wtr = csv.writer(fh)
        wtr.writerow(["kernel", "launches", "FETCH_SIZE_avg_KB_as_reported", "FETCH_MB_x2_corrected", "WRITE_SIZE_avg_KB", "WRITE_MB"])
        for r in rows:
            wtr.writerow([r[0], r[1], "%.1f" % r[2], "%.2f" % r[3], "%.1f" % r[4], "%.2f" % r[5]])
    json.dump({r[0].split("::")[-1].split("<")[0]: {"read_bytes_x2_corrected": int(r[3] * 1e6 * 1.048576), "read_bytes_as_reported": int(r[2] * 1024),
                                                      "write_bytes": int(r[4] * 1024)} for r in rows}, open(out_json, "w"), indent=1)
    print("wrote", out_csv, out_json, len(rows), "kernels")


if __name__ == "__main__":
    main(*sys.argv[1:])
