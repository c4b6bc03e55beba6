"""Average PMC counters per kernel from a rocprofv3 --pmc CSV (counter_collection.csv).
usage: python tools/pmc_summary.py <dir> [name-substring ...]"""
import csv
import glob
import sys
from collections import defaultdict


def main(d, *subs):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if subs and not any(s in k for s in subs):
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for k, cs in acc.items():
        print(k[-60:])
        for c, (v, n) in sorted(cs.items()):
            print("   %-28s %16.1f  (n=%d)" % (c, v / n, n))


if __name__ == "__main__":
    main(*sys.argv[1:])
