"""Merge rocprofv3 --pmc passes into ONE per-kernel JSON (profiles/kernel_counters.json), which bench.py reads for
`roofline.traffic` and the VALU-issue fraction of the compositing kernels.

Every pass is a directory holding *counter_collection.csv (one --pmc run with --kernel-trace only: separate runs per counter
group, as MI355X_MICROARCH.md prescribes — FETCH_SIZE and WRITE_SIZE each in its own pass).  Values are averaged per launch.
HBM read bytes get the guide's gfx950 correction: FETCH_SIZE is reported in KB and under-counts wide coalesced streams by
half, so `read_bytes_x2_corrected` = 2 x 1024 x FETCH_SIZE (an upper bound for gather-heavy kernels); WRITE_SIZE is KB.

usage: python tools/kernel_counters.py <out.json> <label> <pmc_dir> [<pmc_dir> ...]"""
import csv
import glob
import json
import sys
from collections import defaultdict


def short(name):
    k = name.split("(")[0].replace("void ", "").strip()
    k = k.split("::")[-1] if k.startswith("riggs::") else k
    return k.split("<")[0]


def main(out_json, label, *dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "riggs::" not in r["Kernel_Name"]:
                    continue
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    out = {"_label": label, "_note": "per-launch averages of rocprofv3 --pmc passes; FETCH_SIZE / WRITE_SIZE in KB as reported"}
    for k, cs in sorted(acc.items()):
        e = {c: round(v / n, 2) for c, (v, n) in sorted(cs.items())}
        e["launches_sampled"] = max(n for _, n in cs.values())
        if "FETCH_SIZE" in e:
            e["read_bytes_as_reported"] = int(e["FETCH_SIZE"] * 1024)
            e["read_bytes_x2_corrected"] = int(2 * e["FETCH_SIZE"] * 1024)
        if "WRITE_SIZE" in e:
            e["write_bytes"] = int(e["WRITE_SIZE"] * 1024)
        out[k] = e
    json.dump(out, open(out_json, "w"), indent=1)
    print("wrote", out_json, len(out) - 2, "kernels")


if __name__ == "__main__":
    main(*sys.argv[1:])
