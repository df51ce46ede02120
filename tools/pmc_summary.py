#!/usr/bin/env python
"""Summarise rocprofv3 --pmc output (…_counter_collection.csv files under a directory): per kernel (short name) and counter,
dispatches and mean value per dispatch; plus mean duration.  usage: python tools/pmc_summary.py DIR [name-filter]"""
import csv, glob, os, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(n):
    m = re.search(r"conv_fwd_kernel<(.*?)>\(", n)
    if m:
        f = [x.strip() for x in m.group(1).split(",")]
        f = [re.sub(r"\(.*?\)", "", x) for x in f]
        return "conv<" + ",".join(f[:1] + f[1:]) + ">"
    n = re.sub(r"\(.*", "", n)
    return n[-70:]


def main(d, flt=None):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt and flt not in k:
                continue
            a = acc[(k, r["Counter_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for (k, c), (n, v, t) in sorted(acc.items()):
        print(f"{k:90s} {c:24s} n={n:4d} mean={v / n:16.1f} dur_us={t / n:10.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
