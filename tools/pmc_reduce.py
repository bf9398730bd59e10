#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter: hundreds of KB) to one row per
(kernel, grid size, counter) with the MEAN counter value and the number of launches - the form kept under profiles/.

    python tools/pmc_reduce.py <counter_collection.csv> <out.csv>"""
import collections, csv, sys


def main():
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        k = (name, int(r["Grid_Size"]), r["Counter_Name"])
        s = agg.setdefault(k, [0.0, 0])
        s[0] += float(r["Counter_Value"]); s[1] += 1
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Launches"])
        for (name, grid, ctr), (tot, n) in agg.items():
            w.writerow([name, grid, ctr, tot / n, n])


if __name__ == "__main__":
    main()
