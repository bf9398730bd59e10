#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes)
into per-kernel HBM bytes per launch.  Corrections for gfx950 (same guide): the counters are in KiB;
FETCH_SIZE reports exactly 1/2 of a wide coalesced read stream -> doubled; WRITE_SIZE is exact.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_pmc_traffic.json
"""
import collections, csv, glob, json, sys

def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg

fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in fetch:
    if "srmi::" not in k[0] or "pack" in k[0]:
        continue
    f = sum(fetch[k]) / len(fetch[k]) * 1024 * 2          # KiB -> B, x2 (gfx950 FETCH_SIZE correction)
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0]))) * 1024
    out[f"{k[0]}@grid{k[1]}"] = {"launches_sampled": len(fetch[k]), "fetch_bytes": round(f), "write_bytes": round(w),
                                 "hbm_bytes": round(f + w)}
json.dump({"note": "per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE exact; "
                   "workload: bench.py config 1 (16x3x96x96)", "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
