#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (separate runs, as MI355X_MICROARCH.md prescribes) per kernel and launch.

    python tools/pmc_summary.py <fetch csv|dir> <write csv|dir> <out json> [<sq csv|dir>]

HBM traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of a wide coalesced read
stream -> doubled (same guide); WRITE_SIZE is exact.  SQ pass (optional): per kernel the MFMA pipe utilisation
SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (both summed over the shader engines' SQs by the profiler, the ratio is what
the guide's recipe reads) and LDS bank-conflict cycles per LDS-active cycle."""
import collections, csv, glob, json, os, sys


def load(path):
    f = path if os.path.isfile(path) else glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    for k in fetch:
        if "srmi::" not in k[0] or "pack" in k[0]:
            continue
        f = mean(fetch[k]["FETCH_SIZE"]) * 1024 * 2         # KiB -> B, x2 (gfx950 FETCH_SIZE correction)
        w = mean(write.get(k, {}).get("WRITE_SIZE", [])) * 1024
        out[f"{k[0]}@grid{k[1]}"] = {"launches_sampled": len(fetch[k]["FETCH_SIZE"]), "fetch_bytes": round(f),
                                     "write_bytes": round(w), "hbm_bytes": round(f + w)}
    doc = {"note": "per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE exact; "
                   "workload: bench.py config 1 (16x3x96x96), bf16 path then the fp32 path", "kernels": out}
    if len(sys.argv) > 4:
        sq = load(sys.argv[4])
        s = {}
        for k, c in sq.items():
            if "srmi::" not in k[0] or "pack" in k[0]:
                continue
            busy, mfma = mean(c["SQ_BUSY_CYCLES"]), mean(c["SQ_VALU_MFMA_BUSY_CYCLES"])
            lds_act, lds_conf = mean(c["SQ_ACTIVE_INST_LDS"]), mean(c["SQ_LDS_BANK_CONFLICT"])
            s[f"{k[0]}@grid{k[1]}"] = {
                "launches_sampled": len(c["SQ_BUSY_CYCLES"]), "SQ_BUSY_CYCLES": round(busy),
                "SQ_VALU_MFMA_BUSY_CYCLES": round(mfma), "mfma_busy_ratio": round(mfma / busy, 4) if busy else None,
                "mfma_pipe_busy": round(mfma / busy / 32.0, 4) if busy else None,
                "SQ_WAVES": round(mean(c["SQ_WAVES"])), "SQ_INSTS_VALU": round(mean(c["SQ_INSTS_VALU"])),
                "SQ_INSTS_LDS": round(mean(c["SQ_INSTS_LDS"])), "SQ_ACTIVE_INST_LDS": round(lds_act),
                "SQ_LDS_BANK_CONFLICT": round(lds_conf),
                "lds_conflict_per_active": round(lds_conf / lds_act, 4) if lds_act else None}
        doc["sq"] = s
        doc["sq_note"] = ("one --pmc pass of 7 SQ counters.  SQ_VALU_MFMA_BUSY_CYCLES is the sum over the 1024 SIMDs of the "
                          "cycles their MFMA pipe was busy (16 per 16x16x32 bf16 MFMA: it reproduces the kernel's MFMA "
                          "count), SQ_BUSY_CYCLES the sum over the 32 shader engines of the cycles any wave was resident "
                          "(= kernel duration x the clock under load, 1.98 GHz here), so mfma_pipe_busy = ratio x 32 / 1024 "
                          "is the share of the kernel's cycles an average SIMD's MFMA pipe was busy")
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
