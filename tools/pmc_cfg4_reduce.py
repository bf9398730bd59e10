#!/usr/bin/env python3
"""Reduce the rocprofv3 passes of tools/pmc_cfg4.sh (gpurun_out/<tag>_cfg4_*.csv) to one record per K-chunked conv launch
of the last RRDB block: kernel-trace duration, HBM traffic (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
gfx950, WRITE_SIZE exact; separate passes), MFMA-pipe busy share (SQ pass).
    python tools/pmc_cfg4_reduce.py <tag> [<tag> ...] > profiles/r3_cfg4_layers.json"""
import csv, glob, json, os, sys
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
LAYERS = ["64->32", "96->32", "128->32", "160->64"]


def rows(pattern):
    f = glob.glob(os.path.join(OUT, pattern))[0]
    return [r for r in csv.DictReader(open(f)) if "conv_cin" in r["Kernel_Name"]]


def one(tag):
    tr = rows(f"{tag}_cfg4_trace_t_kernel_trace.csv")[-4:]
    fe = rows(f"{tag}_cfg4_pmc_FETCH_SIZE_*_counter_collection.csv")[-4:]
    wr = rows(f"{tag}_cfg4_pmc_WRITE_SIZE_*_counter_collection.csv")[-4:]
    sq = rows(f"{tag}_cfg4_pmc_sq_*_counter_collection.csv")
    disp = sorted({int(r["Dispatch_Id"]) for r in sq})[-4:]
    out = {}
    for i, name in enumerate(LAYERS):
        c = {r["Counter_Name"]: float(r["Counter_Value"]) for r in sq if int(r["Dispatch_Id"]) == disp[i]}
        us = (int(tr[i]["End_Timestamp"]) - int(tr[i]["Start_Timestamp"])) / 1e3
        fb, wb = float(fe[i]["Counter_Value"]) * 2048, float(wr[i]["Counter_Value"]) * 1024
        out[name] = {"kernel": tr[i]["Kernel_Name"].split("(")[0].replace("void ", ""), "trace_us": round(us, 1),
                     "fetch_MB": round(fb / 1e6, 1), "write_MB": round(wb / 1e6, 1),
                     "hbm_TB_per_s": round((fb + wb) / us / 1e6, 2),
                     "mfma_pipe_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"] / 32, 3)}
    return out


print(json.dumps({"note": "cfg4 RRDB dense block (8 x 256 x 256), the four conv launches of the last block; each counter in its "
                          "own rocprofv3 run (tools/pmc_cfg4.sh); durations are the profiler's (isolated launches under the "
                          "tracer run longer than back to back)", **{t: one(t) for t in sys.argv[1:]}}, indent=1))
