#!/bin/bash
# Kernel trace of the SRGAN training step with time stamps: how much kernel time of different queues overlaps.
# Usage (through gpurun): bash tools/trace_overlap.sh <tag>
set -u
TAG=${1:-ov}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
RAW=/tmp/trace_$TAG
mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $RAW/train -o train -- python3 $R/tools/bench_train_step.py --steps 3 --warmup 2 > $OUT/${TAG}_under_rocprof.json 2> $OUT/${TAG}_trace.err || { tail -5 $OUT/${TAG}_trace.err; exit 1; }
F=$(find $RAW/train -name '*kernel_trace.csv' | head -1)
python3 - "$F" > $OUT/${TAG}_overlap.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"][:40]) for r in rows)
# last third of the trace = steady-state steps
t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * 2 // 3
iv = [x for x in iv if x[0] >= t0]
tot = sum(e - s for s, e, *_ in iv)
union, cur_s, cur_e = 0, None, None
for s, e, *_ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
qs = {}
for s, e, q, _ in iv: qs[q] = qs.get(q, 0) + (e - s)
print("kernels", len(iv), "span ms", span / 1e6, "sum of durations ms", tot / 1e6, "union (GPU busy) ms", union / 1e6, "overlapped ms", (tot - union) / 1e6)
print("per queue ms", {k: v / 1e6 for k, v in qs.items()})
PY
cat $OUT/${TAG}_overlap.txt
