#!/bin/bash
# Ablation of conv_thin_kernel on the headline workload (investigation only): SRMI_DBG bits 2 = no MFMAs, 4 = no
# activation math, 8 = no output stores; bits 23-25 = cache policy of its loads / stores (api.hip).  THIN_ABLATE="0 8 ..."
# overrides the list of SRMI_DBG values.  Prints bench.py's per-layer times (ms per launch) and ms per step.
set -e
for d in ${THIN_ABLATE:-0 2 4 8 10 14}; do
  echo -n "SRMI_DBG=$d  "
  SRMI_DBG=$d timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k.split(':')[1].replace('x96k3','').replace('x192k3','').replace('x384k3',''):v['ms'] for k,v in d['roofline']['layers'].items()}, d['ms_per_step'])"
done
