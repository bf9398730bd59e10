#!/bin/bash
# cfg4 (RRDB-23) and cfg2 with the persistent K-chunked conv (product library) against a product-flag library whose
# conv_cin_bf16.hip was compiled with -DSRMI_NO_CINP (launch-per-tile kernel everywhere), alternating on one box.
# Build the second library first:  hipcc ... -DSRMI_NO_CINP -c conv_cin_bf16.hip; g++ -shared ... -o
# super-resolution_amd/srmi/lib/libsrmi_nocinp.so.  Through gpurun.
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for l in libsrmi.so libsrmi_nocinp.so; do
    echo "== $l"
    SRMI_LIB=$R/super-resolution_amd/srmi/lib/$l timeout -k 10 200 python $R/tools/bench_variants.py --only cfg4,cfg2 --steps 10 --warmup 3 2>/dev/null | grep -o '"workload": "cfg[^,]*\|"ms_per_step": [0-9.]*' | paste - - || exit 1
  done
done
