import os, sys, json, subprocess
for dbg in [0,1,2,8,4,16,10,31]:
    env=dict(os.environ, SRMI_DBG=str(dbg))
    out=subprocess.run([sys.executable,"bench.py","--steps","20","--warmup","5","--no-cpu-baseline"],env=env,capture_output=True,text=True).stdout
    r=json.loads(out.strip().splitlines()[-1])
    L=r["roofline"]["layers"]
    print(dbg, {k.replace("_bf16","").replace("x96k3","").replace("x192k3",""):round(v["ms"]/v["launches"]*1000,1) for k,v in L.items() if k.startswith(("c64_bf16", "resblock"))}, flush=True)
