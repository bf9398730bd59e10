import os, sys, json, subprocess
for rb, dbg in [(24,0),(48,0),(96,0),(12,0),(24,2),(24,6),(48,6)]:
    env=dict(os.environ, SRMI_DBG=str(dbg), SRMI_THIN_RB=str(rb))
    out=subprocess.run([sys.executable,"bench.py","--steps","10","--warmup","3","--no-cpu-baseline"],env=env,capture_output=True,text=True).stdout
    r=json.loads(out.strip().splitlines()[-1])
    L=r["roofline"]["layers"]
    print("rb",rb,"dbg",dbg, {k.split(":")[0]:round(v["ms"]/v["launches"]*1000,1) for k,v in L.items() if "thin" in k}, flush=True)
