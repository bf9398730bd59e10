"""Phase ablation of the Cin = 64 conv kernel (timing only; SRMI_DBG bits: 1 no staging, 2 no K loop, 4 no epilogue)."""
import os, sys, json, subprocess
for dbg in [0, 1024, 4, 1, 2, 6, 7, 0]:
    env = dict(os.environ, SRMI_DBG=str(dbg | int(os.environ.get('BASE_DBG', '0'))))
    out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], env=env,
                         capture_output=True, text=True).stdout
    r = json.loads(out.strip().splitlines()[-1])
    L = r["roofline"]["layers"]
    print(dbg, {k.split(":")[1]: round(v["ms"] / v["launches"] * 1000, 1) for k, v in L.items() if k.startswith("c64_bf16")}, flush=True)
