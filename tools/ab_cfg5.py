"""cfg5 (960x540 frame, 5 strips): all strips as one batch (chain per strip, ONE fused tail) against one call per strip,
and against the launch-per-block plan; alternating, one box."""
import sys, os, json
sys.path.insert(0, "" + os.path.dirname(os.path.abspath(__file__)) + ""); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_variants as V
from srmi import engine
for _ in range(2):
    print(json.dumps(V.cfg5(steps=20, warmup=3)))
    print(json.dumps(V.cfg5(steps=20, warmup=3, per_strip=True)))
engine.set_fuse_chain(False)
print(json.dumps(V.cfg5(steps=20, warmup=3)))
engine.set_fuse_chain(True)
