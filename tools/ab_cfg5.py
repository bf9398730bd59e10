import sys, os, json
sys.path.insert(0, "" + os.path.dirname(os.path.abspath(__file__)) + ""); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_variants as V
from srmi import engine
print(json.dumps(V.cfg5(steps=20, warmup=3)))
engine.set_fuse_chain(False)
print(json.dumps(V.cfg5(steps=20, warmup=3)))
engine.set_fuse_chain(True)
print(json.dumps(V.cfg5(steps=20, warmup=3)))
