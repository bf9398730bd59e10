"""Per-launch HIP-event times of the bf16 generator forward at several batch sizes (does the tail of the network,
whose 64ch@384^2 tensor is 18.9 MB per image, run faster per image when it fits the Infinity Cache?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd")]
import torch
import bench
for n in (8, 16, 24, 32, 48):
    bench.BATCH = n
    gen, x, _, _ = bench.build_generator(torch.device("cuda:0"), "bf16")
    with torch.no_grad():
        for _ in range(5):
            gen(x)
        plan, units, kms = bench.per_kernel_times(gen, x, reps=20)
    agg = {}
    for (label, i, cnt), t in zip(units, kms):
        d = plan.args[i].d
        key = f"{label}:{d.Cin}->{plan.args[i + cnt - 1].d.Cout}@{d.H}"
        agg[key] = agg.get(key, 0.0) + t
    print(n, "total %.1f us/img" % (kms.sum() * 1e3 / n), {k: round(v * 1e3 / n, 1) for k, v in agg.items()}, flush=True)
