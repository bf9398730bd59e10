"""Seeded inputs of the teacher-forced single-layer fixtures (tests/golden/layers_cfg3.npz) - TEST INFRASTRUCTURE.
Shared by oracle/make_golden_layers.py (which runs the reference's modules on them in the build container) and
tests/test_layers_cfg3_gpu.py (which regenerates the same tensors on the GPU box; nothing here touches the reference)."""
import numpy as np


def draw(seed, shape, kind="normal", scale=1.0, lo=0.0, hi=1.0):
    """Seeded NumPy draw (PCG64), float32 - the GPU box regenerates the same tensors without the reference."""
    r = np.random.Generator(np.random.PCG64([int(seed), 0x1A7E5]))
    v = r.standard_normal(shape) * scale if kind == "normal" else r.uniform(lo, hi, shape)
    return np.ascontiguousarray(v, dtype=np.float32)


def layer_tensors(name):
    """Recipe of one case: every input tensor by name (shared with tests/test_layers_cfg3_gpu.py)."""
    spec = SPECS[name]
    cin, cout, k = spec["cin"], spec["cout"], spec["k"]
    s = spec["seed"]
    t = {"x": draw(s, spec["x"], scale=1.0), "dy": draw(s + 1, spec["y"], scale=1.0),
         "w": draw(s + 2, (cout, cin, k, k) if not spec.get("linear") else (cout, cin), scale=1.5 / (cin * k * k) ** 0.5),
         "b": draw(s + 3, (cout,), "uniform", lo=-0.2, hi=0.2)}
    if spec.get("bn"):
        t["gamma"] = draw(s + 4, (cout,), "uniform", lo=0.5, hi=1.5)
        t["beta"] = draw(s + 5, (cout,), "uniform", lo=-0.2, hi=0.2)
    return t


SPECS = {
    "g_body": dict(cin=64, cout=64, k=3, stride=1, bn=True, act="prelu", x=(16, 64, 24, 24), y=(16, 64, 24, 24), seed=100),
    "d_s2": dict(cin=128, cout=128, k=3, stride=2, bn=True, act="lrelu", x=(16, 128, 48, 48), y=(16, 128, 24, 24), seed=200),
    "vgg_256": dict(cin=256, cout=256, k=3, stride=1, bn=False, act="relu", x=(16, 256, 24, 24), y=(16, 256, 24, 24), seed=300),
    "d_linear": dict(cin=18432, cout=1024, k=1, stride=1, bn=False, act="lrelu", linear=True, x=(16, 512, 6, 6), y=(16, 1024),
                     seed=400),
}
