#!/usr/bin/env python3
"""Secondary benchmarks (BASELINE configs 2 and 4): SRCNN on 32x1x384x384 and the ESRGAN-style RRDB
generator on 8x3x256x256, forward only, inputs resident in HBM.  One JSON line per model with ms/step, output MP/s,
algorithmic TFLOP/s and the kernels the plan selected.  Not the headline metric (bench.py is)."""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd")]

import torch


def run(name, model, x, out_px, flop_per_step, steps, warmup, compute, per_layer=False):
    model = model.to("cuda:0").eval().set_compute(compute)
    x = x.to("cuda:0")
    with torch.no_grad():
        for _ in range(warmup):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model(x)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
    plan = model._plans.get(model, x.shape[0], x.shape[2], x.shape[3], 1 if compute == "bf16" else 0)
    from srmi.engine import _Seq
    plan.args[0].x = x.data_ptr()
    units = _Seq.launches(plan.args, plan.n, plan.kernels)          # what srmi_conv2d_fwd_seq launches (fused pairs as one)
    rec = {"workload": name, "compute": compute, "ms_per_step": round(el * 1e3, 3),
           "out_mp_per_s": round(out_px / 1e6 / el, 1), "tflops": round(flop_per_step / el / 1e12, 1),
           "kernels": dict(collections.Counter(u[0] for u in units))}
    if per_layer:
        rec["layers"] = layer_table(plan, x)
        # the whole net against its compulsory traffic (input + output only: what ONE fused launch would move)
        io = x.numel() * 4 + out_px * 4
        rec["hbm_frac_compulsory"] = round(io / el / HBM_PEAK, 4)
        rec["hbm_frac_layerwise"] = round(sum(v["algorithmic_bytes"] for v in rec["layers"].values()) / el / HBM_PEAK, 4)
    return rec


HBM_PEAK = 8e12     # B/s, /opt/skills/guides/MI355X_MICROARCH.md


def layer_table(plan, x, reps=10):
    """Per-launch HIP-event time of a conv plan and its achieved HBM rate on the layer's own algorithmic bytes
    (input tensor + output tensor, each once)."""
    import ctypes as C
    from srmi import _lib as L
    from srmi import functional as Fn
    from srmi.engine import _Seq
    lib, st = L.load(), Fn.stream_ptr()
    plan.args[0].x = x.data_ptr()
    esz = {L.F32: 4, L.BF16: 2}
    units = _Seq.launches(plan.args, plan.n, plan.kernels)          # (label, first conv, count): fused pairs are ONE launch
    tot = [0.0] * len(units)
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(units) + 1)]
        ev[0].record()
        for u, (label, i, cnt) in enumerate(units):
            if cnt == 1:
                L.check(lib.srmi_conv2d_fwd(C.byref(plan.args[i]), st))
            elif cnt == 2:
                L.check(lib.srmi_conv2d_fwd_pair(C.byref(plan.args[i]), C.byref(plan.args[i + 1]), st))
            else:
                L.check(lib.srmi_conv2d_fwd_seq(C.byref(plan.args[i]), cnt, st))
            ev[u + 1].record()
        torch.cuda.synchronize()
        for u in range(len(units)):
            tot[u] += ev[u].elapsed_time(ev[u + 1]) * 1e-3 / reps
    out = {}
    for u, (label, i, cnt) in enumerate(units):
        d0, d1 = plan.args[i].d, plan.args[i + cnt - 1].d
        px = d0.N * d0.H * d0.W
        # algorithmic bytes of the LAUNCH: its first conv's input + its last conv's output, each once
        nbytes = px * d0.Cin * esz[d0.in_dtype] + d1.N * d1.H * d1.W * d1.Cout * esz[d1.out_dtype]
        flops = sum(2.0 * plan.args[j].d.N * plan.args[j].d.H * plan.args[j].d.W * plan.args[j].d.Cin * plan.args[j].d.Cout *
                    plan.args[j].d.kh * plan.args[j].d.kw for j in range(i, i + cnt))
        chans = "->".join([str(d0.Cin)] + [str(plan.args[j].d.Cout) for j in range(i, i + cnt)])
        out[f"{i}:{label}:{chans}k{d0.kh}"] = {
            "us": round(tot[u] * 1e6, 1), "algorithmic_bytes": nbytes, "GB/s": round(nbytes / tot[u] / 1e9, 1),
            "frac_of_8TB/s": round(nbytes / tot[u] / HBM_PEAK, 4), "tflops": round(flops / tot[u] / 1e12, 1)}
    return out


def cfg2(steps=10, warmup=3, compute="bf16", per_layer=True):
    """BASELINE config 2: SRCNN on 32x1x384x384, both parameterisations (the reference's 9-3-5 128/64, SRCNN/srcnn.py:205-210,
    and BASELINE's 9-1-5 64/32)."""
    from srmi.keras_models import SRCNN
    x2 = torch.rand(32, 1, 384, 384, generator=torch.Generator().manual_seed(0))
    px2 = 32 * 384 * 384
    return [run("cfg2 SRCNN 9-3-5 128/64 (reference), 32x1x384x384", SRCNN(), x2, px2, px2 * 171_392.0, steps, warmup,
                compute, per_layer=per_layer),
            run("cfg2 SRCNN 9-1-5 64/32 (BASELINE), 32x1x384x384", SRCNN(k=(9, 1, 5), ch=(64, 32)), x2, px2, px2 * 16_064.0,
                steps, warmup, compute, per_layer=per_layer)]


def cfg5_tiling(dev="cuda:0", H=540, W=960, halo=8, per_strip=False):
    """How a 960x540 frame is cut: horizontal strips that each fit the GPU as ONE chained launch of the generator's front
    (srmi.tiling.strips_for_chain: 5 strips of 121 x 960 on 256 CUs), all strips of a frame as one batch: the library runs
    the front strip by strip and the fused tail once for the batch (per_strip: one call per strip, the form before the
    chain took sub-batches); else 2 x 4 tiles as one batch."""
    from srmi import engine
    from srmi.tiling import strips_for_chain
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    ny = strips_for_chain(H, W, halo, cus) if engine._FUSE_CHAIN else None
    if ny is not None and ny <= 8:
        if per_strip:
            return dict(grid=(ny, 1), max_batch=1), f"{ny} strips of equal extended shape (one chained launch of the front each)"
        return dict(grid=(ny, 1)), (f"{ny} strips of equal extended shape as one batch (one chained launch of the front per strip, "
                                    "one fused-tail launch for all)")
    return dict(grid=(2, 4)), "2 x 4 tiles of equal extended shape 278x252, one batch of eight"


def cfg5(steps=10, warmup=3, compute="bf16", gen=None, per_strip=False):
    """BASELINE config 5 on ONE GPU: one 960x540 frame -> 3840x2160 with the 8 px halo (tiling: cfg5_tiling)."""
    from srmi.srgan_model import Generator
    from srmi.tiling import sr_tiled
    from bench import synthetic_state
    if gen is None:
        gen = Generator()
        gen.load_state_dict(synthetic_state(gen, 0))
        gen = gen.to("cuda:0").eval().set_compute(compute)
    frame = (torch.rand(1, 3, 540, 960, generator=torch.Generator().manual_seed(0)) * 2 - 1).to("cuda:0")
    kw, how = cfg5_tiling(per_strip=per_strip)
    with torch.no_grad():
        for _ in range(warmup):
            sr_tiled(gen, frame, halo=8, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out, ntiles = sr_tiled(gen, frame, halo=8, **kw)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
    flop = 8.2944e6 * 249_624          # useful output pixels only (halo recompute not counted)
    return {"workload": f"cfg5 SRResNet x4, 960x540 -> 3840x2160, 8 px halo, {how}, 1 GPU",
            "compute": compute, "ms_per_frame": round(el * 1e3, 3), "out_mp_per_s": round(8.2944 / el, 1),
            "tflops": round(flop / el / 1e12, 1), "frac_of_bf16_peak": round(flop / el / 2.5e15, 4), "tiles": ntiles}


def cfg4(steps=10, warmup=3, compute="bf16", blocks=23, per_layer=False):
    """BASELINE config 4: ESRGAN-style RRDB generator (ESRGAN/untitled0.py:18-67 topology), 8x3x256x256 -> x4, bf16."""
    from srmi.keras_models import RRDBNet
    x4 = torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(0)) * 2 - 1
    lr_px = 8 * 256 * 256
    rec = run(f"cfg4 RRDBNet {blocks} blocks, 8x3x256x256 -> x4", RRDBNet(num_blocks=blocks), x4, lr_px * 16,
              lr_px * 2.0 * (175_104 * blocks + 803_520), steps, warmup, compute, per_layer=per_layer)
    rec["frac_of_bf16_peak"] = round(rec["tflops"] / 2500.0, 4)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--compute", default="bf16")
    ap.add_argument("--blocks", type=int, default=23)
    ap.add_argument("--layers", action="store_true", help="per-layer table for cfg4 too (long)")
    ap.add_argument("--only", default="", help="comma list of cfg2,cfg5,cfg4 (default: all)")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    if not only or "cfg2" in only:
        for rec in cfg2(args.steps, args.warmup, args.compute):
            print(json.dumps(rec), flush=True)
    if not only or "cfg5" in only:
        print(json.dumps(cfg5(args.steps, args.warmup, args.compute)), flush=True)
    if not only or "cfg4" in only:
        print(json.dumps(cfg4(args.steps, args.warmup, args.compute, args.blocks, args.layers)), flush=True)


if __name__ == "__main__":
    main()
