#!/usr/bin/env python3
"""Headline benchmark: x4 SR output megapixels / second of the SRResNet generator forward
(BASELINE.json metric; workload = configs[0] "SRResNet x4 generator (16 res blocks, 64ch, 2x
PixelShuffle) forward, batch 16x3x96x96 LR, random init", the configuration north_star's target
">=60 % of peak bf16 MFMA" is quoted on).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--compute bf16|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one `Generator.forward(lr) -> (sr, feat)` through the drop-in module surface (NCHW f32 in
HBM -> NCHW f32 out), i.e. 37 convs in 20 launches (ResBlocks fused, second Upsampler + last_conv fused) + 1 re-layout of `feat`.  Inference shards by batch
item with no collective: every rank runs the same per-GPU batch (weak scaling), the
timed region is bracketed by barrier + synchronize, the slowest rank's time is used, and `value` is
the whole-job aggregate.

Extra objects on the JSON line:
  roofline     - the dominant kernel (resblock_c64_kernel: two fused 64->64 convs, 16 launches per step) against the dense
                 bf16 MFMA peak; its time is measured with HIP events on the launch stream.
  cpu_baseline - the oracle (plain torch.nn.functional restatement, pinned to the reference) timed on
                 this box's host cores, rank 0, N=1 only.
  secondary    - N=1 only: the other BASELINE configurations, each a sub-second leg measured in this same run
                 (tools/bench_variants.py, tools/bench_train_step.py): cfg2 SRCNN (both parameterisations, per-layer
                 HBM fraction), cfg3 SRGAN training step (ms/step, host-enqueue ms, f32-equivalent TFLOP/s), cfg4 RRDB-23
                 bf16 (ms, TFLOP/s, fraction of peak), cfg5 tiled 960x540 frame (ms/frame, MP/s).  --no-secondary skips it.

`--workload cfg5` (with any --gpus N) switches the timed step to BASELINE config 5: a stream of 960x540 frames, each cut into
2 x 4 LR tiles of equal extended shape 278x252 (8 px halo) that are dealt round-robin to the ranks (srmi.tiling.shard, no collective); a step =
one frame; value = useful output MP/s (halo pixels not counted); total work is fixed as N grows ("scaling": "strong").
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_MFMA_TFLOPS = 157.3  # exact-f32 MFMA
BATCH, LR_H, LR_W, NUM_BLOCK = 16, 96, 96, 16
WARMUP_MIN_S = 0.3              # untimed warm-up runs at least this long (clock ramp after idle), whatever --warmup says
FLOP_PER_OUT_PIXEL = 249_624  # SURVEY App. A / BASELINE.md section 4 (conv MACs x 2 only)
PROFILE_TAG = "r3"            # profiles/<tag>_bench_kernel_stats.csv, <tag>_pmc_traffic.json: the committed rocprofv3 runs


def conv_flops(d):
    """algorithmic FLOPs of one fused conv launch: 2 * N*Ho*Wo * Cout * kh*kw*Cin."""
    ho = (d.H + 2 * d.pad - d.kh) // d.stride + 1
    wo = (d.W + 2 * d.pad - d.kw) // d.stride + 1
    return 2.0 * d.N * ho * wo * d.Cout * d.kh * d.kw * d.Cin


def synthetic_state(gen, seed=0):
    """Random-init weights of the architecture (no checkpoints exist offline): conv W ~ N(0, 0.75/sqrt(fan_in)),
    non-trivial BatchNorm statistics, PReLU 0.25.  Values do not affect timing; they keep activations O(1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in gen.state_dict().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = v.clone()
        elif k.endswith("running_var"):
            sd[k] = torch.empty(v.shape).uniform_(0.5, 1.5, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        elif v.dim() == 4:
            sd[k] = torch.randn(v.shape, generator=g) * (0.75 / (v.shape[1] * v.shape[2] * v.shape[3]) ** 0.5)
        elif k.endswith("bias"):
            sd[k] = torch.empty(v.shape).uniform_(-0.1, 0.1, generator=g)
        elif v.numel() == 1:
            sd[k] = torch.full(v.shape, 0.25)
        else:
            sd[k] = torch.empty(v.shape).uniform_(0.5, 1.5, generator=g)
    return sd


def build_generator(device, compute):
    from srmi.srgan_model import Generator
    gen = Generator(num_block=NUM_BLOCK)
    sd = synthetic_state(gen, seed=0)
    gen.load_state_dict(sd)
    x = torch.empty((BATCH, 3, LR_H, LR_W)).uniform_(-1.0, 1.0, generator=torch.Generator().manual_seed(1))
    return gen.to(device).eval().set_compute(compute), x.to(device), sd, x


def per_kernel_times(gen, x, reps):
    """HIP-event time of every launch of one forward, averaged over `reps` instrumented forwards.
    Launches go to torch's current stream, which is the stream the events are recorded on."""
    from srmi import _lib as L
    from srmi import functional as Fn
    plan = gen._plans.get(gen, x.shape[0], x.shape[2], x.shape[3], 1 if gen.compute == "bf16" else 0)
    lib = L.load()
    n = plan.n
    plan.args[0].x = x.data_ptr()
    sr = torch.empty(plan.sr_shape, dtype=torch.float32, device=x.device)
    plan.args[n - 1].y = sr.data_ptr()
    if plan.feat_idx is not None:                # the NCHW f32 copy of `feat` some launch writes on the side (x_copy)
        feat = torch.empty((plan.N, plan.nf, plan.H, plan.W), dtype=torch.float32, device=x.device)
        plan.args[plan.feat_idx].x_copy = feat.data_ptr()
    units = plan.launches                       # (label, first conv index, convs in the launch)
    # Runs of identical consecutive launches (the 16 fused ResBlocks) share ONE event pair: recording an event costs the
    # GPU timeline a few microseconds (two events with nothing between them read 5 us apart), which a per-launch pair
    # would add to every 28 us kernel; the run's time / its length is the average launch-to-launch duration.
    def sig(u):
        label, i, cnt = units[u]
        d = plan.args[i].d
        return (label, cnt, d.N, d.H, d.W, d.Cin, d.Cout, d.kh)
    groups, u = [], 0
    while u < len(units):
        v = u + 1
        while v < len(units) and sig(v) == sig(u):
            v += 1
        groups.append((u, v))
        u = v
    tot = np.zeros(len(units))
    st = Fn.stream_ptr()
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(groups) + 1)]
        ev[0].record()
        for gi, (u0, u1) in enumerate(groups):
            for u in range(u0, u1):
                label, i, cnt = units[u]
                if cnt > 2:         # the chained trunk: one persistent launch of cnt / 2 ResBlocks
                    L.check(lib.srmi_conv2d_fwd_seq(C.byref(plan.args[i]), cnt, st))
                elif cnt == 2:
                    L.check(lib.srmi_conv2d_fwd_pair(C.byref(plan.args[i]), C.byref(plan.args[i + 1]), st))
                else:
                    L.check(lib.srmi_conv2d_fwd(C.byref(plan.args[i]), st))
            ev[gi + 1].record()
        torch.cuda.synchronize()
        for gi, (u0, u1) in enumerate(groups):
            tot[u0:u1] += ev[gi].elapsed_time(ev[gi + 1]) / (u1 - u0)
    return plan, units, tot / reps   # ms per launch


def host_cpu_info():
    """CPU model, physical and logical core counts of the host, and the CPUs this process may run on."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid)); pid = cid = None
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    return {"model": model, "physical_cores": len(phys) or None, "logical_cpus": logical or os.cpu_count(),
            "cpus_allowed": allowed}


def baseline_threads(info):
    """Threads for the CPU leg: the job's CPU share.  OMP_NUM_THREADS is what the box allots a one-GPU job (16 of a
    many-socket host); without it, the CPUs this process may run on, at most the physical core count."""
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        return int(env)
    n = info["cpus_allowed"]
    if info["physical_cores"]:
        n = min(n, info["physical_cores"])
    return max(1, n)


def hbm_bound_layers(plan, units, kms, gen, x):
    """Achieved HBM GB/s of the memory-bound launches of the step (algorithmic bytes = compulsory input + output, SURVEY
    8d3): last_conv 64->3 + tanh (128 B bf16 in + 12 B f32 out per output pixel) and the NHWC bf16 -> NCHW f32 re-layout
    of `feat` (2 + 4 B per element), the latter timed here with its own pair of HIP events."""
    from srmi import _lib as L
    from srmi import functional as Fn
    out = {}
    for u, (label, i, cnt) in enumerate(units):
        d = plan.args[i].d
        if label == "c64_thin_bf16":
            nbytes = d.N * d.H * d.W * (d.Cin * 2 + d.Cout * 4)
            out[f"{label}:{d.Cin}->{d.Cout}@{d.H}x{d.W}"] = (nbytes, kms[u] * 1e-3)
        if label == "upfuse_bf16":        # MFMA-bound, listed for its traffic: bf16 input once + the f32 image once
            d2 = plan.args[i + 1].d
            nbytes = d.N * d.H * d.W * d.Cin * 2 + d2.N * d2.H * d2.W * d2.Cout * 4
            out[f"{label}:{d.Cin}->{d.Cout}->{d2.Cout}@{d2.H}x{d2.W}(mfma-bound)"] = (nbytes, kms[u] * 1e-3)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        Fn.layout_convert(plan.feat_nhwc, plan.N, plan.nf, plan.H, plan.W, L.NHWC, L.NCHW, torch.float32)
    e1.record()
    torch.cuda.synchronize()
    out[f"feat_relayout:nhwc_bf16->nchw_f32@{plan.H}x{plan.W}x{plan.nf}"] = (
        plan.N * plan.H * plan.W * plan.nf * 6, e0.elapsed_time(e1) * 1e-3 / reps)
    return {k: {"algorithmic_bytes": b, "us": round(t * 1e6, 2), "GB/s": round(b / t / 1e9, 1),
                "frac_of_8TB/s": round(b / t / 8e12, 4)} for k, (b, t) in out.items()}


def fp32_path(dev, steps=3):
    """The f32 path (SRMI_COMPUTE_F32: the <=1e-4 parity path) on the same workload: MP/s and its fraction of the
    157 TFLOP/s f32-MFMA peak.  Convs with Cin, Cout multiples of 32 run on the bf16 matrix cores with operands split
    into three bf16 (six products per k: an f32 FMA chain's rounding, csrc/conv_b3_f32.hip), so the fraction is against a
    peak the path is no longer bound by; SRMI_F32_EXACT=1 in the environment keeps the exact-f32 MFMA kernel."""
    gen, x, _, _ = build_generator(dev, "fp32")
    with torch.no_grad():
        gen(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gen(x)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps
    mp = BATCH * (4 * LR_H) * (4 * LR_W) / 1e6
    tf = mp * 1e6 * FLOP_PER_OUT_PIXEL / t / 1e12
    return {"value": round(mp / t, 2), "unit": "MP/s", "ms_per_step": round(t * 1e3, 3), "dtype": "f32",
            "tflops": round(tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "steps": steps,
            "arith": "exact f32 MFMA" if os.environ.get("SRMI_F32_EXACT", "0") not in ("", "0") else
                     "3 x bf16 split operands, 6 MFMA products per k, f32 accumulate (exact f32 MFMA where Cin or Cout % 32 != 0)"}


def cpu_baseline(threads, sd, x, info=None):
    """The oracle (oracle/srgan_oracle.py: plain torch.nn.functional restatement, pinned bit-exact to the
    imported reference in the build container) on the host cores, same weights and inputs as the GPU leg."""
    from oracle import srgan_oracle as O
    torch.set_num_threads(threads)
    times = []
    with torch.no_grad():
        O.generator_forward(sd, x, num_block=NUM_BLOCK)          # warm-up
        for _ in range(3):
            t0 = time.perf_counter()
            O.generator_forward(sd, x, num_block=NUM_BLOCK)
            times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    mp = BATCH * (4 * LR_H) * (4 * LR_W) / 1e6
    res = {"value": round(mp / t, 4), "unit": "MP/s", "cores": threads, "kind": "port",
           "sample": f"1 warm-up + 3 timed forwards of the same 16x3x96x96 batch, median {t:.2f} s, fp32, "
                     f"torch {torch.__version__} CPU, torch.get_num_threads()={torch.get_num_threads()}"}
    if info:
        res["cpu"] = info
    return res


def secondary(dev):
    """The other BASELINE configurations on this GPU, each leg a fraction of a second of device time.
    A leg that fails reports its error instead of taking the headline line down with it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_train_step as T
    import bench_variants as V
    out = {}

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:     # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    leg("cfg2_srcnn", lambda: V.cfg2(steps=20, warmup=3))
    leg("cfg3_srgan_train_step", lambda: T.cfg3(steps=10, warmup=3, dev=dev))
    leg("cfg4_rrdb23", lambda: V.cfg4(steps=10, warmup=3))
    leg("cfg5_tiled_960x540", lambda: V.cfg5(steps=20, warmup=3))
    return out


def run_cfg5(args, dev, dist, rank, world):
    """BASELINE config 5 as the timed step: frames of 960x540 -> 3840x2160, tiles sharded over the ranks."""
    from srmi.srgan_model import Generator
    from srmi.tiling import max_over_ranks, sr_tiled
    gen = Generator(num_block=NUM_BLOCK)
    gen.load_state_dict(synthetic_state(gen, seed=0))
    gen = gen.to(dev).eval().set_compute(args.compute)
    frame = torch.empty((1, 3, 540, 960)).uniform_(-1.0, 1.0, generator=torch.Generator().manual_seed(1)).to(dev)
    out = torch.zeros((1, 3, 2160, 3840), dtype=torch.float32, device=dev)
    steps = max(args.steps, 32)                      # a stream of >= 32 frames
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_variants as V
    kw, how = V.cfg5_tiling(dev)
    from srmi.tiling import plan_tiles_equal
    nt = len(plan_tiles_equal(540, 960, kw["grid"][0], kw["grid"][1], 8))

    def frame_rank(f):        # the stream's (frame, tile) units go round-robin to the ranks: tile t of frame f -> (f nt + t) % world
        return (rank - f * nt) % world

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        for f in range(args.warmup):
            sr_tiled(gen, frame, halo=8, rank=frame_rank(f), world=world, out=out, **kw)
        barrier()
        t0 = time.perf_counter()
        ntiles = 0
        for f in range(steps):
            ntiles += sr_tiled(gen, frame, halo=8, rank=frame_rank(f), world=world, out=out, **kw)[1]
        barrier()
        elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, dev)
    if rank != 0:
        return None
    mp_frame = 3840 * 2160 / 1e6
    value = mp_frame * steps / elapsed
    tf = value * 1e6 * FLOP_PER_OUT_PIXEL / 1e12
    return {"metric": "x4 SR output megapixels/sec (whole node)", "value": round(value, 2), "unit": "MP/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if args.compute == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "cfg5: tiled SRResNet x4 inference, 960x540 -> 3840x2160 frames, 8 px halo, " + how +
                                   "; the stream's (frame, tile) units dealt round-robin to the ranks, no collective; useful pixels only",
                       "tiles_per_frame": nt, "tiles_rank0": ntiles, "parallelism": f"tile-sharded x{world}, no collective"},
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_BF16_TFLOPS * world, "unit": "TFLOP/s",
                         "frac": round(tf / (PEAK_BF16_TFLOPS * world), 4), "traffic": None,
                         "note": "whole-job useful FLOPs (249,624 per output pixel) / time; the per-kernel roofline is on the cfg1 line"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg2..cfg5 legs of the N=1 line")
    ap.add_argument("--workload", default="cfg1", choices=["cfg1", "cfg5"])
    ap.add_argument("--no-chain", action="store_true",
                    help="A/B: one launch per ResBlock instead of the chained trunk (srmi.engine.set_fuse_chain(False))")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as written: start the N ranks ourselves as FRESH child processes.  Nothing in
        # this process has touched the GPU yet (no torch.cuda call above), and it never will: it only relays.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
        for ln in child.stdout.splitlines():
            if not ln.startswith("{"):
                print(ln, file=sys.stderr)
        if child.returncode != 0 or not lines:
            raise SystemExit(child.returncode or 1)
        print(lines[-1], flush=True)
        return
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks, or run without torch.distributed.run")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    if args.workload == "cfg5":
        result = run_cfg5(args, dev, dist, rank, world)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(result), flush=True)
        return

    if args.no_chain:
        from srmi import engine as _engine
        _engine.set_fuse_chain(False)
    gen, x, sd_cpu, x_cpu = build_generator(dev, args.compute)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # W untimed warm-up steps - and, still untimed, further ones until WARMUP_MIN_S of wall time have passed: the first
        # tens of milliseconds after an idle period run at ramping clocks (20 steps timed right behind 5 warm-up steps read
        # 13 % slower than the same 20 steps behind a 0.3 s warm-up; 200+ timed steps do not care).  The count is reported.
        tw0 = time.perf_counter()
        warm_run = 0
        while warm_run < args.warmup or time.perf_counter() - tw0 < WARMUP_MIN_S:
            gen(x)
            warm_run += 1
            if warm_run % 16 == 0:
                torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sr, feat = gen(x)
        barrier()
        elapsed = time.perf_counter() - t0
    from srmi.tiling import max_over_ranks
    elapsed = max_over_ranks(elapsed, dist, dev)      # slowest rank

    out_mp_per_step = world * BATCH * (4 * LR_H) * (4 * LR_W) / 1e6
    value = out_mp_per_step * args.steps / elapsed
    ms = elapsed / args.steps * 1e3

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events, same process, same data) -------------
        with torch.no_grad():
            plan, units, kms = per_kernel_times(gen, x, reps=max(5, min(20, args.steps)))
        # dominant kernel = resblock_c64_kernel (two fused 64->64 convs; 16 launches = 59 % of the step's FLOPs and
        # ~60 % of its time); the whole 64-channel MFMA family (resblock + conv_c64) is reported beside it.
        chained = any(label == "resblock_chain_bf16" for (label, i, cnt) in units)
        dom = ("resblock_chain_bf16" if chained else "resblock_bf16") if args.compute == "bf16" else "b3_f32"
        dom_kernel = "resblock_chain_kernel" if chained else "resblock_c64_kernel"
        fam = ("c64_bf16", "resblock_bf16", "resblock_chain_bf16") if args.compute == "bf16" else ("generic_f32", "f32t", "b3_f32", "b3h_f32")
        peak = PEAK_BF16_TFLOPS if args.compute == "bf16" else PEAK_F32_MFMA_TFLOPS
        per_layer, flops, t_dom, n_dom, f_fam, t_fam, n_blocks = {}, 0.0, 0.0, 0, 0.0, 0.0, 0
        for u, (label, i, cnt) in enumerate(units):
            d = plan.args[i].d
            fl = sum(conv_flops(plan.args[i + c].d) for c in range(cnt))
            key = f"{label}:{d.Cin}->{plan.args[i + cnt - 1].d.Cout}@{d.H}x{d.W}k{d.kh}"
            e = per_layer.setdefault(key, [0, 0.0, 0.0])
            e[0] += 1; e[1] += kms[u]; e[2] += fl
            if label == dom:
                flops += fl; t_dom += kms[u] * 1e-3; n_dom += 1
                n_blocks += fl / (2 * conv_flops(plan.args[i + 1].d))     # (a chain's head and tail convs count by their FLOPs)
            if label in fam:
                f_fam += fl; t_fam += kms[u] * 1e-3
        achieved = flops / t_dom / 1e12
        # compulsory HBM bytes of one launch of the dominant kernel: the tensor it reads and the tensor(s) it writes
        algo_bytes = 2 * BATCH * LR_H * LR_W * 64 * 2            # a fused ResBlock: 64-channel bf16 in and out
        for (label, i, cnt) in units:
            if label == "resblock_chain_bf16":                   # image in; `feat` (NCHW f32) and the last link's tensor out
                d0, d1 = plan.args[i].d, plan.args[i + cnt - 1].d
                algo_bytes = d0.N * d0.H * d0.W * d0.Cin * (2 if d0.in_dtype == 1 else 4) + d1.N * d1.H * d1.W * d1.Cout * 2
                if plan.feat_idx is not None and i <= plan.feat_idx < i + cnt:
                    algo_bytes += plan.N * plan.nf * plan.H * plan.W * 4
        layers = {k: {"launches": v[0], "ms": round(v[1], 4), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                  for k, v in per_layer.items()}
        trace_us = None      # average duration of the dominant kernel in the committed kernel trace (profiles/)
        try:
            import csv
            for r in csv.DictReader(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_bench_kernel_stats.csv"))):
                if dom_kernel in r["Name"] and args.compute == "bf16":
                    trace_us = round(float(r["AverageNs"]) / 1e3, 2)
                    break
        except Exception:
            trace_us = None
        traffic = None       # HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_summary.py)
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_traffic.json")))["kernels"]
            traffic = next(v["hbm_bytes"] for k, v in pm.items() if dom_kernel in k) if args.compute == "bf16" else None
        except Exception:
            traffic = None
        roofline = {"bound": "mfma", "kernel": dom_kernel if args.compute == "bf16" else "conv_generic_kernel",
                    "launches_per_step": n_dom, "resblocks_per_launch": (n_blocks / n_dom) if args.compute == "bf16" else None,
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_source": f"profiles/{PROFILE_TAG}_pmc_traffic.json: committed rocprofv3 --pmc passes of this command "
                                      "(FETCH_SIZE x2 + WRITE_SIZE per launch); NOT measured in this run (PMC needs the profiler)",
                    "avg_launch_us": round(t_dom / n_dom * 1e6, 2),
                    "avg_resblock_us": round(t_dom / n_blocks * 1e6, 2) if n_blocks else None,
                    # average kernel duration in the committed rocprofv3 kernel trace of this command (kernels of consecutive
                    # launches overlap by ~2 us there: the next one's waves start while the previous one's stores drain)
                    "kernel_trace_avg_us_from_committed_profile": trace_us,
                    "frac_by_committed_kernel_trace": round(flops / n_dom / (trace_us * 1e-6) / 1e12 / peak, 4) if trace_us else None,
                    "flops_per_launch": flops / n_dom,
                    # (for the chained trunk: what the links hand each other - two-pixel halo strips, write-through - is traffic
                    #  beyond these bytes: the PMC figure shows it)
                    "algorithmic_bytes_per_launch": algo_bytes,
                    "mfma_family_tflops": round(f_fam / t_fam / 1e12, 2),
                    "whole_step_frac": round(value / world * 1e6 * FLOP_PER_OUT_PIXEL / 1e12 / peak, 4),
                    "sum_kernel_ms": round(float(kms.sum()), 4), "layers": layers,
                    "hbm_bound_layers": hbm_bound_layers(plan, units, kms, gen, x),
                    "method": "HIP events (torch's current stream = the launch stream) around every launch (one pair around each run of "
                              "identical consecutive launches, e.g. the 16 ResBlocks: time / run length) of "
                              "instrumented forwards run right after the timed region, same process/data; achieved = "
                              "algorithmic conv FLOPs of the kernel's launches / their event time (measured live); traffic and "
                              "kernel_trace_avg_us_from_committed_profile are read from profiles/ (see traffic_source)"}
        result = {
            "metric": "x4 SR output megapixels/sec (whole node)", "value": round(value, 2), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": warm_run, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.compute == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "SRResNet x4 generator forward (16 res blocks, 64ch, 2x PixelShuffle), "
                                   "batch 16x3x96x96 LR per GPU -> 16x3x384x384, random-init recipe weights, "
                                   "inputs resident in HBM, drop-in Generator.forward incl. NCHW<->NHWC edges",
                       "per_gpu_batch": BATCH, "lr": [LR_H, LR_W], "parallelism": f"batch-sharded x{world}, no collective"},
            "roofline": roofline,
        }
        if world == 1 and args.compute == "bf16":
            del gen                                   # plan buffers of the bf16 model are no longer needed
            result["fp32_path"] = fp32_path(dev)
        if world == 1 and args.compute == "bf16" and not args.no_secondary:
            result["secondary"] = secondary(dev)
        if world == 1 and not args.no_cpu_baseline:
            info = host_cpu_info()
            result["cpu_baseline"] = cpu_baseline(baseline_threads(info), sd_cpu, x_cpu, info)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
