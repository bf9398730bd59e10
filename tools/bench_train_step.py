#!/usr/bin/env python3
"""Secondary benchmark (BASELINE config 3 / SURVEY 8d2 cfg3): one SRGAN fine-tune iteration, P/mode.py:148-179,
per GPU LR 16x3x24x24 -> HR 16x3x96x96, Discriminator(patch_size=96), VGG19-shaped frozen stack (seeded weights),
Adam for G and D, fp32 training kernels.  With torch.distributed.run and N ranks the two gradient exchanges of
SURVEY 8e2 run over RCCL (srmi.parallel.average_gradients).

    python tools/bench_train_step.py [--steps K] [--warmup W]

Prints one JSON line: steps/s (whole job), ms per step, HR megapixels/s.  Not the headline metric (bench.py is).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd")]

import torch
import torch.optim as optim


def build_step(dev, B=16, L=24, rank=0, overlap=True, literal=False, torch_adam=False):
    """Models, optimizers and the iteration of P/mode.py:148-179 on `dev`.  Returns (step, world-independent state dict)."""
    from bench import synthetic_state
    from srmi import losses, parallel
    from srmi.srgan_model import Discriminator, Generator
    from srmi.vgg19 import vgg19

    generator = Generator()
    generator.load_state_dict(synthetic_state(generator, 0))
    generator = generator.to(dev).train()
    discriminator = Discriminator(patch_size=L * 4)
    dsd = synthetic_state(discriminator, 1)
    for k, v in discriminator.state_dict().items():          # Linear layers: keep torch's init scale
        if v.dim() == 2:
            dsd[k] = torch.randn(v.shape, generator=torch.Generator().manual_seed(2)) / v.shape[1] ** 0.5
    discriminator.load_state_dict(dsd)
    discriminator = discriminator.to(dev).train()
    vgg_net = vgg19(pre_trained=False)
    vsd = {k: (torch.randn(v.shape, generator=torch.Generator().manual_seed(3)) * (2.0 / (v.shape[1] * 9)) ** 0.5
               if v.dim() == 4 else torch.zeros(v.shape)) for k, v in vgg_net.state_dict().items()}
    vgg_net.load_state_dict(vsd)
    vgg_net = vgg_net.to(dev).eval()
    g = torch.Generator().manual_seed(10 + rank)
    lr = torch.empty(B, 3, L, L).uniform_(-1, 1, generator=g).to(dev)
    gt = torch.empty(B, 3, 4 * L, 4 * L).uniform_(-1, 1, generator=g).to(dev)

    l2_loss, cross_ent, tv_loss = losses.MSELoss(), losses.BCELoss(), losses.TVLoss()
    VGG_loss = losses.perceptual_loss(vgg_net)
    from srmi import optim as srmi_optim
    Adam = optim.Adam if torch_adam else srmi_optim.Adam          # srmi.optim.Adam: one launch per model (what mode.train uses)
    g_optim = Adam(generator.parameters(), lr=1e-4)
    d_optim = Adam(discriminator.parameters(), lr=1e-4)
    real_label = torch.ones((B, 1), device=dev)
    fake_label = torch.zeros((B, 1), device=dev)
    parallel.broadcast_parameters(generator)
    parallel.broadcast_parameters(discriminator)
    # gradient exchange of SURVEY 8e2: overlapped with backward (hooks armed around each model's OWN backward)
    g_avg = parallel.GradientAverager(generator.parameters()) if overlap else None
    d_avg = parallel.GradientAverager(discriminator.parameters()) if overlap else None

    from srmi.mode import _no_param_grads

    # The iteration as srmi.mode.train runs it (tests/test_mode_train_gpu.py: bit-equal to the reference's loop body in
    # every parameter, buffer and loss): the generator output is detached for the D step and D's parameters collect no
    # gradients in the G step - the reference computes both and throws them away (P/mode.py:149-160, :165-178; SURVEY 3.3
    # "minimal necessary work").  `literal=True` keeps the wasted work.
    from srmi import autograd as AG

    def step():
        # (srmi.mode.train: ONE generator forward per iteration - the reference's second one, P/mode.py:164, repeats the first
        #  bit for bit; BatchNorm's running statistics move twice)
        with AG.bn_running_updates(1 if literal else 2):
            output, _ = generator(lr)
        fake_prob = discriminator(output if literal else output.detach())
        real_prob = discriminator(gt)
        d_loss = cross_ent(real_prob, real_label) + cross_ent(fake_prob, fake_label)
        g_optim.zero_grad(); d_optim.zero_grad()
        if overlap:
            with d_avg.sync():
                d_loss.backward()
        else:
            d_loss.backward()
            parallel.average_gradients(discriminator.parameters())
        d_optim.step()
        if literal:
            output, _ = generator(lr)
        with (contextlib.nullcontext() if literal else _no_param_grads(discriminator)):
            fake_prob = discriminator(output)
            _percep, hr_feat, sr_feat = VGG_loss((gt + 1.0) / 2.0, (output + 1.0) / 2.0, layer="relu5_4")
            g_loss = 0.006 * _percep + 1e-3 * cross_ent(fake_prob, real_label) + \
                0.0 * tv_loss(0.006 * (hr_feat - sr_feat) ** 2) + l2_loss(output, gt)
            g_optim.zero_grad(); d_optim.zero_grad()
            if overlap:
                with g_avg.sync():
                    g_loss.backward()
            else:
                g_loss.backward()
                parallel.average_gradients(generator.parameters())
        g_optim.step()
        return d_loss, g_loss

    return step


MIN_STEP_TFLOP = 0.68      # minimal work of one iteration at B=16, LR 24 (3.4e11 MACs)


def cfg3(steps=10, warmup=3, B=16, L=24, dev=None, dist=None, rank=0, world=1, overlap=True, host_profile=False, literal=False,
         torch_adam=False):
    """Time `steps` SRGAN iterations.  Returns the record dict (rank 0 prints it in main())."""
    dev = dev or torch.device("cuda:0")
    step = build_step(dev, B, L, rank, overlap, literal, torch_adam)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    prof = None
    if host_profile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for _ in range(steps):
        d_loss, g_loss = step()
    t_host = time.perf_counter() - t0          # host time to ENQUEUE the steps (no synchronisation inside a step)
    if prof is not None:
        import pstats
        prof.disable()
        for key in ("tottime", "cumtime"):
            pstats.Stats(prof, stream=sys.stderr).sort_stats(key).print_stats(30)
    barrier()
    el = time.perf_counter() - t0
    from srmi.tiling import max_over_ranks
    el = max_over_ranks(el, dist, dev)
    ms = el / steps * 1e3
    rec = {"metric": "SRGAN fine-tune steps/sec (whole job)", "value": round(world * steps / el, 3),
           "unit": "steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(t_host / steps * 1e3, 3),
           "hr_mp_per_s": round(world * B * (4 * L) ** 2 / 1e6 / (ms * 1e-3), 3),
           "dtype": "f32", "data": "synthetic", "scaling": "weak",
           "config": {"workload": f"SRGAN step P/mode.py:148-179, per GPU LR {B}x3x{L}x{L} -> HR {4*L}, "
                                  f"D patch {4*L}, VGG19-shaped frozen stack, Adam x2",
                      "optimizer": "torch.optim.Adam (foreach)" if torch_adam else "srmi.optim.Adam (flat buffers, one launch per model)",
                      "gradient_exchange": "overlapped (GradientAverager.sync)" if overlap else "after backward",
                      "iteration": "reference-literal (wasted G backward in the D step and D weight gradients in the G step kept)"
                                   if literal else "as srmi.mode.train: one G forward per iteration (the reference's second repeats the first bit "
                                   "for bit; BN running statistics moved twice), detached for the D step, no D weight gradients "
                                   "in the G step (bit-equal results, tests/test_mode_train_gpu.py)"},
           "d_loss": round(float(d_loss.detach()), 5), "g_loss": round(float(g_loss.detach()), 5)}
    if B == 16 and L == 24:
        tf = MIN_STEP_TFLOP / (ms * 1e-3)
        rec["f32_equiv_tflops_minimal_work"] = round(tf, 1)
        rec["frac_of_f32_mfma_peak"] = round(tf / 157.3, 4)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lr-size", type=int, default=24)
    ap.add_argument("--no-overlap", action="store_true", help="exchange gradients after backward (average_gradients)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of srmi.optim.Adam")
    ap.add_argument("--literal", action="store_true", help="the reference's loop body word for word (its wasted backward work kept)")
    ap.add_argument("--host-profile", action="store_true", help="cProfile of the timed steps' enqueue path (main thread) to stderr")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    rec = cfg3(args.steps, args.warmup, args.batch, args.lr_size, dev, dist, rank, world, not args.no_overlap, args.host_profile, args.literal, args.torch_adam)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
