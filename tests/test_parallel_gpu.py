"""GPU, world_size 2 on ONE card (both ranks use cuda:0; gloo carries the exchange, staged through the host): the
data-parallel step of SURVEY 8e2 with the real srmi modules on the HIP training kernels - each rank runs the
reference's pre-training step (P/mode.py:101-106) on its half of the batch, gradients are averaged by
`srmi.parallel` (both forms), and the result is compared with ONE process that runs the two half-batches itself
(separate forwards = per-rank BatchNorm statistics), averages the gradients and takes the same Adam step."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NB, SHAPE = 2, (4, 3, 8, 8)


def _setup():
    from oracle import srgan_oracle as O
    from oracle import weights as W
    from srmi.srgan_model import Generator
    gen = Generator(num_block=NB)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=0, num_block=NB)))
    gen = gen.to(DEV).train()
    lr = torch.from_numpy(W.lr_input(SHAPE, seed=3)).to(DEV)
    gt = torch.from_numpy(W.lr_input((SHAPE[0], 3, SHAPE[2] * 4, SHAPE[3] * 4), seed=4)).to(DEV)
    return gen, lr, gt


def _worker(rank, world, port, overlapped, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "super-resolution_amd")]
    import torch.distributed as dist
    from srmi import losses, parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        gen, lr, gt = _setup()
        if rank == 1:
            with torch.no_grad():
                gen.last_conv.body[0].bias.add_(1.0)          # broadcast must undo this
        parallel.broadcast_parameters(gen, src=0)
        h = SHAPE[0] // world
        opt = torch.optim.Adam(gen.parameters(), lr=1e-4)
        avg = parallel.GradientAverager(gen.parameters(), max_buckets=3) if overlapped else None
        out, _ = gen(lr[rank * h:(rank + 1) * h])
        loss = losses.MSELoss()(gt[rank * h:(rank + 1) * h], out)
        opt.zero_grad()
        if overlapped:
            avg.begin()
        loss.backward()
        nb = avg.finish() if overlapped else parallel.average_gradients(gen.parameters(), max_buckets=3)
        opt.step()
        torch.cuda.synchronize()
        q.put((rank, nb, {n: p.grad.cpu().numpy().copy() for n, p in gen.named_parameters()},
               {n: p.detach().cpu().numpy().copy() for n, p in gen.named_parameters()}))
    except BaseException as e:          # the parent must hear about a failure instead of waiting for a dead rank
        import traceback
        q.put((rank, "error", traceback.format_exc(), repr(e)))
        raise
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


@pytest.mark.parametrize("overlapped", [False, True])
def test_two_rank_step_with_srmi_modules(overlapped):
    from srmi import losses
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + os.getpid() % 2000 + int(overlapped)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlapped, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            r = q.get(timeout=100)
            assert r[1] != "error", r[2]
            res.append(r)
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():            # never leave a rank behind on the card
                p.kill()
                p.join(timeout=10)
    res.sort(key=lambda r: r[0])
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][1] == res[1][1] >= 1
    # one process, two micro-batches, separate forwards, averaged gradients, one Adam step (the N-rank oracle)
    gen, lr, gt = _setup()
    grads = []
    for r in range(2):
        gen.zero_grad()
        out, _ = gen(lr[r * 2:(r + 1) * 2])
        losses.MSELoss()(gt[r * 2:(r + 1) * 2], out).backward()
        grads.append({n: p.grad.clone() for n, p in gen.named_parameters()})
    opt = torch.optim.Adam(gen.parameters(), lr=1e-4)
    for n, p in gen.named_parameters():
        p.grad = (grads[0][n] + grads[1][n]) * 0.5
    opt.step()
    for n, p in gen.named_parameters():
        want = ((grads[0][n] + grads[1][n]) * 0.5).cpu().numpy()
        # deterministic kernels, same arithmetic (sum, then * 0.5): the averaged gradients are the same bits on both ranks
        assert np.array_equal(res[0][2][n], res[1][2][n]), n
        assert np.array_equal(res[0][2][n], want), n
        assert np.array_equal(res[0][3][n], p.detach().cpu().numpy()), n


# ---- the full SRGAN iteration (P/mode.py:148-179), one overlapped averager per model ------------------------------
def _gan_setup():
    from oracle import srgan_oracle as O
    from oracle import weights as W
    from srmi.srgan_model import Discriminator
    gen, lr, gt = _setup()
    disc = Discriminator(patch_size=SHAPE[2] * 4)
    disc.load_state_dict(O.to_torch(W.discriminator_weights(seed=2, patch_size=SHAPE[2] * 4)))
    return gen, disc.to(DEV).train(), lr, gt


def _gan_fns(gen, disc, lr, gt):
    from srmi import losses
    bce, mse = losses.BCELoss(), losses.MSELoss()
    ones = torch.ones((lr.shape[0], 1), device=DEV)
    zeros = torch.zeros((lr.shape[0], 1), device=DEV)

    def d_loss():
        out, _ = gen(lr)                                   # not detached (P/mode.py:149-150)
        return bce(disc(gt), ones) + bce(disc(out), zeros)

    def g_loss():
        out, _ = gen(lr)
        return 1e-3 * bce(disc(out), ones) + mse(out, gt)
    return d_loss, g_loss


def _gan_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "super-resolution_amd")]
    import torch.distributed as dist
    from srmi import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        gen, disc, lr, gt = _gan_setup()
        parallel.broadcast_parameters(gen); parallel.broadcast_parameters(disc)
        h = SHAPE[0] // world
        d_loss, g_loss = _gan_fns(gen, disc, lr[rank * h:(rank + 1) * h], gt[rank * h:(rank + 1) * h])
        g_opt = torch.optim.Adam(gen.parameters(), lr=1e-4)
        d_opt = torch.optim.Adam(disc.parameters(), lr=1e-4)
        g_avg = parallel.GradientAverager(gen.parameters(), max_buckets=3)
        d_avg = parallel.GradientAverager(disc.parameters(), max_buckets=3)
        for _ in range(2):
            loss = d_loss()
            g_opt.zero_grad(); d_opt.zero_grad()
            with d_avg.sync():
                loss.backward()
            d_opt.step()
            loss = g_loss()
            g_opt.zero_grad(); d_opt.zero_grad()
            with g_avg.sync():
                loss.backward()
            g_opt.step()
        torch.cuda.synchronize()
        q.put((rank, "ok", {"g." + n: p.detach().cpu().numpy().copy() for n, p in gen.named_parameters()} |
               {"d." + n: p.detach().cpu().numpy().copy() for n, p in disc.named_parameters()}))
    except BaseException as e:
        import traceback
        q.put((rank, "error", traceback.format_exc(), repr(e)))
        raise
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def test_two_rank_srgan_iteration_with_srmi_modules():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            r = q.get(timeout=100)
            assert r[1] != "error", r[2]
            res.append(r)
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    res.sort(key=lambda r: r[0])
    assert all(p.exitcode == 0 for p in procs)
    # one process: per optimizer step two micro-batches (separate forwards), averaged gradients, one Adam step
    gen, disc, lr, gt = _gan_setup()
    fns = [_gan_fns(gen, disc, lr[r * 2:(r + 1) * 2], gt[r * 2:(r + 1) * 2]) for r in range(2)]
    g_opt = torch.optim.Adam(gen.parameters(), lr=1e-4)
    d_opt = torch.optim.Adam(disc.parameters(), lr=1e-4)

    def averaged(model, which):
        acc = None
        for r in range(2):
            gen.zero_grad(); disc.zero_grad()
            fns[r][which]().backward()
            gr = [p.grad.clone() for p in model.parameters()]
            acc = gr if acc is None else [a + b for a, b in zip(acc, gr)]
        gen.zero_grad(); disc.zero_grad()
        for p, a in zip(model.parameters(), acc):
            p.grad = a * 0.5
    for _ in range(2):
        averaged(disc, 0); d_opt.step()
        averaged(gen, 1); g_opt.step()
    want = {"g." + n: p.detach().cpu().numpy() for n, p in gen.named_parameters()} | \
           {"d." + n: p.detach().cpu().numpy() for n, p in disc.named_parameters()}
    for n, w in want.items():
        assert np.array_equal(res[0][2][n], res[1][2][n]), n          # both ranks hold the same parameters
        assert np.array_equal(res[0][2][n], w), n                     # deterministic kernels: the oracle's bits
