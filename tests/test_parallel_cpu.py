"""CPU, world_size 2 over gloo: the one exchange step of data-parallel training - bucketed
gradient averaging equals the mean of the per-rank gradients, and two ranks that each see half of a batch
and average their gradients take the same Adam step as one process that averages the two half-batch
gradients itself (the N-rank oracle of SURVEY 8e2)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from srmi import parallel


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.PReLU(), nn.Conv2d(8, 3, 3, padding=1))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(4, 3, 6, 6, generator=g), torch.randn(4, 3, 6, 6, generator=g)


def _worker(rank, world, port, q, overlapped=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _model()
        if rank == 1:                                # start from different parameters: broadcast must fix it
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(1.0)
        parallel.broadcast_parameters(m, src=0)
        x, y = _data()
        xs, ys = x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        if overlapped:
            # local gradients first (no exchange), then the same backward with the hooks in place
            nn.functional.mse_loss(m(xs), ys).backward()
            local = [p.grad.clone() for p in m.parameters()]
            avg = parallel.GradientAverager(m.parameters(), max_buckets=2)
            opt.zero_grad()                               # grads -> None, as the reference loop leaves them
            avg.begin()
            nn.functional.mse_loss(m(xs), ys).backward()
            nb = avg.finish()
            assert all(p.grad.data_ptr() == avg.buckets[avg.where[id(p)][0]][0].data_ptr() + 4 * avg.where[id(p)][1]
                       for p in m.parameters())           # .grad is a view of its bucket: no copy back
            avg.close()
        else:
            loss = nn.functional.mse_loss(m(xs), ys)
            opt.zero_grad()
            loss.backward()
            local = [p.grad.clone() for p in m.parameters()]
            nb = parallel.average_gradients(m.parameters(), max_buckets=2)
        opt.step()
        q.put((rank, nb, [t.numpy() for t in local], [p.grad.numpy().copy() for p in m.parameters()],
               [p.detach().numpy().copy() for p in m.parameters()]))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("overlapped", [False, True])
def test_two_rank_gradient_average_matches_single_process_oracle(overlapped):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + int(overlapped)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlapped)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] >= 1
    # averaged gradient == mean of the local gradients, identical on both ranks
    for l0, l1, a0, a1 in zip(res[0][2], res[1][2], res[0][3], res[1][3]):
        assert abs((l0 + l1) / 2 - a0).max() <= 1e-7
        assert (a0 == a1).all()
    # single-process oracle: two micro-batches, separate forwards, gradients averaged, one Adam step
    m = _model()
    x, y = _data()
    grads = []
    for r in range(2):
        m.zero_grad()
        nn.functional.mse_loss(m(x[r * 2:(r + 1) * 2]), y[r * 2:(r + 1) * 2]).backward()
        grads.append([p.grad.clone() for p in m.parameters()])
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    for p, g0, g1 in zip(m.parameters(), *grads):
        p.grad = (g0 + g1) / 2
    opt.step()
    for p, q0, q1 in zip(m.parameters(), res[0][4], res[1][4]):
        assert abs(p.detach().numpy() - q0).max() <= 1e-6
        assert (q0 == q1).all()


def test_average_is_noop_without_process_group():
    m = _model()
    nn.functional.mse_loss(m(_data()[0]), _data()[1]).backward()
    before = [p.grad.clone() for p in m.parameters()]
    assert parallel.average_gradients(m.parameters()) == 0
    assert all(torch.equal(a, p.grad) for a, p in zip(before, m.parameters()))


# ---- the full SRGAN iteration (P/mode.py:148-179) with one averager per model ---------------------------------------
# d_loss.backward() reaches G's parameters (the generator output is not detached) and g_loss.backward() reaches D's:
# every hook fires in BOTH backward passes of an iteration; only the armed averager may act.

def _gan():
    torch.manual_seed(5)
    G = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.PReLU(), nn.Conv2d(8, 3, 3, padding=1), nn.Tanh())
    D = nn.Sequential(nn.Conv2d(3, 8, 3, stride=2, padding=1), nn.BatchNorm2d(8), nn.LeakyReLU(), nn.Flatten(),
                      nn.Linear(8 * 3 * 3, 1), nn.Sigmoid())
    return G, D


def _gan_data():
    g = torch.Generator().manual_seed(6)
    return torch.rand(4, 3, 6, 6, generator=g) * 2 - 1, torch.rand(4, 3, 6, 6, generator=g) * 2 - 1


def _gan_losses(G, D, lr, gt):
    bce, mse = nn.BCELoss(), nn.MSELoss()
    ones, zeros = torch.ones(lr.shape[0], 1), torch.zeros(lr.shape[0], 1)

    def d_loss():
        out = G(lr)                                         # NOT detached, as in the reference
        return bce(D(gt), ones) + bce(D(out), zeros)

    def g_loss():
        out = G(lr)
        return 1e-3 * bce(D(out), ones) + mse(out, gt)
    return d_loss, g_loss


def _gan_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G, D = _gan()
        parallel.broadcast_parameters(G); parallel.broadcast_parameters(D)
        lr, gt = _gan_data()
        d_loss, g_loss = _gan_losses(G, D, lr[rank * 2:(rank + 1) * 2], gt[rank * 2:(rank + 1) * 2])
        g_opt, d_opt = torch.optim.Adam(G.parameters(), lr=1e-3), torch.optim.Adam(D.parameters(), lr=1e-3)
        g_avg = parallel.GradientAverager(G.parameters(), max_buckets=2)
        d_avg = parallel.GradientAverager(D.parameters(), max_buckets=2)
        reduced = []
        for _ in range(2):
            loss = d_loss()
            g_opt.zero_grad(); d_opt.zero_grad()
            with d_avg.sync():
                loss.backward()
            reduced.append(sum(1 for b in g_avg.buckets if b[4] or b[2]))      # G's hooks fired disarmed: nothing recorded
            d_opt.step()
            loss = g_loss()
            g_opt.zero_grad(); d_opt.zero_grad()
            with g_avg.sync():
                loss.backward()
            g_opt.step()
        q.put((rank, reduced, [p.detach().numpy().copy() for p in list(G.parameters()) + list(D.parameters())]))
    finally:
        dist.destroy_process_group()


def test_two_rank_srgan_iteration_with_both_averagers():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 0] and res[1][1] == [0, 0]
    # single-process oracle: per step two micro-batches with separate forwards, gradients averaged, one Adam step
    G, D = _gan()
    lr, gt = _gan_data()
    fns = [_gan_losses(G, D, lr[r * 2:(r + 1) * 2], gt[r * 2:(r + 1) * 2]) for r in range(2)]
    g_opt, d_opt = torch.optim.Adam(G.parameters(), lr=1e-3), torch.optim.Adam(D.parameters(), lr=1e-3)

    def averaged(model, which):
        acc = None
        for r in range(2):
            G.zero_grad(); D.zero_grad()
            fns[r][which]().backward()
            gr = [p.grad.clone() for p in model.parameters()]
            acc = gr if acc is None else [a + b for a, b in zip(acc, gr)]
        G.zero_grad(); D.zero_grad()
        for p, a in zip(model.parameters(), acc):
            p.grad = a / 2
    for _ in range(2):
        averaged(D, 0); d_opt.step()
        averaged(G, 1); g_opt.step()
    want = [p.detach().numpy() for p in list(G.parameters()) + list(D.parameters())]
    for w, a, b in zip(want, res[0][2], res[1][2]):
        assert (a == b).all()
        assert abs(w - a).max() <= 2e-6
