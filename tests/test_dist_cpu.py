"""CPU, world_size 2 over gloo: the N>1 path of the hot path is a static partition with NO data-path
collective.  Checks that the ranks' tile sets are disjoint and complete, that stitched tiles
reproduce the frame, and the bench's max-over-ranks timing reduction."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from srmi import tiling


def _stub_generator(x):
    """Stand-in with the Generator's signature and a zero-pixel receptive field (no HIP on CPU)."""
    return F.interpolate(x, scale_factor=4, mode="nearest"), None


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        lr = torch.rand(2, 3, 54, 96)
        out, n = tiling.sr_tiled(_stub_generator, lr, tile=(27, 24), halo=8, rank=rank, world=world)
        mine = torch.zeros(8, dtype=torch.int64)
        for i in range(8):
            if i % world == rank:
                mine[i] = 1
        # test-only reductions (the product path exchanges nothing): union of the ranks' work
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        cnt = torch.tensor([n])
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        slow = tiling.max_over_ranks(1.0 + rank, dist)
        ok = torch.equal(out, _stub_generator(lr)[0]) and int(cnt) == 8 and slow == float(world)
        q.put((rank, bool(ok), n))
    finally:
        dist.destroy_process_group()


def test_two_ranks_partition_tiles_without_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert sum(r[2] for r in res) == 8 and res[0][2] == 4


def test_plan_tiles_cover_exactly_once():
    for (H, W, th, tw, halo) in [(540, 960, 270, 240, 8), (17, 23, 5, 7, 3), (8, 8, 8, 8, 0), (9, 5, 4, 4, 40)]:
        cover = torch.zeros(H, W, dtype=torch.int32)
        for (y0, y1, x0, x1, ey0, ey1, ex0, ex1) in tiling.plan_tiles(H, W, th, tw, halo):
            cover[y0:y1, x0:x1] += 1
            assert 0 <= ey0 <= y0 < y1 <= ey1 <= H and 0 <= ex0 <= x0 < x1 <= ex1 <= W
            assert y0 - ey0 <= halo and ey1 - y1 <= halo
        assert int(cover.min()) == 1 and int(cover.max()) == 1
    assert len(tiling.plan_tiles(540, 960, 270, 240, 8)) == 8      # BASELINE config 5: one tile per GPU


def test_equal_extended_tiles_cover_exactly_once_and_share_a_shape():
    for (H, W, ny, nx, halo) in [(540, 960, 2, 4, 8), (17, 23, 3, 4, 2), (100, 100, 1, 1, 8), (64, 64, 5, 3, 4), (9, 50, 1, 6, 3)]:
        cover = torch.zeros(H, W, dtype=torch.int32)
        plan = tiling.plan_tiles_equal(H, W, ny, nx, halo)
        assert len(plan) == ny * nx
        for (y0, y1, x0, x1, ey0, ey1, ex0, ex1) in plan:
            cover[y0:y1, x0:x1] += 1
            assert 0 <= ey0 <= y0 < y1 <= ey1 <= H and 0 <= ex0 <= x0 < x1 <= ex1 <= W
        assert int(cover.min()) == 1 and int(cover.max()) == 1
        hs, ws = {t[5] - t[4] for t in plan}, {t[7] - t[6] for t in plan}
        assert max(hs) - min(hs) <= 1 and max(ws) - min(ws) <= 1
    assert {(t[5] - t[4], t[7] - t[6]) for t in tiling.plan_tiles_equal(540, 960, 2, 4, 8)} == {(278, 252)}   # cfg5: ONE batch
    lr = torch.rand(1, 3, 20, 31)
    out, n = tiling.sr_tiled(_stub_generator, lr, halo=2, grid=(2, 3))
    assert n == 6 and torch.equal(out, _stub_generator(lr)[0])
    parts = [tiling.sr_tiled(_stub_generator, lr, halo=2, grid=(2, 3), rank=r, world=2)[0] for r in range(2)]
    assert torch.equal(parts[0] + parts[1], _stub_generator(lr)[0])


def test_shard_is_a_partition():
    units = list(range(11))
    parts = [tiling.shard(units, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == units
    assert tiling.shard([], 0, 2) == []
    with pytest.raises(ValueError):
        tiling.shard(units, 4, 4)


def test_single_process_tiling_matches_whole_frame():
    lr = torch.rand(1, 3, 20, 31)
    out, n = tiling.sr_tiled(_stub_generator, lr, tile=(7, 9), halo=2)
    assert n == 12 and torch.equal(out, _stub_generator(lr)[0])
    assert tiling.max_over_ranks(0.25) == 0.25


def test_strips_for_chain_and_max_batch():
    """cfg5 on 256 CUs: five strips of 121 x 960 (11 x 20 = 220 tiles of 12 x 48 each); max_batch = 1 runs them one call
    per strip and stitches the same frame as one batch."""
    assert tiling.strips_for_chain(540, 960, 8, 256) == 5
    assert {t[5] - t[4] for t in tiling.plan_tiles_equal(540, 960, 5, 1, 8)} <= {120, 121}
    assert tiling.strips_for_chain(540, 960, 8, 304) == 4
    assert tiling.strips_for_chain(100, 100000, 8, 256) is None
    assert tiling.strips_for_chain(96, 96, 8, 256) == 1
    lr = torch.rand(1, 3, 50, 40)
    calls = []

    def fn(b):
        calls.append(b.shape[0])
        return _stub_generator(b)
    a, n = tiling.sr_tiled(fn, lr, halo=2, grid=(5, 1), max_batch=1)
    assert n == 5 and calls == [1] * 5
    b, _ = tiling.sr_tiled(_stub_generator, lr, halo=2, grid=(5, 1))
    assert torch.equal(a, b)
