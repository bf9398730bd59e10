"""Tiled x4 inference for large frames (BASELINE config 5: 960x540 -> 3840x2160, 8-px halo) and the
no-collective sharding of tiles / batch items over ranks.

The reference has no tiling code: `test_only` runs the whole image at batch 1 (P/mode.py:393-396).  Here a
frame is cut into LR core tiles, each extended by `halo` LR pixels (clipped at the image border), the
generator runs on the extended tiles (tiles of equal shape are stacked into one batch) and the core region
x scale is copied into the output.  Parity is defined per tile: a tile-with-halo through this path is the
same tensor the generator returns for that crop; with halo >= the receptive-field radius (39.75 LR px for
16 blocks) the stitched frame equals whole-frame inference, with BASELINE's 8 px it is an approximation.

Units (tiles or batch items) are dealt round-robin to ranks; no rank ever needs another rank's data.
"""
from __future__ import annotations

from collections import defaultdict

import torch


def shard(units, rank, world):
    """Static round-robin partition: unit i belongs to rank i % world."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(units)[rank::world]


def plan_tiles(H, W, tile_h, tile_w, halo):
    """[(core y0, y1, x0, x1, ext y0, y1, x0, x1)] covering H x W exactly once with core tiles."""
    if tile_h <= 0 or tile_w <= 0 or halo < 0:
        raise ValueError("tile size must be positive and halo non-negative")
    tiles = []
    for y0 in range(0, H, tile_h):
        y1 = min(y0 + tile_h, H)
        for x0 in range(0, W, tile_w):
            x1 = min(x0 + tile_w, W)
            tiles.append((y0, y1, x0, x1, max(y0 - halo, 0), min(y1 + halo, H), max(x0 - halo, 0), min(x1 + halo, W)))
    return tiles


def _equal_cuts(L, n, halo):
    """Cut [0, L) into n core intervals whose EXTENDED lengths (core + halo on every side that has a neighbour) are as
    equal as integers allow: border tiles get `halo` more core pixels than interior ones."""
    if n <= 1 or L < n:
        return [(0, L)]
    total = L + 2 * halo * (n - 1)                  # sum of the extended lengths
    base, extra = divmod(total, n)
    cuts, pos = [], 0
    for i in range(n):
        ext = base + (1 if i < extra else 0)
        core = ext - (halo if i in (0, n - 1) else 2 * halo)
        core = max(1, min(core, L - pos - (n - 1 - i)))
        end = L if i == n - 1 else pos + core
        cuts.append((pos, end))
        pos = end
    return cuts


def plan_tiles_equal(H, W, ny, nx, halo):
    """Like plan_tiles, for an ny x nx grid whose tiles all have (nearly) the SAME extended shape, so that every tile
    of a frame goes through ONE launch sequence as one batch (plan_tiles' equal core tiles give border tiles a smaller
    extended shape than interior ones: 960x540 in 2 x 4 tiles with an 8-pixel halo is two batches of four, 278x248 and
    278x256; here it is one batch of eight 278x252 tiles)."""
    if ny <= 0 or nx <= 0 or halo < 0:
        raise ValueError("grid must be positive and halo non-negative")
    tiles = []
    for (y0, y1) in _equal_cuts(H, ny, halo):
        for (x0, x1) in _equal_cuts(W, nx, halo):
            tiles.append((y0, y1, x0, x1, max(y0 - halo, 0), min(y1 + halo, H), max(x0 - halo, 0), min(x1 + halo, W)))
    return tiles


def strips_for_chain(H, W, halo, cus, th=12, tw=48):
    """Fewest horizontal strips ny such that ONE extended strip (all W columns) is at most `cus` tiles of th x tw pixels:
    the shape for which the generator's front runs as one persistent chained launch per strip (every workgroup of the
    chain must be resident at once: include/srmi.h SRMI_CONV_FUSE_CHAIN).  960x540 with an 8-pixel halo on 256 CUs: 5
    strips of 121 rows = 11 x 20 = 220 workgroups each.  None if even a th-row strip is too wide."""
    tx = -(-W // tw)
    if tx > cus:
        return None
    for ny in range(1, H + 1):
        ext = max(t[5] - t[4] for t in plan_tiles_equal(H, W, ny, 1, halo))
        if -(-ext // th) * tx <= cus:
            return ny
    return None


@torch.no_grad()
def sr_tiled(fn, lr, tile=(270, 240), halo=8, scale=4, rank=0, world=1, out=None, grid=None, max_batch=None):
    """Run `fn(lr_tiles) -> (sr, feat)` (a Generator) over this rank's tiles of `lr` (N,C,H,W).

    Returns (out, n_tiles): `out` is (N,C,scale*H,scale*W) with this rank's core regions filled (zeros
    elsewhere when world > 1: summing the ranks' outputs, or writing them to disjoint files, gives the frame)."""
    N, C, H, W = lr.shape
    # grid=(ny, nx): equal EXTENDED tile shapes (one batch per frame); otherwise equal core tiles of `tile` pixels
    plan = plan_tiles_equal(H, W, grid[0], grid[1], halo) if grid is not None else plan_tiles(H, W, tile[0], tile[1], halo)
    tiles = shard(plan, rank, world)
    if out is None:
        out = torch.zeros((N, C, scale * H, scale * W), dtype=torch.float32, device=lr.device)
    groups = defaultdict(list)
    for t in tiles:
        groups[(t[5] - t[4], t[7] - t[6])].append(t)          # equal extended shapes share one launch sequence
    for (_eh, _ew), group in groups.items():
        # max_batch: at most that many tiles per call of fn (not needed for strips_for_chain's strips: the library runs the
        # chained front in sub-batches of whole images that fit the CUs and the tail once for the batch)
        step = len(group) if not max_batch else max(1, int(max_batch))
        for g0 in range(0, len(group), step):
            ts = group[g0:g0 + step]
            batch = torch.cat([lr[:, :, t[4]:t[5], t[6]:t[7]] for t in ts], dim=0).contiguous()
            sr = fn(batch)[0]
            for i, (y0, y1, x0, x1, ey0, _ey1, ex0, _ex1) in enumerate(ts):
                oy, ox = scale * (y0 - ey0), scale * (x0 - ex0)
                out[:, :, scale * y0:scale * y1, scale * x0:scale * x1] = \
                    sr[i * N:(i + 1) * N, :, oy:oy + scale * (y1 - y0), ox:ox + scale * (x1 - x0)]
    return out, len(tiles)


def max_over_ranks(seconds, dist=None, device="cpu"):
    """Slowest rank's elapsed time (the bench contract): all-reduce MAX, identity for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
