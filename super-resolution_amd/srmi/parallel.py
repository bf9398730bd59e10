"""Data-parallel training exchange: one gradient all-reduce per optimizer step.

The reference trains on one device (P/mode.py:141-179).  Sharded over ranks - one process per GPU,
16 samples each, BatchNorm statistics local to the rank as in DDP - the only exchange the SRGAN step
needs is the average of each model's gradients right before its optimizer step:

    d_loss.backward(); average_gradients(discriminator.parameters()); d_optim.step()     # :160-161
    g_loss.backward(); average_gradients(generator.parameters());     g_optim.step()     # :178-179

Gradients are packed into a few large flat buckets (xGMI rings are per-link bound: fewer, larger
collectives) and reduced with `torch.distributed.all_reduce` - RCCL on the "nccl" backend, gloo on CPU.
Buckets are launched asynchronously in reverse parameter order so the first ones overlap the packing of
the rest.  The frozen VGG has no gradients and takes no part.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _buckets(params, max_buckets):
    ps = [p for p in params if p.grad is not None]
    # de-duplicate shared parameters (the Generator's single PReLU appears once in .parameters(), but be safe)
    seen, uniq = set(), []
    for p in ps:
        if id(p) not in seen:
            seen.add(id(p))
            uniq.append(p)
    if not uniq:
        return []
    total = sum(p.grad.numel() for p in uniq)
    target = max(1, -(-total // max(1, max_buckets)))
    out, cur, n = [], [], 0
    for p in reversed(uniq):                      # backward produces the last layers' gradients first
        cur.append(p)
        n += p.grad.numel()
        if n >= target:
            out.append(cur)
            cur, n = [], 0
    if cur:
        out.append(cur)
    return out


def average_gradients(params, group=None, max_buckets=4):
    """In-place mean of `.grad` over the ranks of `group`.  No-op when torch.distributed is not initialised or
    the world has one rank.  Returns the number of buckets reduced."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    buckets = _buckets(list(params), max_buckets)
    work = []
    for b in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in b])
        h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        work.append((b, flat, h))
    inv = 1.0 / world
    for b, flat, h in work:
        h.wait()
        off = 0
        for p in b:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad).mul_(inv))
            off += n
    return len(work)


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (call once before training)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            # .detach() shares the version counter with t (t.data would not): cached inference plans see the write
            dist.broadcast(t.detach(), src=src, group=group)
    from .engine import invalidate
    invalidate(module)      # belt and braces: a collective backend that writes below torch's version tracking
