"""Data-parallel training exchange: one gradient all-reduce per optimizer step.

The reference trains on one device (P/mode.py:141-179).  Sharded over ranks - one process per GPU,
16 samples each, BatchNorm statistics local to the rank as in DDP - the only exchange the SRGAN step
needs is the average of each model's gradients right before its optimizer step:

    d_loss.backward(); average_gradients(discriminator.parameters()); d_optim.step()     # :160-161
    g_loss.backward(); average_gradients(generator.parameters());     g_optim.step()     # :178-179

Gradients are packed into a few large flat buckets (xGMI rings are per-link bound: fewer, larger
collectives) and reduced with `torch.distributed.all_reduce` - RCCL on the "nccl" backend, gloo on CPU.
The frozen VGG has no gradients and takes no part.

Two forms:
  * `average_gradients(params)` after `backward()` - the two-line change to the reference loop shown above;
  * `GradientAverager(params)` - persistent flat buckets filled by post-accumulate-grad hooks WHILE the model's own
    backward runs (`with avg.sync(): loss.backward()`; the hooks ignore every other backward):
    a bucket's all-reduce starts as soon as its last gradient has been produced (the discriminator's 75 MB Linear
    gradient reduces while the conv layers' backward is still computing), `.grad` becomes a view of the bucket (no
    copy back), and `finish()` before `optim.step()` only waits and scales.  One pass over the gradients instead of
    three, overlapped with backward.
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


def _all_reduce(flat, group, async_op):
    """SUM all-reduce of a flat bucket.  gloo cannot be relied on for device tensors in every build: a device bucket
    under gloo (debug / single-GPU rehearsal runs) is staged through the host."""
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


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
        h = _all_reduce(flat, group, True)
        work.append((b, flat, h))
    inv = 1.0 / world
    for b, flat, h in work:
        if h is not None:
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
    host_staged = dist.get_backend(group) == "gloo"
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            # .detach() shares the version counter with t (t.data would not): cached inference plans see the write
            if host_staged and t.is_cuda:            # gloo + device tensors: through the host (see _all_reduce)
                h = t.detach().cpu()
                dist.broadcast(h, src=src, group=group)
                t.detach().copy_(h)
            else:
                dist.broadcast(t.detach(), src=src, group=group)
    from .engine import invalidate
    invalidate(module)      # belt and braces: a collective backend that writes below torch's version tracking


class GradientAverager:
    """Overlapped form of `average_gradients` for one model (see the module docstring).

        d_avg = GradientAverager(discriminator.parameters())
        g_avg = GradientAverager(generator.parameters())
        ...
        with d_avg.sync():         # P/mode.py:160-161
            d_loss.backward()      # hooks fill the buckets and launch their all-reduces as backward proceeds
        d_optim.step()             # leaving the block waited and scaled by 1/world; p.grad is a view of its bucket
        ...
        with g_avg.sync():         # P/mode.py:178-179
            g_loss.backward()
        g_optim.step()

    The hooks only act while the averager is ARMED (`begin()` ... `finish()`, or the `sync()` block).  That matters in
    the reference's SRGAN iteration: `d_loss.backward()` also reaches the generator's parameters (the generator output
    is not detached, P/mode.py:149-160) and `g_loss.backward()` reaches the discriminator's (:165,178); those foreign
    gradients are thrown away by the next `zero_grad()` and must neither be copied into a bucket nor start an
    all-reduce.  A disarmed hook does nothing at all.

    Per-parameter ready flags (not a counter): a bucket's all-reduce starts when every one of its parameters has
    reported in this armed backward.  `finish()` waits on every outstanding handle before a bucket can be reused.
    Works with the reference loop's `zero_grad()` (grads set to None): each backward hands the hook a fresh gradient,
    which is copied into the bucket once.  With world size 1 (or no process group) it does nothing."""

    def __init__(self, params, group=None, max_buckets=4):
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.armed = False
        self.params = []
        seen = set()
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        self.buckets = []          # [flat tensor, [(param, offset, numel)], set of ready param ids, work handle, launched]
        self.where = {}
        self.handles = []
        if not self.active or not self.params:
            return
        self.inv = 1.0 / dist.get_world_size(group)
        total = sum(p.numel() for p in self.params)
        target = max(1, -(-total // max(1, max_buckets)))
        cur, n = [], 0
        groups = []
        for p in reversed(self.params):               # backward produces the last layers' gradients first
            cur.append(p)
            n += p.numel()
            if n >= target:
                groups.append(cur)
                cur, n = [], 0
        if cur:
            groups.append(cur)
        for gi, ps in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
            slots, off = [], 0
            for p in ps:
                slots.append((p, off, p.numel()))
                self.where[id(p)] = (gi, off)
                off += p.numel()
            self.buckets.append([flat, slots, set(), None, False])
        for p in self.params:
            self.handles.append(p.register_post_accumulate_grad_hook(self._hook))

    # ---- arming -------------------------------------------------------------------------------------------------
    def begin(self):
        """Arm the hooks for the next backward of THIS model's loss."""
        if self.armed:
            raise RuntimeError("GradientAverager.begin(): already armed (finish() the previous backward first)")
        self.armed = True
        return self

    def sync(self):
        """Context manager: `with avg.sync(): loss.backward()` = begin(), backward, finish()."""
        return _Sync(self)

    def _hook(self, p):
        if not self.armed:
            return                              # a foreign backward (or a disarmed averager): leave p.grad alone
        gi, off = self.where[id(p)]
        b = self.buckets[gi]
        if b[4]:
            raise RuntimeError("GradientAverager: a gradient arrived after its bucket's all-reduce was launched "
                               "(two backward passes inside one begin()/finish())")
        view = b[0][off:off + p.numel()].view_as(p)
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view
        b[2].add(id(p))
        if len(b[2]) == len(b[1]):
            b[3] = _all_reduce(b[0], self.group, True)
            b[4] = True

    def finish(self):
        """Wait for every bucket, scale to the mean, disarm.  Buckets whose parameters received no gradient this
        backward (unused parameters) are skipped on every rank alike.  Returns the number of buckets reduced."""
        if not self.active:
            self.armed = False
            return 0
        done, partial = 0, False
        for b in self.buckets:
            if b[4]:
                if b[3] is not None:
                    b[3].wait()                 # never reuse a bucket with an all-reduce in flight
                b[0].mul_(self.inv)
                done += 1
            elif b[2]:
                partial = True
            b[2], b[3], b[4] = set(), None, False
        self.armed = False
        if partial:
            raise RuntimeError("GradientAverager: a bucket received gradients for only some of its parameters; "
                               "use average_gradients() for models with conditionally unused parameters")
        return done

    def close(self):
        for h in self.handles:
            h.remove()
        self.handles = []


class _Sync:
    def __init__(self, avg):
        self.avg = avg

    def __enter__(self):
        return self.avg.begin()

    def __exit__(self, et, ev, tb):
        if et is None:
            self.avg.finish()
        else:                       # backward raised: drop the half-filled state, do not mask the error
            for b in self.avg.buckets:
                if b[3] is not None:
                    b[3].wait()
                b[2], b[3], b[4] = set(), None, False
            self.avg.armed = False
        return False
