"""`srmi.optim.Adam`: torch.optim.Adam's update (P/mode.py:90, :132 - defaults, lr 1e-4) as ONE launch per model.

torch's default (foreach) Adam walks a model's 141 / 22 parameter tensors through a dozen multi-tensor kernels per step
and spends milliseconds of host time doing it (on a box with slow cores: 6 ms of the SRGAN iteration's 18).  Here a
parameter group's parameters, gradients and both moments live in flat f32 buffers: the parameters are re-pointed at
views of the flat buffer when the optimizer is built (after `.to(device)`, as the reference builds it), `step()` gathers
the gradients with one `torch.cat` into the flat gradient buffer and calls `srmi_adam_step` once.

Same constructor keywords as torch.optim.Adam for what the reference uses (`lr`, `betas`, `eps`); `param_groups` /
`zero_grad()` / `state_dict()` come from torch.optim.Optimizer, so `lr_scheduler.StepLR(g_optim, 2000, 0.1)` (P/mode.py:133)
works unchanged.  The update follows torch's arithmetic order with scalars formed in double (include/srmi.h); it agrees
with torch.optim.Adam to an ulp of each parameter per step, not bit for bit (tests/test_train_ops_gpu.py).
Parameters whose gradient is None in a step are skipped, as torch skips them (their moments do not move).
"""
from __future__ import annotations

import torch

from . import _lib as L
from .functional import stream_ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("srmi.optim.Adam: weight_decay / amsgrad are not implemented (the reference uses neither)")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("srmi.optim.Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in ps):
                raise RuntimeError("srmi.optim.Adam: parameters must be float32 tensors on one GPU (build the optimizer "
                                   "after model.to(device), as the reference does)")
            sizes = [p.numel() for p in ps]
            n = sum(sizes)
            flat = torch.empty(n, dtype=torch.float32, device=dev)
            offs, off = [], 0
            with torch.no_grad():
                for p, sz in zip(ps, sizes):
                    view = flat[off:off + sz].view(p.shape)
                    view.copy_(p)
                    p.data = view                    # same Parameter object, storage inside the flat buffer
                    offs.append(off)
                    off += sz
            self._flat.append(dict(params=ps, offs=offs, sizes=sizes, n=n, p=flat, g=torch.empty_like(flat),
                                   m=torch.zeros_like(flat), v=torch.zeros_like(flat), step=0))
        # (cached inference plans / packed weights are keyed by each parameter's version and data_ptr: the new storage
        #  is seen at their next use)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            grads = [p.grad for p in f["params"]]
            if all(g is None for g in grads):
                continue
            f["step"] += 1
            b1, b2 = group["betas"]
            st = stream_ptr(f["p"].device)
            # runs of consecutive parameters that have a gradient (normally ONE run = the whole model)
            runs, i, k = [], 0, len(grads)
            while i < k:
                if grads[i] is None:
                    i += 1
                    continue
                j = i
                while j < k and grads[j] is not None:
                    j += 1
                runs.append((i, j))
                i = j
            for (i, j) in runs:
                lo = f["offs"][i]
                hi = f["offs"][j - 1] + f["sizes"][j - 1]
                torch.cat([g.reshape(-1) for g in grads[i:j]], out=f["g"][lo:hi])
                L.check(lib.srmi_adam_step(f["p"][lo:hi].data_ptr(), f["g"][lo:hi].data_ptr(), f["m"][lo:hi].data_ptr(),
                                           f["v"][lo:hi].data_ptr(), hi - lo, float(group["lr"]), float(b1), float(b2),
                                           float(group["eps"]), f["step"], st))
                # the kernel wrote the parameters below torch's version tracking: tell the packed-weight / plan caches
                torch.autograd.graph.increment_version(f["params"][i:j])
        return loss

    def moments(self, group=0):
        """(step, exp_avg, exp_avg_sq) of a parameter group as flat tensors (checkpointing / tests)."""
        f = self._flat[group]
        return f["step"], f["m"], f["v"]
