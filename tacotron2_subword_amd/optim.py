"""FusedAdam: torch.optim.Adam (the reference's optimizer, train.py:210-211) with `step()` running as two HIP passes —
global gradient-norm clipping (torch.nn.utils.clip_grad_norm_, train.py:322-323) fused with the Adam update
(csrc/optim.hip).  Same hyper-parameters, same `state_dict` layout (step / exp_avg / exp_avg_sq per parameter), so
optimizer states interchange with torch.optim.Adam and with the reference's checkpoints.

    opt = FusedAdam(model.parameters(), lr=..., weight_decay=...)
    loss.backward()
    grad_norm = opt.step(max_norm=hparams.grad_clip_thresh)     # returns the total gradient norm (device scalar)

Differences from clip_grad_norm_ + Adam.step: gradients are not rescaled in place (the clip coefficient is applied
while they are read); parameters without a gradient are skipped, as in torch."""
import ctypes as C

import torch

from . import _lib as L


class _AdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("numel", C.c_long),
                ("first_chunk", C.c_int), ("pad_", C.c_int)]


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False)
        # Two pinned staging tables used alternately, each guarded by an event recorded behind its upload: the host may run
        # a full iteration ahead of the GPU (no sync in the training loop), and p.grad addresses change every step, so a
        # single table could be overwritten before its H2D copy had run.
        self._slots = [dict(host=None, event=None), dict(host=None, event=None)]
        self._turn = 0
        self._dev = None
        self._partial = None
        self.last_norm = None

    def _table(self, nbytes):
        """Next pinned table of at least nbytes; waits (host side) for the upload that last used it."""
        slot = self._slots[self._turn]
        self._turn ^= 1
        if slot["event"] is not None:
            slot["event"].synchronize()
        if slot["host"] is None or slot["host"].numel() < nbytes:
            slot["host"] = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        return slot

    def _fast_step(self, lib, max_norm):
        """The steady state of a training loop: one parameter group, every parameter with a gradient and a state, all at the
        same step count.  Everything that does not change between steps (parameter / moment addresses, sizes, chunk
        offsets) is cached; a step writes the gradient addresses into the pinned table as one int64 column, bumps the step
        counters with one foreach call and launches.  (The general path below costs ~0.6 ms of host time per step, which
        the GPU spends idle between the end of backward and the optimizer kernels.)  Returns None when it does not apply."""
        import numpy as np
        if len(self.param_groups) != 1:
            return None
        group = self.param_groups[0]
        live = [p.grad is not None for p in group["params"]]           # (dead parameters keep grad = None and are skipped, as in torch)
        ps = [p for p, l in zip(group["params"], live) if l]
        if not ps:
            return None
        c = getattr(self, "_fast", None)
        if c is None or c["live"] != live:
            if any(len(self.state[p]) == 0 for p in ps):
                return None
            if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
                return None
            steps = {int(self.state[p]["step"].item()) for p in ps}
            if len(steps) != 1:
                return None
            rows, chunk = np.zeros((len(ps), 6), dtype=np.int64), 0
            for i, p in enumerate(ps):
                st = self.state[p]
                rows[i, 0], rows[i, 2], rows[i, 3], rows[i, 4] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                rows[i, 5] = chunk                                        # first_chunk in the low 32 bits, pad in the high ones
                chunk += lib.t2_adam_chunks(p.numel())
            c = self._fast = dict(n=len(ps), live=live, rows=rows, chunks=chunk, step=steps.pop(), ids=[(id(p), id(self.state[p]["exp_avg"])) for p in ps],
                                  steps=[self.state[p]["step"] for p in ps])
        grads = [p.grad for p in ps]
        # a parameter or a moment replaced by load_state_dict invalidates the cache (object identity), and so does storage
        # that moved under the same object (p.data = ..., model.float()/.to()): the cached addresses are compared with
        # the live ones every step
        if any(id(p) != a or id(self.state[p]["exp_avg"]) != b for p, (a, b) in zip(ps, c["ids"])) or \
                [p.data_ptr() for p in ps] != c["rows"][:, 0].tolist() or \
                [self.state[p]["exp_avg"].data_ptr() for p in ps] != c["rows"][:, 2].tolist() or \
                [self.state[p]["exp_avg_sq"].data_ptr() for p in ps] != c["rows"][:, 3].tolist():
            self._fast = None
            return self._fast_step(lib, max_norm)                       # rebuilt from the live addresses (equal by construction: no second retry)
        if not all(g.is_contiguous() and not g.is_sparse and g.dtype == torch.float32 for g in grads):
            raise RuntimeError("FusedAdam: dense contiguous fp32 CUDA parameters only")
        n = c["n"]
        c["rows"][:, 1] = [g.data_ptr() for g in grads]
        torch._foreach_add_(c["steps"], 1)
        c["step"] += 1
        slot = self._table(n * 48)
        slot["host"][:n * 48].numpy().view(np.int64).reshape(n, 6)[:] = c["rows"]
        if self._dev is None or self._dev.numel() < n * 48:
            self._dev = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        self._dev[:n * 48].copy_(slot["host"][:n * 48], non_blocking=True)
        slot["event"] = torch.cuda.Event()
        slot["event"].record()
        if self._partial is None or self._partial.numel() < c["chunks"] + 2:
            self._partial = torch.empty(c["chunks"] + 2, dtype=torch.float32, device="cuda")
        norm_out = torch.empty(4, dtype=torch.float32, device="cuda")
        lib.t2_adam_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_int, C.c_void_p]
        b1, b2 = group["betas"]
        L.check(lib.t2_adam_step(self._dev.data_ptr(), n, c["chunks"], self._partial.data_ptr(), norm_out.data_ptr(),
                                 float(max_norm) if max_norm else 0.0, float(group["lr"]), float(b1), float(b2),
                                 float(group["eps"]), float(group["weight_decay"]), c["step"], L.stream()))
        self.last_norm = norm_out[0]
        return self.last_norm

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        lib.t2_adam_chunks.argtypes, lib.t2_adam_chunks.restype = [C.c_long], C.c_int
        # an aborted persistent kernel the host has already seen: raise before anything else (the kernels below read the same
        # status word on the device and skip the update on their own when the host has not seen it yet)
        from . import ops
        ops.check_chain_status()
        fast = self._fast_step(lib, max_norm)
        if fast is not None:
            return loss if closure is not None else fast
        # One gradient norm over ALL parameter groups (clip_grad_norm_(model.parameters()), train.py:322-323).  Rows are
        # grouped by (hyper-parameters, step count): torch.optim.Adam corrects the bias with each parameter's own step,
        # which differs between parameters once one of them starts receiving gradients later or a state is partly restored.
        batches = {}
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() and not p.grad.is_sparse):
                    raise RuntimeError("FusedAdam: dense contiguous fp32 CUDA parameters only")
                st = self.state[p]
                if len(st) == 0:                                 # torch.optim.Adam's state layout
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                batches.setdefault((gi, int(st["step"].item())), []).append(p)
        if not batches:
            self.last_norm = None
            return loss if closure is not None else None
        order = sorted(batches)
        rows, spans, chunk = [], [], 0
        for key in order:
            first, chunk0 = len(rows), chunk
            for p in batches[key]:
                st = self.state[p]
                rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), chunk))
                chunk += lib.t2_adam_chunks(p.numel())
            spans.append((first, len(rows) - first, chunk - chunk0))
        n = len(rows)
        slot = self._table(n * 48)
        tab = (_AdamTensor * n).from_buffer(memoryview(slot["host"].numpy())[:n * 48])
        for i, r in enumerate(rows):
            tab[i].p, tab[i].g, tab[i].m, tab[i].v, tab[i].numel, tab[i].first_chunk = r
        if self._dev is None or self._dev.numel() < n * 48:
            self._dev = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        self._dev[:n * 48].copy_(slot["host"][:n * 48], non_blocking=True)
        slot["event"] = torch.cuda.Event()
        slot["event"].record()
        if self._partial is None or self._partial.numel() < chunk + 2:
            self._partial = torch.empty(chunk + 2, dtype=torch.float32, device="cuda")
        norm_out = torch.empty(4, dtype=torch.float32, device="cuda")
        lib.t2_adam_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_int, C.c_void_p]
        lib.t2_adam_norm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        single = len(order) == 1
        if not single:       # norm + clip coefficient over every row first, then one update launch per (group, step)
            L.check(lib.t2_adam_norm(self._dev.data_ptr(), n, chunk, self._partial.data_ptr(), norm_out.data_ptr(),
                                     float(max_norm) if max_norm else 0.0, L.stream()))
        for (gi, step), (first, cnt, nchunks) in zip(order, spans):
            group = self.param_groups[gi]
            b1, b2 = group["betas"]
            # single batch: norm and update in one call; otherwise max_norm < 0 = "use the coefficient already in norm_out"
            L.check(lib.t2_adam_step(self._dev.data_ptr() + first * 48, cnt, nchunks, self._partial.data_ptr(), norm_out.data_ptr(),
                                     (float(max_norm) if max_norm else 0.0) if single else -1.0, float(group["lr"]), float(b1), float(b2),
                                     float(group["eps"]), float(group["weight_decay"]), step, L.stream()))
        self.last_norm = norm_out[0]
        return loss if closure is not None else self.last_norm
