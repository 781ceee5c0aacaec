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
        self._host = None            # pinned staging of the tensor table
        self._dev = None
        self._partial = None
        self.last_norm = None

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        lib.t2_adam_chunks.argtypes, lib.t2_adam_chunks.restype = [C.c_long], C.c_int
        norm_out = None
        for group in self.param_groups:
            rows, chunk = [], 0
            b1, b2 = group["betas"]
            step = None
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
                step = int(st["step"].item()) if step is None else step
                rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), chunk))
                chunk += lib.t2_adam_chunks(p.numel())
            if not rows:
                continue
            n = len(rows)
            if self._host is None or self._host.numel() < n * 48:
                self._host = torch.empty(n * 48, dtype=torch.uint8).pin_memory()
                self._dev = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
            tab = (_AdamTensor * n).from_buffer(memoryview(self._host.numpy())[:n * 48])
            for i, r in enumerate(rows):
                tab[i].p, tab[i].g, tab[i].m, tab[i].v, tab[i].numel, tab[i].first_chunk = r
            self._dev[:n * 48].copy_(self._host[:n * 48], non_blocking=True)
            if self._partial is None or self._partial.numel() < chunk + 2:
                self._partial = torch.empty(chunk + 2, dtype=torch.float32, device="cuda")
            norm_out = torch.empty(2, dtype=torch.float32, device="cuda")
            lib.t2_adam_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_int, C.c_void_p]
            L.check(lib.t2_adam_step(self._dev.data_ptr(), n, chunk, self._partial.data_ptr(), norm_out.data_ptr(),
                                     float(max_norm) if max_norm else 0.0, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                     float(group["weight_decay"]), step, L.stream()))
        self.last_norm = None if norm_out is None else norm_out[0]
        return loss if closure is not None else self.last_norm
