"""Host-side wrappers over the C ABI: allocate outputs/workspaces as torch tensors, pass raw
device pointers, enqueue on torch's current stream.  No arithmetic happens here."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def _i32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.to(dtype=torch.int32).contiguous()


def gemm(A: torch.Tensor, B: torch.Tensor, *, trans_a: bool = False, trans_b: bool = True, bias=None, act: int = 0,
         alpha: float = 1.0, beta: float = 0.0, out: Optional[torch.Tensor] = None, splitk: int = 0,
         ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = act(alpha * op(A) @ op(B) + bias) + beta*C with op(A) = A[M,K] (or A[K,M]^T) and
    op(B) = B[N,K]^T (trans_b, nn.Linear weight layout) or B[K,N]."""
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    N = B.shape[0] if trans_b else B.shape[1]
    sam, sak = (1, A.stride(0)) if trans_a else (A.stride(0), 1)
    sbn, sbk = (B.stride(0), 1) if trans_b else (1, B.stride(0))
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    L.check(L.lib().t2_gemm(L.ptr(A), L.ptr(B), L.ptr(out), M, N, K, sam, sak, sbn, sbk, out.stride(0), L.ptr(bias), act,
                            alpha, beta, L.ptr(ws), 0 if ws is None else ws.numel() * 4, splitk, L.stream()))
    return out


def rng_keep_mask(seed: int, site: int, n: int, p: float, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.uint8, device=device)
    L.check(L.lib().t2_rng_keep_mask(seed, site, n, p, L.ptr(out), L.stream()))
    return out


def rng_normal(seed: int, site: int, n: int, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.float32, device=device)
    L.check(L.lib().t2_rng_normal(seed, site, n, L.ptr(out), L.stream()))
    return out


# Callables run (on the launching thread, with the launch stream current) right before a teacher-forced pass that may launch
# persistent chain kernels.  A persistent grid needs every CU: a collective still running on another stream would keep
# part of the grid from becoming resident until it ends, so distributed.apply_gradient_allreduce registers a callable here
# that makes the launch stream wait for the reductions in flight (a stream-level wait, the host does not block).
PRE_PERSISTENT = []

# Sticky status of the persistent kernels (include/t2amd.h t2_chain_status): a chain whose hand-off timed out writes a code
# into a word in page-locked host memory.  Reading it is a plain load — no copy, no event, no synchronisation — so every
# entry point below looks at it, and so does whoever has just synchronised for a reason of their own (Decoder.inference
# after its stop indices arrived, a training loop after loss.item()).  On the device the optimizer reads the same word and
# changes no parameter while it is set, and the passes overwrite an aborted chain's outputs with NaN; the host-side raise
# may therefore come one or two (asynchronous) steps late, but nothing is ever trained on or returned as data in between.
_status_buf = (C.c_uint32 * 4)()


def chain_status_words():
    """(code, 0, 0, 0) of the current device's sticky status block; code 0 = every persistent kernel so far ran through."""
    L.check(L.lib().t2_chain_status(_status_buf))
    return tuple(int(v) for v in _status_buf)


def check_chain_status(block: bool = False) -> None:
    """Raises if a persistent chain kernel aborted since the last check (and clears the status, so that a caller who
    catches the error can carry on with the per-step path: _lib.set_chain(False)).  block: synchronise the device first,
    i.e. cover everything enqueued so far."""
    if not torch.cuda.is_available():
        return
    if block:
        torch.cuda.synchronize()
    st = chain_status_words()
    if any(st):
        L.check(L.lib().t2_chain_status_clear())
        raise RuntimeError(f"t2amd: a persistent chain kernel aborted (status {st}: a hand-off between workgroups timed out — is "
                           "another process or another persistent kernel using this GPU?).  The outputs of that pass were overwritten "
                           "with NaN and no optimizer step has been applied since")


def debug_report_abort(code: int = 99) -> None:
    """Tests: enqueue a kernel that reports an abort exactly as a timed-out chain would."""
    L.check(L.lib().t2_debug_report_abort(code, L.stream()))


class DecoderPass:
    """Outputs + saved-activation workspace of one teacher-forced decoder pass."""

    def __init__(self, dims, B, T, Tin, Tsub, device):
        self.dims, self.B, self.T, self.Tin, self.Tsub = dims, B, T, Tin, Tsub
        self.layout = L.decoder_layout(dims, B, T, Tin, Tsub)
        self.ws = torch.empty(self.layout.total_floats, dtype=torch.float32, device=device)
        self.mel = torch.empty(B, T, dims.n_mel, dtype=torch.float32, device=device)
        self.gate = torch.empty(B, T, dtype=torch.float32, device=device)
        self.align = torch.empty(B, T, Tin, dtype=torch.float32, device=device)
        self.align_sub = torch.empty(B, T, Tsub, dtype=torch.float32, device=device)

    def chain_status(self):
        """Status words of the persistent kernels — (forward attention chain, forward decoder-LSTM chain, backward
        decoder-LSTM chain, backward attention chain): 0 = completed (or not used).
        Synchronises the device: for tests and the end of a benchmark, not for the training loop."""
        w = self.ws[self.layout.chain:self.layout.chain + 4].view(torch.int32).cpu()
        return tuple(int(v) for v in w)

    def view(self, name: str, *shape) -> torch.Tensor:
        off = getattr(self.layout, name)
        n = 1
        for s in shape:
            n *= s
        return self.ws[off:off + n].view(*shape)


def decoder_forward(W: L.DecoderWeights, dims: L.Dims, memory, memory_sub, mem_lengths, sub_lengths, mels, *,
                    training: bool, prenet_dropout: bool, seed: int, keep=None, dp: Optional[DecoderPass] = None) -> DecoderPass:
    """Teacher-forced decoder (Decoder.forward, model.py:392-428).  mels: [B,n_mel,T].  dp: a pass whose
    memory-independent part decoder_prologue has already run (the caller has ordered the streams)."""
    check_chain_status()
    B, Tin, _ = memory.shape
    Tsub, T = (1 if memory_sub is None else memory_sub.shape[1]), mels.shape[2]     # single-stream: no second memory
    phase = 0 if dp is None else 2
    if dp is None:
        dp = DecoderPass(dims, B, T, Tin, Tsub, memory.device)
    assert (dp.B, dp.T, dp.Tin, dp.Tsub) == (B, T, Tin, Tsub)
    ml, sl = _i32(mem_lengths), _i32(sub_lengths)
    a = L.DecoderFwdArgs(B, T, Tin, Tsub, L.ptr(memory), L.ptr(memory_sub), L.ptr(ml), L.ptr(sl), L.ptr(mels),
                         L.ptr(dp.mel), L.ptr(dp.gate), L.ptr(dp.align), L.ptr(dp.align_sub), L.ptr(dp.ws),
                         int(training), int(prenet_dropout), seed, phase)
    L.check(L.lib().t2_decoder_forward(C.byref(dims), C.byref(W), C.byref(a), L.stream()))
    dp._keep = (ml, sl, keep)
    return dp


def decoder_prologue(W: L.DecoderWeights, dims: L.Dims, dp: DecoderPass, mels, *, training: bool, prenet_dropout: bool,
                     seed: int) -> None:
    """The part of decoder_forward that needs no encoder output (teacher inputs, prenets, hoisted attention-LSTM input
    GEMMs, bf16 shadows), on the CURRENT stream: t2_decoder_forward phase 1."""
    a = L.DecoderFwdArgs(dp.B, dp.T, dp.Tin, dp.Tsub, None, None, None, None, L.ptr(mels),
                         L.ptr(dp.mel), L.ptr(dp.gate), L.ptr(dp.align), L.ptr(dp.align_sub), L.ptr(dp.ws),
                         int(training), int(prenet_dropout), seed, 1)
    L.check(L.lib().t2_decoder_forward(C.byref(dims), C.byref(W), C.byref(a), L.stream()))


def side_join() -> None:
    """Make the current stream wait for the library's side stream (deferred weight gradients)."""
    L.check(L.lib().t2_side_join(L.stream()))


def decoder_backward(W: L.DecoderWeights, P: dict, dims: L.Dims, dp: DecoderPass, memory, memory_sub, d_mel, d_gate, *,
                     training: bool, prenet_dropout: bool, seed: int, d_align=None, d_align_sub=None, prefix="decoder.",
                     defer: Optional[list] = None):
    """Backward of decoder_forward.  P: the (reference-keyed) weight dict, used for gradient shapes.
    Returns (grads dict keyed like P, d_memory, d_memory_sub).  defer: a list -> the weight gradients are left on the
    library's side stream (d_memory* are complete on the current stream); the tensors that stream still uses are appended
    to the list and the caller must call side_join() before reading a gradient or dropping the list."""
    check_chain_status()
    single = dims.n_streams == 1
    dev = memory.device
    G = {prefix + k: torch.empty_like(P[prefix + k]) for k in L.decoder_param_keys(dims.attention_kind, single)}
    GS = L.decoder_grads(G, prefix, single, dims.attention_kind)
    bl = L.decoder_bwd_layout(dims, dp.B, dp.T, dp.Tin, dp.Tsub)
    bws = torch.empty(bl.total_floats, dtype=torch.float32, device=dev)
    d_mem = torch.empty_like(memory)
    d_mem_sub = None if single else torch.empty_like(memory_sub)
    a = L.DecoderBwdArgs(dp.B, dp.T, dp.Tin, dp.Tsub, L.ptr(memory), L.ptr(memory_sub), L.ptr(dp.align), L.ptr(dp.align_sub),
                         L.ptr(d_mel), L.ptr(d_gate), L.ptr(d_align), L.ptr(d_align_sub), L.ptr(d_mem), L.ptr(d_mem_sub),
                         L.ptr(dp.ws), L.ptr(bws), int(training), int(prenet_dropout), seed, int(defer is not None))
    L.check(L.lib().t2_decoder_backward(C.byref(dims), C.byref(W), C.byref(GS), C.byref(a), L.stream()))
    if defer is not None:
        defer.extend([bws, dp, memory, memory_sub, d_mel, d_gate, d_align, d_align_sub])   # not G: AccumulateGrad must be able to steal it
    return G, d_mem, d_mem_sub


def decoder_infer(W: L.DecoderWeights, dims: L.Dims, memory, memory_sub, *, max_steps: int, gate_threshold: float,
                  prenet_dropout: bool, seed: int = 0, poll_every: int = 0, mem_lengths=None, sub_lengths=None):
    """Autoregressive decode (Decoder.inference, model.py:430-492) for any B.
    poll_every: decoder steps between two looks at the stop counter (0 = the library's choice: 16 for per-step launches, 32 =
    one persistent launch per interval when the persistent decode loop runs).
    Returns (DecoderPass sized for max_steps, steps_run, stop_index[B] (int32, -1 = never stopped))."""
    check_chain_status()
    B, Tin, _ = memory.shape
    Tsub = 1 if memory_sub is None else memory_sub.shape[1]
    dp = DecoderPass(dims, B, max_steps, Tin, Tsub, memory.device)
    stop = torch.empty(B, dtype=torch.int32, device=memory.device)
    done = torch.empty(1, dtype=torch.int32, device=memory.device)
    steps = C.c_int(0)
    ml, sl = _i32(mem_lengths), _i32(sub_lengths)
    a = L.DecoderInferArgs(B, Tin, Tsub, max_steps, poll_every, gate_threshold, L.ptr(memory), L.ptr(memory_sub),
                           L.ptr(ml), L.ptr(sl), L.ptr(dp.mel), L.ptr(dp.gate), L.ptr(dp.align), L.ptr(dp.align_sub),
                           L.ptr(stop), L.ptr(done), L.ptr(dp.ws), int(prenet_dropout), seed, C.pointer(steps))
    L.check(L.lib().t2_decoder_infer(C.byref(dims), C.byref(W), C.byref(a), L.stream()))
    return dp, steps.value, stop


def finalize_bct(x_btc: torch.Tensor, lengths: Optional[torch.Tensor], fill: float) -> torch.Tensor:
    """[B,T,C] -> [B,C,T] with frames >= length set to `fill` (parse_output, model.py:531-541)."""
    B, T, Cc = x_btc.shape
    out = torch.empty(B, Cc, T, dtype=torch.float32, device=x_btc.device)
    ln = _i32(lengths)
    L.check(L.lib().t2_finalize_bct(L.ptr(x_btc), L.ptr(out), B, T, Cc, L.ptr(ln), fill, L.stream()))
    return out


def mask_bt_(x: torch.Tensor, lengths: torch.Tensor, fill: float) -> torch.Tensor:
    ln = _i32(lengths)
    L.check(L.lib().t2_mask_bt(L.ptr(x), x.shape[0], x.shape[1], L.ptr(ln), fill, L.stream()))
    return x


def mask_btc_(x: torch.Tensor, lengths: torch.Tensor, fill: float) -> torch.Tensor:
    """x[b,t,:] = fill for t >= lengths[b], in place (x: [B,T,C])."""
    ln = _i32(lengths)
    L.check(L.lib().t2_mask_btc(L.ptr(x), x.shape[0], x.shape[1], x.shape[2], L.ptr(ln), fill, L.stream()))
    return x
