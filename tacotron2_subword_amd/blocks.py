"""Autograd nodes for the encoder / postnet building blocks, each a thin host wrapper that
allocates outputs and calls the HIP library (no arithmetic in Python):

  linear          nn.Linear / LinearNorm                      -> t2_gemm_ex, t2_colsum
  embedding       nn.Embedding                                -> t2_embedding_forward/backward
  conv_bn_stack   [Conv1d(k) + BatchNorm1d + act + dropout]*n -> t2_conv_bn_forward/backward
  bilstm          packed / unpacked bidirectional nn.LSTM     -> t2_gemm_ex + t2_lstm_seq_forward/backward

Activations are channels-last: [B, T, C].
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
_WS = {}


def _scratch(dev, n_floats: int) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): kernels on one stream run in order, so reuse within a stream
    is safe; the two front ends run on different streams (model.BERT_Tacotron2._fronts) and must not share one."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
    t = _WS.get(key)
    if t is None or t.numel() < n_floats:
        t = torch.empty(int(n_floats * 1.25) + 1024, dtype=torch.float32, device=dev)
        _WS[key] = t
    return t


def gemm_ex(A, Bm, out, M, N, K, sam, sak, sbn, sbk, ldc, *, bias=None, act=0, alpha=1.0, beta=0.0, crow_mod=0, crow_mul=0,
            batch=1, bsA=0, bsB=0, bsC=0, splitk_ws=True):
    ws = _scratch(out.device, 8 << 20) if splitk_ws and beta == 0.0 else None
    a = L.GemmArgs(L.ptr(A), L.ptr(Bm), L.ptr(out), M, N, K, sam, sak, sbn, sbk, ldc, batch, bsA, bsB, bsC, alpha, beta,
                   L.ptr(bias), act, crow_mod, crow_mul, L.ptr(ws), 0 if ws is None else ws.numel() * 4, 0)
    L.check(L.lib().t2_gemm_ex(C.byref(a), L.stream()))
    return out


def colsum(x2d: torch.Tensor) -> torch.Tensor:
    M, N = x2d.shape
    out = torch.empty(N, dtype=torch.float32, device=x2d.device)
    sc = torch.empty(64 * N, dtype=torch.float32, device=x2d.device)
    L.check(L.lib().t2_colsum(L.ptr(x2d), x2d.stride(0), M, N, L.ptr(out), L.ptr(sc), L.stream()))
    return out


class _LinearFn(torch.autograd.Function):
    """y[M,N] = x[M,K] . W[N,K]^T + b."""

    @staticmethod
    def forward(ctx, x, W, b):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        M, K = x2.shape
        N = W.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        gemm_ex(x2, W, y, M, N, K, K, 1, W.stride(0), 1, N, bias=b)
        ctx.save_for_backward(x2, W)
        ctx.has_bias, ctx.shape = b is not None, shape
        return y.view(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        M, K = x2.shape
        N = W.shape[0]
        dy2 = dy.reshape(M, N).contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
            gemm_ex(dy2, W, dx, M, K, N, N, 1, 1, W.stride(0), K)                 # dy . W   (B(k=n, j=k) = W[n,k])
            dx = dx.view(ctx.shape)
        if ctx.needs_input_grad[1]:
            dW = torch.empty(N, K, dtype=torch.float32, device=dy.device)
            gemm_ex(dy2, x2, dW, N, K, M, 1, N, 1, K, K)                           # dy^T . x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        return dx, dW, db


def linear(x, W, b=None):
    return _LinearFn.apply(x, W, b)


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table):
        ids = ids.contiguous().long()
        rows, D = ids.numel(), table.shape[1]
        out = torch.empty(*ids.shape, D, dtype=torch.float32, device=table.device)
        L.check(L.lib().t2_embedding_forward(L.ptr(ids), L.ptr(table), L.ptr(out), rows, D, L.stream()))
        ctx.save_for_backward(ids)
        ctx.vocab = table.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dout = dout.contiguous()
        D = dout.shape[-1]
        dtable = torch.empty(ctx.vocab, D, dtype=torch.float32, device=dout.device)
        L.check(L.lib().t2_embedding_backward(L.ptr(ids), L.ptr(dout), L.ptr(dtable), ids.numel(), D, ctx.vocab, L.stream()))
        return None, dtable


def embedding(ids, table):
    return _EmbeddingFn.apply(ids, table)


class _ConvStackFn(torch.autograd.Function):
    """n x [Conv1d(k, same) + BatchNorm1d + act + dropout] on x[B,T,Cin]; optional residual (the input)
    added to the last layer's output.  params: (w, bias, gamma, beta) per layer; buffers: (run_mean, run_var)."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        x = x.contiguous()
        B, T, _ = x.shape
        n = len(cfg["acts"])
        dev = x.device
        saved, cur = [], x
        for i in range(n):
            w, bias, gamma, beta = params[4 * i:4 * i + 4]
            rm, rv = cfg["buffers"][i]
            Cout, Cin, K = w.shape
            M = B * T
            z = torch.empty(M, Cout, dtype=torch.float32, device=dev)
            y = torch.empty(B, T, Cout, dtype=torch.float32, device=dev)
            st = torch.empty(3, Cout, dtype=torch.float32, device=dev)      # mean, invstd, var
            nws = Cout * Cin * K + 4 + 128 * Cout + (M * Cin + Cout * Cin * K) // 2 + 4096    # + bf16 operand staging + split-K
            ws = _scratch(dev, nws)
            res = x if (cfg["residual"] and i == n - 1) else None
            a = L.ConvBnArgs(B, T, Cin, Cout, K, L.ptr(cur), L.ptr(w.detach()), L.ptr(bias.detach()), L.ptr(gamma.detach()),
                             L.ptr(beta.detach()), L.ptr(rm), L.ptr(rv), int(cfg["training"]), 1e-5, cfg["acts"][i],
                             cfg["drop_p"] if cfg["training"] else 0.0, cfg["seed"], cfg["site0"] + i, L.ptr(res),
                             L.ptr(z), L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(y), L.ptr(ws), ws.numel())
            L.check(L.lib().t2_conv_bn_forward(C.byref(a), L.stream()))
            saved.append((cur, z, st))
            cur = y
        ctx.cfg, ctx.saved_acts, ctx.params = cfg, saved, [p.detach() for p in params]
        return cur

    @staticmethod
    def backward(ctx, dy):
        cfg, params = ctx.cfg, ctx.params
        n = len(cfg["acts"])
        dev = dy.device
        dy = dy.contiguous()
        d_res = dy if cfg["residual"] else None
        grads = [None] * (4 * n)
        for i in range(n - 1, -1, -1):
            w, bias, gamma, beta = params[4 * i:4 * i + 4]
            xin, z, st = ctx.saved_acts[i]
            B, T, Cin = xin.shape
            Cout, _, K = w.shape
            M = B * T
            dw, db = torch.empty_like(w), torch.empty_like(bias)
            dg, dbt = torch.empty_like(gamma), torch.empty_like(beta)
            need_dx = i > 0 or ctx.needs_input_grad[0]
            dx = torch.empty(B, T, Cin, dtype=torch.float32, device=dev) if need_dx else None
            nws = M * Cout + Cout * Cin * K + 128 * Cout + 16 + (8 << 20) + (M * (Cout + K * max(Cin, Cout))) // 2 + Cout * Cin * K
            ws = _scratch(dev, nws)
            a = L.ConvBnBwdArgs(B, T, Cin, Cout, K, L.ptr(xin), L.ptr(w), L.ptr(gamma), L.ptr(beta), L.ptr(z), L.ptr(st[0]), L.ptr(st[1]),
                                int(cfg["training"]), 1e-5, cfg["acts"][i], cfg["drop_p"] if cfg["training"] else 0.0, cfg["seed"],
                                cfg["site0"] + i, L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(dg), L.ptr(dbt), L.ptr(dx), 0,
                                L.ptr(ws), ws.numel())
            L.check(L.lib().t2_conv_bn_backward(C.byref(a), L.stream()))
            grads[4 * i:4 * i + 4] = [dw, db, dg, dbt]
            dy = dx
        if dy is not None and d_res is not None:
            dy = dy + d_res
        ctx.saved_acts = None
        return (dy, None) + tuple(grads)


def conv_bn_stack(x, layers, acts, *, training, drop_p, seed, site0, residual=False):
    """layers: list of (conv_module, bn_module) pairs (ConvNorm.conv, nn.BatchNorm1d)."""
    params, buffers = [], []
    for conv, bn in layers:
        params += [conv.weight, conv.bias, bn.weight, bn.bias]
        buffers.append((bn.running_mean, bn.running_var))
        if training:
            bn.num_batches_tracked += 1
    cfg = dict(acts=list(acts), buffers=buffers, training=bool(training), drop_p=float(drop_p), seed=int(seed), site0=int(site0),
               residual=bool(residual))
    return _ConvStackFn.apply(x, cfg, *params)


class _BiLstmFn(torch.autograd.Function):
    """x[B,T,E] -> [B,T,2H]: one-layer bidirectional LSTM; lengths (int32 [B]) gives packed-sequence
    semantics (model.py:104-112), None the unpacked form (model.py:122-123)."""

    @staticmethod
    def forward(ctx, x, lengths, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        B, T, E = x.shape
        H = w_hh.shape[1]
        dev = x.device
        xt = x.transpose(0, 1).contiguous()                              # time-major rows (t,b)
        pre = torch.empty(2, T * B, 4 * H, dtype=torch.float32, device=dev)
        bsum = torch.stack((b_ih.detach() + b_hh.detach(), b_ih_r.detach() + b_hh_r.detach()))
        for d, W in enumerate((w_ih, w_ih_r)):
            gemm_ex(xt, W.detach(), pre[d], T * B, 4 * H, E, E, 1, W.stride(0), 1, 4 * H, bias=bsum[d])
        out = torch.empty(T, B, 2 * H, dtype=torch.float32, device=dev)
        cells = torch.empty(2, T, B, H, dtype=torch.float32, device=dev)
        gates = torch.empty(2, T, B, 4 * H, dtype=torch.float32, device=dev)
        a = L.LstmSeqArgs()
        a.nstreams, a.B, a.T, a.H = 2, B, T, H
        a.pre[0], a.pre[1] = pre[0].data_ptr(), pre[1].data_ptr()
        a.w_hh[0], a.w_hh[1] = L.ptr(w_hh.detach()), L.ptr(w_hh_r.detach())
        a.reverse[0], a.reverse[1] = 0, 1
        a.lengths = L.ptr(lengths)
        a.h[0], a.h[1] = out.data_ptr(), out.data_ptr() + 4 * H
        a.ldh = 2 * H
        a.c[0], a.c[1] = cells[0].data_ptr(), cells[1].data_ptr()
        a.gates[0], a.gates[1] = gates[0].data_ptr(), gates[1].data_ptr()
        # exchange space of the persistent chain (one launch for all steps); its own tensor, not the shared scratch: the
        # two encoders' chains are in flight at the same time on their two streams
        xws = torch.empty(int(L.lib().t2_lstm_seq_chain_ws_floats(2, B, H, 0)), dtype=torch.float32, device=dev)
        a.ws, a.ws_floats = L.ptr(xws), xws.numel()
        L.check(L.lib().t2_lstm_seq_forward(C.byref(a), L.stream()))
        ctx.save_for_backward(xt, out, cells, gates, w_ih, w_hh, w_ih_r, w_hh_r)
        ctx.dims = (B, T, E, H)
        return out.transpose(0, 1).contiguous()

    @staticmethod
    def backward(ctx, dout):
        xt, out, cells, gates, w_ih, w_hh, w_ih_r, w_hh_r = ctx.saved_tensors
        B, T, E, H = ctx.dims
        dev = dout.device
        dh = dout.transpose(0, 1).contiguous()                           # [T,B,2H]
        dpre = torch.empty(2, T * B, 4 * H, dtype=torch.float32, device=dev)
        dwhh = torch.empty(2, 4 * H, H, dtype=torch.float32, device=dev)
        nws = max(2 * (B * H + 8 * B * H) + 64, int(L.lib().t2_lstm_seq_chain_ws_floats(2, B, H, 1))) + (8 << 20)
        ws = _scratch(dev, nws)
        a = L.LstmSeqBwdArgs()
        a.nstreams, a.B, a.T, a.H = 2, B, T, H
        a.w_hh[0], a.w_hh[1] = L.ptr(w_hh), L.ptr(w_hh_r)
        a.reverse[0], a.reverse[1] = 0, 1
        a.h[0], a.h[1] = out.data_ptr(), out.data_ptr() + 4 * H
        a.ldh = 2 * H
        a.c[0], a.c[1] = cells[0].data_ptr(), cells[1].data_ptr()
        a.gates[0], a.gates[1] = gates[0].data_ptr(), gates[1].data_ptr()
        a.dh[0], a.dh[1] = dh.data_ptr(), dh.data_ptr() + 4 * H
        a.lddh = 2 * H
        a.dpre[0], a.dpre[1] = dpre[0].data_ptr(), dpre[1].data_ptr()
        a.dw_hh[0], a.dw_hh[1] = dwhh[0].data_ptr(), dwhh[1].data_ptr()
        a.ws, a.ws_floats = L.ptr(ws), ws.numel()
        L.check(L.lib().t2_lstm_seq_backward(C.byref(a), L.stream()))
        TB = T * B
        dxt = torch.empty(TB, E, dtype=torch.float32, device=dev)
        res = []
        for d, W in enumerate((w_ih, w_ih_r)):
            gemm_ex(dpre[d], W, dxt, TB, E, 4 * H, 4 * H, 1, 1, W.stride(0), E, beta=float(d))       # dpre . W_ih
            dW = torch.empty(4 * H, E, dtype=torch.float32, device=dev)
            gemm_ex(dpre[d], xt, dW, 4 * H, E, TB, 1, 4 * H, 1, E, E)                                # dpre^T . x
            db = colsum(dpre[d])
            res.append((dW, dwhh[d], db, db))
        dx = dxt.view(T, B, E).transpose(0, 1).contiguous()
        return (dx, None) + res[0] + res[1]


def bilstm(x, lengths, lstm):
    """lstm: nn.LSTM(E, H, 1, batch_first=True, bidirectional=True) used as a parameter container."""
    ln = None if lengths is None else lengths.to(dtype=torch.int32).contiguous()
    return _BiLstmFn.apply(x, ln, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                           lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse)
