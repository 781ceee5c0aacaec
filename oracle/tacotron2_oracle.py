"""CPU oracle for the Tacotron2 (dual-stream BERT_Tacotron2) acoustic-model hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``tacotron2_subword_amd/`` may import this
file; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and only as the checker / the reported CPU baseline.

This is a *functional restatement* (plain fp32 torch-CPU tensor arithmetic, no
``nn.Module``) of the reference's op sequence.  Every function cites the reference
``file:line`` it follows (paths relative to the reference checkout).  The parameter
dictionary ``P`` uses the reference's ``state_dict`` key names verbatim.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's own
``model.py`` in the build container, loads recipe weights (``oracle/recipe.py``) and
records inputs/outputs/gradients into ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors (the reference itself ships no tests or
golden vectors for this path — SURVEY.md §4, §8c).

Randomness (dropout keep-masks, SMA pre-sigmoid noise) is never drawn here: it is an
*input* (``rnd`` dict), so the same masks can be replayed through the reference, this
oracle and the HIP path.  A missing entry means "identity / no noise".
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration (values restated from hparams.py:55-95)
# --------------------------------------------------------------------------------------
def default_hparams() -> dict:
    return dict(
        n_mel_channels=80,                      # hparams.py:55
        n_symbols=313, sub_n_symbols=5500,      # hparams.py:62-63
        symbols_embedding_dim=512,              # hparams.py:65
        attention="StepwiseMonotonicAttention",  # hparams.py:67
        encoder_kernel_size=5, encoder_n_convolutions=3, encoder_embedding_dim=512,  # :70-72
        BERT_embedding_dim=768,                 # :73
        n_frames_per_step=1, decoder_rnn_dim=1024, prenet_dim=256,  # :76-78
        max_decoder_steps=1000, gate_threshold=0.001,  # :79-80
        p_attention_dropout=0.1, p_decoder_dropout=0.1,  # :81-82
        attention_rnn_dim=1024, attention_dim=128,  # :85-86
        attention_location_n_filters=32, attention_location_kernel_size=31,  # :89-90
        postnet_embedding_dim=512, postnet_kernel_size=5, postnet_n_convolutions=5,  # :93-95
        mask_padding=True,                      # :105
    )


def get_mask_from_lengths(lengths: Tensor, max_len: Optional[int] = None) -> Tensor:
    """utils.py:10-14 (device-agnostic): True where index < length."""
    if max_len is None:
        max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, dtype=torch.long, device=lengths.device)
    return ids < lengths.unsqueeze(1)


def _drop(x: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    """F.dropout with an explicit keep-mask (1 = keep).  keep=None -> identity."""
    if keep is None:
        return x
    return x * keep.to(x.dtype) * (1.0 / (1.0 - p))


def _get(rnd: Optional[dict], key: str, idx=None):
    if rnd is None or key not in rnd or rnd[key] is None:
        return None
    v = rnd[key]
    return v if idx is None else v[idx]


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def prenet(x: Tensor, w1: Tensor, w2: Tensor, keep1=None, keep2=None) -> Tensor:
    """model.py:13-24.  Two bias-free linears, ReLU, dropout p=0.5 (always on in the
    reference; here on iff a keep mask is supplied)."""
    x = _drop(F.relu(F.linear(x, w1)), keep1, 0.5)
    x = _drop(F.relu(F.linear(x, w2)), keep2, 0.5)
    return x


def batch_norm_1d(x: Tensor, P: Dict[str, Tensor], prefix: str, training: bool,
                  new_stats: Optional[dict], eps: float = 1e-5, momentum: float = 0.1) -> Tensor:
    """nn.BatchNorm1d on [B,C,T] (model.py:42,53,62,89).  Training: batch statistics
    over (B,T) including padded frames (the reference does not mask them), biased var
    for normalisation, unbiased var for the running estimate."""
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    if training:
        mean = x.mean(dim=(0, 2))
        var = x.var(dim=(0, 2), unbiased=False)
        if new_stats is not None:
            n = x.shape[0] * x.shape[2]
            unb = var * (n / max(n - 1, 1))
            new_stats[prefix + ".running_mean"] = (1 - momentum) * P[prefix + ".running_mean"] + momentum * mean.detach()
            new_stats[prefix + ".running_var"] = (1 - momentum) * P[prefix + ".running_var"] + momentum * unb.detach()
            new_stats[prefix + ".num_batches_tracked"] = P[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    xn = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + eps)
    return xn * w[None, :, None] + b[None, :, None]


def conv_bn(x: Tensor, P, conv_prefix: str, bn_prefix: str, training: bool, new_stats) -> Tensor:
    w = P[conv_prefix + ".conv.weight"]
    b = P.get(conv_prefix + ".conv.bias")
    pad = (w.shape[2] - 1) // 2                 # layers.py:25-27
    y = F.conv1d(x, w, b, padding=pad)
    return batch_norm_1d(y, P, bn_prefix, training, new_stats)


def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTMCell: gate order i,f,g,o (model.py:150-156,193-195)."""
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    H = h.shape[1]
    i, f, g, o = gates[:, :H], gates[:, H:2 * H], gates[:, 2 * H:3 * H], gates[:, 3 * H:]
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c2 = f * c + i * g
    h2 = o * torch.tanh(c2)
    return h2, c2


def bilstm(x: Tensor, lengths: Optional[Tensor], P, prefix: str) -> Tensor:
    """nn.LSTM(512, 256, 1, batch_first, bidirectional) over a packed sequence
    (model.py:93-95,104-112) or, with lengths=None, the un-packed inference form
    (model.py:122-123).  Output is re-padded to max(lengths); positions beyond an
    item's length are zero; the reverse direction starts at the item's last token."""
    B, T, _ = x.shape
    if lengths is None:
        lengths = torch.full((B,), T, dtype=torch.long)
    Tm = int(lengths.max().item())
    outs = []
    for suffix, order in (("", range(Tm)), ("_reverse", range(Tm - 1, -1, -1))):
        w_ih, w_hh = P[f"{prefix}.weight_ih_l0{suffix}"], P[f"{prefix}.weight_hh_l0{suffix}"]
        b_ih, b_hh = P[f"{prefix}.bias_ih_l0{suffix}"], P[f"{prefix}.bias_hh_l0{suffix}"]
        H = w_hh.shape[1]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        out = [None] * Tm
        for t in order:
            act = (lengths > t).to(x.dtype).unsqueeze(1)
            h2, c2 = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
            h = act * h2 + (1 - act) * h
            c = act * c2 + (1 - act) * c
            out[t] = act * h2
        outs.append(torch.stack(out, dim=1))
    return torch.cat(outs, dim=2)


def encoder(x: Tensor, lengths: Optional[Tensor], P, prefix: str, training: bool,
            keep=None, new_stats=None) -> Tensor:
    """Encoder.forward / .inference (model.py:97-125).  x: [B,512,T_in] embedded input."""
    for i in range(3):
        x = F.relu(conv_bn(x, P, f"{prefix}.convolutions.{i}.0", f"{prefix}.convolutions.{i}.1",
                           training, new_stats))
        x = _drop(x, None if keep is None else keep[i], 0.5)
    return bilstm(x.transpose(1, 2), lengths, P, f"{prefix}.lstm")


def postnet(x: Tensor, P, training: bool, keep=None, new_stats=None, prefix="postnet") -> Tensor:
    """Postnet.forward (model.py:65-70). x: [B,80,T]."""
    n = 5
    for i in range(n):
        x = conv_bn(x, P, f"{prefix}.convolutions.{i}.0", f"{prefix}.convolutions.{i}.1", training, new_stats)
        if i < n - 1:
            x = torch.tanh(x)
        x = _drop(x, None if keep is None else keep[i], 0.5)
    return x


# --------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------
def sma_energies(query: Tensor, pm: Tensor, P, prefix: str) -> Tensor:
    """StepwiseMonotonicAttention.get_energies (attention.py:365-372)."""
    pq = F.linear(query.unsqueeze(1), P[prefix + ".query_layer.linear_layer.weight"])
    return F.linear(torch.tanh(pq + pm), P[prefix + ".v.weight"]).squeeze(-1)


def sma_step(query, memory, pm, prev_align, mask, P, prefix, noise=None):
    """StepwiseMonotonicAttention.forward (attention.py:374-398): energies, mask -> -inf,
    optional pre-sigmoid noise*2.0 (training, :346-348), p = sigmoid(e), recurrence (:337),
    context = alignment @ memory."""
    e = sma_energies(query, pm, P, prefix)
    if mask is not None:
        e = e.masked_fill(mask, -float("inf"))
    if noise is not None:
        e = e + noise * 2.0
    p = torch.sigmoid(e)
    pad = prev_align.new_zeros(prev_align.size(0), 1)
    align = prev_align * p + torch.cat((pad, prev_align[:, :-1] * (1.0 - p[:, :-1])), dim=1)
    ctx = torch.bmm(align.unsqueeze(1), memory).squeeze(1)
    return ctx, align


def lsa_step(query, memory, pm, w_prev, w_cum, mask, P, prefix, score_mask_value=-float("inf")):
    """LocationSensitiveAttention.forward (attention.py:64-85) with LocationLayer (:7-23)."""
    cat = torch.stack((w_prev, w_cum), dim=1)                       # model.py:351
    cw = P[prefix + ".location_layer.location_conv.conv.weight"]
    loc = F.conv1d(cat, cw, None, padding=(cw.shape[2] - 1) // 2).transpose(1, 2)
    loc = F.linear(loc, P[prefix + ".location_layer.location_dense.linear_layer.weight"])
    pq = F.linear(query.unsqueeze(1), P[prefix + ".query_layer.linear_layer.weight"])
    e = F.linear(torch.tanh(pq + loc + pm), P[prefix + ".v.linear_layer.weight"]).squeeze(-1)
    if mask is not None:
        e = e.masked_fill(mask, score_mask_value)
    w = F.softmax(e, dim=1)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    return ctx, w


def fa2_step(query, memory, pm, w_prev, w_cum, mask, P, prefix, log_alpha):
    """ForwardAttentionV2.forward (attention.py:128-151) with the log_alpha the caller hands it: model.py:266-270
    initialises it to [0, -1e4, -1e4, ...] and :355 passes that same tensor at every step (the new log_alpha is
    commented out of the return, :151), so the 'forward' recursion never advances.  Restated literally."""
    cat = torch.stack((w_prev, w_cum), dim=1)
    cw = P[prefix + ".location_layer.location_conv.conv.weight"]
    loc = F.conv1d(cat, cw, None, padding=(cw.shape[2] - 1) // 2).transpose(1, 2)
    loc = F.linear(loc, P[prefix + ".location_layer.location_dense.linear_layer.weight"])
    pq = F.linear(query.unsqueeze(1), P[prefix + ".query_layer.linear_layer.weight"])
    log_energy = F.linear(torch.tanh(pq + loc + pm), P[prefix + ".v.linear_layer.weight"]).squeeze(-1)
    score_mask_value = -float(1e20)                                  # attention.py:100
    if mask is not None:
        log_energy = log_energy.masked_fill(mask, score_mask_value)
    shifted = F.pad(log_alpha[:, :-1], [1, 0], "constant", score_mask_value)
    biased = torch.logsumexp(torch.cat([log_alpha.unsqueeze(2), shifted.unsqueeze(2)], 2), 2)
    w = F.softmax(biased + log_energy, dim=1)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    return ctx, w


def gmm_step(query, memory, mu_prev, mask, P, prefix):
    """GMMAttention.forward, version '2', K = 5 (attention.py:427-506).  mu_prev: [B,K,1] (init_attention: zeros)."""
    K, eps = 5, 1e-5
    hid = torch.tanh(F.linear(query, P[prefix + ".mlp.0.weight"], P[prefix + ".mlp.0.bias"]))
    interm = F.linear(hid, P[prefix + ".mlp.2.weight"], P[prefix + ".mlp.2.bias"]).view(query.size(0), -1, K)
    omega_hat, delta_hat, sigma_hat = (c.squeeze(1) for c in interm.chunk(3, dim=1))
    sigma = (F.softplus(sigma_hat) + eps).unsqueeze(-1)
    delta = F.softplus(delta_hat).unsqueeze(-1)
    omega = F.softmax(omega_hat, dim=-1).unsqueeze(-1)
    Z = torch.sqrt(2 * math.pi * sigma ** 2)
    mu = mu_prev + delta
    j = torch.arange(0, memory.size(1), device=memory.device).view(1, 1, -1)
    alignment = torch.sum(omega / Z * torch.exp(-(j - mu) ** 2 / (sigma ** 2) / 2), 1)
    if mask is not None:
        alignment = alignment.masked_fill(mask, -float("inf"))
    w = F.softmax(alignment, dim=1)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    return ctx, w, mu


def dca_step(query, memory, align_prev, mask, P, prefix):
    """DynamicConvolutionAttention.forward (attention.py:236-289).  align_prev: [B,Tin] (init_attention: one-hot at 0)."""
    Kd, Cd = 21, 8
    prior = P[prefix + ".P"]
    p = F.conv1d(F.pad(align_prev.unsqueeze(1), (prior.numel() - 1, 0)), prior.view(1, 1, -1))
    p = torch.log(p.clamp_min(1e-6)).squeeze(1)
    G = F.linear(torch.tanh(F.linear(query, P[prefix + ".W.weight"], P[prefix + ".W.bias"])), P[prefix + ".V.weight"])
    g = F.conv1d(align_prev.unsqueeze(0), G.view(-1, 1, Kd), padding=(Kd - 1) // 2, groups=query.size(0))
    g = g.view(query.size(0), Cd, -1).transpose(1, 2)
    f = F.conv1d(align_prev.unsqueeze(1), P[prefix + ".F.weight"], padding=(Kd - 1) // 2).transpose(1, 2)
    e = F.linear(torch.tanh(F.linear(f, P[prefix + ".U.weight"]) + F.linear(g, P[prefix + ".T.weight"], P[prefix + ".T.bias"])),
                 P[prefix + ".v.weight"]).squeeze(-1) + p
    if mask is not None:
        e = e.masked_fill(mask, -float("inf"))
    w = F.softmax(e, dim=1)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    return ctx, w


def _attend(kind, query, memory, pm, w_prev, w_cum, mask, P, prefix, st=None):
    if kind == "DynamicConvolutionAttention":
        key = "a_" + prefix
        a_prev = st.gmm.get(key)
        if a_prev is None:                                       # attention.py:232-234
            a_prev = memory.new_zeros(memory.shape[0], memory.shape[1])
            a_prev[:, 0] = 1.0
        ctx, w = dca_step(query, memory, a_prev, mask, P, prefix)
        st.gmm[key] = w
        return ctx, w
    if kind == "GMMAttention":
        key = "mu_" + prefix
        mu_prev = st.gmm.get(key)
        if mu_prev is None:
            mu_prev = memory.new_zeros(memory.shape[0], 5, 1)
        ctx, w, st.gmm[key] = gmm_step(query, memory, mu_prev, mask, P, prefix)
        return ctx, w
    if kind == "ForwardAttentionV2":
        la = memory.new_full((memory.shape[0], memory.shape[1]), -float(1e4))
        la[:, 0] = 0.0
        return fa2_step(query, memory, pm, w_prev, w_cum, mask, P, prefix, la)
    return lsa_step(query, memory, pm, w_prev, w_cum, mask, P, prefix)


# --------------------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------------------
class DecState:
    """Decoder.initialize_decoder_states (model.py:223-270).  memory_sub=None gives the classic
    single-stream decoder (the API the reference's stale GTA.py expects; no reference counterpart,
    so that variant is NOT pinned by golden vectors)."""

    def __init__(self, memory, memory_sub, mask, mask_sub, P, hp):
        B, Tin, _ = memory.shape
        self.single = memory_sub is None
        if self.single:
            memory_sub = memory[:, :1]
        Tsub = memory_sub.shape[1]
        Ha, Hd, E = hp["attention_rnn_dim"], hp["decoder_rnn_dim"], hp["encoder_embedding_dim"]
        z = lambda *s: memory.new_zeros(*s)
        self.ah, self.ac, self.ahb, self.acb = z(B, Ha), z(B, Ha), z(B, Ha), z(B, Ha)
        self.dh, self.dc = z(B, Hd), z(B, Hd)
        self.w, self.wcum, self.ctx = z(B, Tin), z(B, Tin), z(B, E)
        self.wb, self.wcumb, self.ctxb = z(B, Tsub), z(B, Tsub), z(B, E)
        self.memory, self.memory_sub = memory, memory_sub
        self.pm = F.linear(memory, P["decoder.attention_layer.memory_layer.linear_layer.weight"])
        self.pmb = None if self.single else F.linear(memory_sub, P["decoder.attention_layer_bert.memory_layer.linear_layer.weight"])
        self.mask, self.mask_sub = mask, mask_sub
        self.gmm = {}                               # GMMAttention: mixture means per attention module (init_attention: zeros)
        self.sma = hp["attention"] == "StepwiseMonotonicAttention"
        if self.sma:                                # attention.py:324-328
            self.align = z(B, Tin); self.align[:, 0] = 1.0
            self.alignb = z(B, Tsub); self.alignb[:, 0] = 1.0


def decode_step(st: DecState, xp: Tensor, xb: Tensor, P, hp, rnd=None, t: int = 0, trace=None):
    """Decoder.decode (model.py:322-390)."""
    pa, pd = hp["p_attention_dropout"], hp["p_decoder_dropout"]
    # attention LSTMs (model.py:337-346)
    st.ah, st.ac = lstm_cell(torch.cat((xp, st.ctx), -1), st.ah, st.ac,
                             P["decoder.attention_rnn.weight_ih"], P["decoder.attention_rnn.weight_hh"],
                             P["decoder.attention_rnn.bias_ih"], P["decoder.attention_rnn.bias_hh"])
    st.ah = _drop(st.ah, _get(rnd, "att_h_keep", t), pa)
    st.ac = _drop(st.ac, _get(rnd, "att_c_keep", t), pa)
    if st.single:
        return _decode_step_single(st, P, hp, rnd, t, trace)
    st.ahb, st.acb = lstm_cell(torch.cat((xb, st.ctxb), -1), st.ahb, st.acb,
                               P["decoder.attention_rnn_bert.weight_ih"], P["decoder.attention_rnn_bert.weight_hh"],
                               P["decoder.attention_rnn_bert.bias_ih"], P["decoder.attention_rnn_bert.bias_hh"])
    st.ahb = _drop(st.ahb, _get(rnd, "att_h_bert_keep", t), pa)
    st.acb = _drop(st.acb, _get(rnd, "att_c_bert_keep", t), pa)
    # attention (model.py:351-359)
    if st.sma:
        st.ctx, st.align = sma_step(st.ah, st.memory, st.pm, st.align, st.mask, P,
                                    "decoder.attention_layer", _get(rnd, "sma_noise", t))
        st.w = st.align
        st.ctxb, st.alignb = sma_step(st.ahb, st.memory_sub, st.pmb, st.alignb, st.mask_sub, P,
                                      "decoder.attention_layer_bert", _get(rnd, "sma_noise_bert", t))
        st.wb = st.alignb
    else:
        st.ctx, st.w = _attend(hp["attention"], st.ah, st.memory, st.pm, st.w, st.wcum, st.mask, P, "decoder.attention_layer", st)
        st.ctxb, st.wb = _attend(hp["attention"], st.ahb, st.memory_sub, st.pmb, st.wb, st.wcumb, st.mask_sub, P,
                                  "decoder.attention_layer_bert", st)
    st.wcum = st.wcum + st.w
    st.wcumb = st.wcumb + st.wb
    # decoder LSTM (model.py:362-373)
    din = torch.cat((st.ah, st.ctx, st.ahb, st.ctxb), -1)
    st.dh, st.dc = lstm_cell(din, st.dh, st.dc,
                             P["decoder.decoder_rnn.weight_ih"], P["decoder.decoder_rnn.weight_hh"],
                             P["decoder.decoder_rnn.bias_ih"], P["decoder.decoder_rnn.bias_hh"])
    st.dh = _drop(st.dh, _get(rnd, "dec_h_keep", t), pd)
    st.dc = _drop(st.dc, _get(rnd, "dec_c_keep", t), pd)
    # projections (model.py:382-388)
    dhc = torch.cat((st.dh, st.ctx, st.ctxb), dim=1)
    mel = F.linear(dhc, P["decoder.linear_projection.linear_layer.weight"], P["decoder.linear_projection.linear_layer.bias"])
    gate = F.linear(dhc, P["decoder.gate_layer.linear_layer.weight"], P["decoder.gate_layer.linear_layer.bias"])
    if trace is not None:
        trace.append(dict(att_h=st.ah, att_c=st.ac, att_h_bert=st.ahb, att_c_bert=st.acb, ctx=st.ctx,
                          ctx_bert=st.ctxb, w=st.w, w_bert=st.wb, dec_h=st.dh, dec_c=st.dc, mel=mel, gate=gate))
    return mel, gate, st.w, st.wb


def _decode_step_single(st, P, hp, rnd, t, trace):
    """Classic single-stream remainder of Decoder.decode: attention, decoder LSTM on [att_h|ctx],
    projections on [dec_h|ctx]."""
    pd = hp["p_decoder_dropout"]
    if st.sma:
        st.ctx, st.align = sma_step(st.ah, st.memory, st.pm, st.align, st.mask, P, "decoder.attention_layer", _get(rnd, "sma_noise", t))
        st.w = st.align
    else:
        st.ctx, st.w = _attend(hp["attention"], st.ah, st.memory, st.pm, st.w, st.wcum, st.mask, P, "decoder.attention_layer", st)
    st.wcum = st.wcum + st.w
    st.dh, st.dc = lstm_cell(torch.cat((st.ah, st.ctx), -1), st.dh, st.dc,
                             P["decoder.decoder_rnn.weight_ih"], P["decoder.decoder_rnn.weight_hh"],
                             P["decoder.decoder_rnn.bias_ih"], P["decoder.decoder_rnn.bias_hh"])
    st.dh = _drop(st.dh, _get(rnd, "dec_h_keep", t), pd)
    st.dc = _drop(st.dc, _get(rnd, "dec_c_keep", t), pd)
    dhc = torch.cat((st.dh, st.ctx), dim=1)
    mel = F.linear(dhc, P["decoder.linear_projection.linear_layer.weight"], P["decoder.linear_projection.linear_layer.bias"])
    gate = F.linear(dhc, P["decoder.gate_layer.linear_layer.weight"], P["decoder.gate_layer.linear_layer.bias"])
    return mel, gate, st.w, st.w


def decoder_forward(memory, memory_sub, mels, mem_lengths, sub_lengths, P, hp, rnd=None, trace=None):
    """Decoder.forward (model.py:392-428), teacher forced.  mels: [B,80,T]."""
    B = memory.shape[0]
    x = mels.transpose(1, 2).transpose(0, 1)                       # [T,B,80]  (:283-288)
    go = memory.new_zeros(1, B, hp["n_mel_channels"])
    x = torch.cat((go, x), dim=0)                                   # [T+1,B,80] (:410-411)
    T = x.shape[0] - 1
    k = lambda name, i: _get(rnd, name, i)
    xp = prenet(x[:T], P["decoder.prenet.layers.0.linear_layer.weight"], P["decoder.prenet.layers.1.linear_layer.weight"],
                k("prenet_keep", 0), k("prenet_keep", 1))
    single = memory_sub is None
    xb = xp if single else prenet(x[:T], P["decoder.prenet_bert.layers.0.linear_layer.weight"],
                                  P["decoder.prenet_bert.layers.1.linear_layer.weight"], k("prenet_bert_keep", 0), k("prenet_bert_keep", 1))
    st = DecState(memory, memory_sub, ~get_mask_from_lengths(mem_lengths, memory.shape[1]),
                  None if single else ~get_mask_from_lengths(sub_lengths, memory_sub.shape[1]), P, hp)
    mel_o, gate_o, al, alb = [], [], [], []
    for t in range(T):
        m, g, w, wb = decode_step(st, xp[t], xb[t], P, hp, rnd, t, trace)
        mel_o.append(m); gate_o.append(g.squeeze(1)); al.append(w); alb.append(wb)
    mel_o = torch.stack(mel_o).transpose(0, 1).transpose(1, 2)      # [B,80,T]  (:314-318)
    gate_o = torch.stack(gate_o).transpose(0, 1).contiguous()       # [B,T]
    return mel_o, gate_o, torch.stack(al).transpose(0, 1), torch.stack(alb).transpose(0, 1)


def decoder_inference(memory, memory_sub, P, hp, max_decoder_steps=None, gate_threshold=None,
                      prenet_keep=None, prenet_bert_keep=None):
    """Decoder.inference (model.py:430-492) for B == 1 (the only batch the reference
    supports: the stop test at :461,480 takes bool() of a [B,1] tensor).  Optional
    per-step prenet keep masks ([steps,2,1,256]) replay the always-on prenet dropout."""
    assert memory.shape[0] == 1
    single = memory_sub is None
    mds = hp["max_decoder_steps"] if max_decoder_steps is None else max_decoder_steps
    thr = hp["gate_threshold"] if gate_threshold is None else gate_threshold
    st = DecState(memory, memory_sub, None, None, P, hp)
    x = memory.new_zeros(1, hp["n_mel_channels"])
    mel_o, gate_o, al, alb = [], [], [], []
    flag = True
    while True:
        i = len(mel_o)
        kp = (None, None) if prenet_keep is None else (prenet_keep[i, 0], prenet_keep[i, 1])
        kb = (None, None) if prenet_bert_keep is None else (prenet_bert_keep[i, 0], prenet_bert_keep[i, 1])
        xp = prenet(x, P["decoder.prenet.layers.0.linear_layer.weight"], P["decoder.prenet.layers.1.linear_layer.weight"], *kp)
        xb = xp if single else prenet(x, P["decoder.prenet_bert.layers.0.linear_layer.weight"],
                                      P["decoder.prenet_bert.layers.1.linear_layer.weight"], *kb)
        m, g, w, wb = decode_step(st, xp, xb, P, hp)
        mel_o.append(m); gate_o.append(g); al.append(w); alb.append(wb)
        if torch.sigmoid(g).item() > thr:
            break
        if len(mel_o) == mds:
            flag = False
            break
        x = m
    mel_o = torch.stack(mel_o).transpose(0, 1).transpose(1, 2)
    gate_o = torch.stack(gate_o).transpose(0, 1).contiguous()       # [1,T',1]
    return mel_o, gate_o, torch.stack(al).transpose(0, 1), torch.stack(alb).transpose(0, 1), flag


# --------------------------------------------------------------------------------------
# full model
# --------------------------------------------------------------------------------------
def front_end(P, hp, ids, lengths, cls, which: str, training: bool, rnd=None, new_stats=None) -> Tensor:
    """embedding -> encoder -> cat CLS -> linear converter (model.py:546-554 / 563-572)."""
    sub = which == "sub"
    emb = F.embedding(ids, P["embedding_sub.weight" if sub else "embedding.weight"]).transpose(1, 2)
    enc = encoder(emb, lengths, P, "encoder_sub" if sub else "encoder", training,
                  _get(rnd, "encsub_keep" if sub else "enc_keep"), new_stats)
    Tm = enc.shape[1]
    name = "linear_converter_sub" if sub else "linear_converter"
    return F.linear(torch.cat([enc, cls[:, :Tm]], 2), P[name + ".linear_layer.weight"], P[name + ".linear_layer.bias"])


def forward(P, hp, x, training: bool = False, rnd=None, new_stats=None, trace=None):
    """BERT_Tacotron2.forward + parse_output (model.py:531-560).
    x = (text, text_lengths, bert_lengths, mels, (max_in, max_out), output_lengths,
         sub_ids, phoneme_cls, bert_cls)."""
    text, tl, bl, mels, _, ol, sub_ids, pcls, bcls = x
    mem = front_end(P, hp, text, tl, pcls, "phone", training, rnd, new_stats)
    mem_sub = front_end(P, hp, sub_ids, bl, bcls, "sub", training, rnd, new_stats)
    mel, gate, al, alb = decoder_forward(mem, mem_sub, mels, tl, bl, P, hp, rnd, trace)
    post = mel + postnet(mel, P, training, _get(rnd, "post_keep"), new_stats)
    if hp["mask_padding"] and ol is not None:
        # model.py:537-539 fills IN PLACE on .data, i.e. behind autograd's back.  Two visible
        # consequences that a drop-in must reproduce: (a) no gradient is blocked by the mask
        # (harmless: targets are 0 / 1 there), (b) the postnet's first conv has already saved
        # `mel` as its input, so its weight gradient is computed from the MASKED mel.
        m = ~get_mask_from_lengths(ol, mel.shape[2])
        mel.data.masked_fill_(m[:, None, :].expand_as(mel), 0.0)
        post.data.masked_fill_(m[:, None, :].expand_as(post), 0.0)
        gate.data.masked_fill_(m, 1e3)
    return [mel, post, gate, al, alb]


def inference(P, hp, ids, sub_ids, pcls, bcls, **kw):
    """BERT_Tacotron2.inference (model.py:562-582), B == 1, eval-mode BN."""
    mem = front_end(P, hp, ids, None, pcls, "phone", False)
    mem_sub = front_end(P, hp, sub_ids, None, bcls, "sub", False)
    mel, gate, al, alb, flag = decoder_inference(mem, mem_sub, P, hp, **kw)
    post = mel + postnet(mel, P, False)
    return [mel, post, gate, al, alb, flag]


def loss(y_pred, y):
    """Tacotron2Loss.forward default branch (loss_function.py:12-22,65-66)."""
    mel_t, gate_t = y[0], y[1]
    mel_o, post_o, gate_o = y_pred[0], y_pred[1], y_pred[2]
    mel_loss = F.mse_loss(mel_o, mel_t) + F.mse_loss(post_o, mel_t)
    gate_loss = F.binary_cross_entropy_with_logits(gate_o.reshape(-1, 1), gate_t.reshape(-1, 1))
    return mel_loss + gate_loss, mel_loss, gate_loss


def loss_align(y_pred, y, x, alignloss: str, iters: int = 0):
    """Tacotron2Loss.forward with the alignment-guide branches (loss_function.py:12-66).  Quirks restated, not fixed:
    L2 compares both alignments with the phone-level target (:29-31); KL (:32-54) replaces exact zeros by 1e-6,
    takes mel_len from x[4] = (max_input_len, max_output_len) indexed by batch item, and applies both slices
    `[:mel_len-1][:text_len-1]` to the frame axis."""
    total, mel_loss, gate_loss = loss(y_pred, y)
    al = alb = None
    if alignloss == "L2" and iters < 40000:
        al = F.mse_loss(y_pred[3], y[2])
        alb = F.mse_loss(y_pred[4], y[2])
    elif alignloss == "KL" and iters < 40000:
        fix = lambda t: torch.where(t == 0, torch.full_like(t, 0.000001), t)
        ao, abo, at = fix(y_pred[3]), fix(y_pred[4]), fix(y[2])
        al = alb = 0
        for b in range(at.size(0)):
            n = int(x[4][b]) - 1
            m = int(x[1][b]) - 1
            o, ob, t = ao[b][:n][:m], abo[b][:n][:m], at[b][:n][:m]
            al = al + (t * (t.log() - o.log())).sum(-1).mean()
            alb = alb + (t * (t.log() - ob.log())).sum(-1).mean()
    if al is not None:
        total = total + al + alb
    return total, mel_loss, gate_loss, al, alb


def forward_single(P, hp, x, training: bool = False, rnd=None, new_stats=None):
    """Classic single-stream Tacotron2.forward (the API of GTA.py:57-59): x = (text, text_lengths, mels,
    max_len, output_lengths) -> [mel, mel_postnet, gate, align].  No reference counterpart."""
    text, tl, mels, _, ol = x
    emb = F.embedding(text, P["embedding.weight"]).transpose(1, 2)
    mem = encoder(emb, tl, P, "encoder", training, _get(rnd, "enc_keep"), new_stats)
    mel, gate, al, _ = decoder_forward(mem, None, mels, tl, None, P, hp, rnd)
    post = mel + postnet(mel, P, training, _get(rnd, "post_keep"), new_stats)
    if hp["mask_padding"] and ol is not None:
        m = ~get_mask_from_lengths(ol, mel.shape[2])
        mel.data.masked_fill_(m[:, None, :].expand_as(mel), 0.0)
        post.data.masked_fill_(m[:, None, :].expand_as(post), 0.0)
        gate.data.masked_fill_(m, 1e3)
    return [mel, post, gate, al]


def inference_single(P, hp, ids, **kw):
    emb = F.embedding(ids, P["embedding.weight"]).transpose(1, 2)
    mem = encoder(emb, None, P, "encoder", False)
    mel, gate, al, _, flag = decoder_inference(mem, None, P, hp, **kw)
    post = mel + postnet(mel, P, False)
    return [mel, post, gate, al, flag]
