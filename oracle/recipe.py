"""Deterministic weight / input recipes shared by the golden-vector generator, the oracle
tests and the GPU parity tests.  TEST INFRASTRUCTURE ONLY (see tacotron2_oracle.py).

No trained checkpoint of the reference exists, and a default-size
state_dict is 250 MB, so golden fixtures store *outputs only*; the weights are rebuilt
from this recipe (numpy PCG64 streams keyed by the state_dict key -> stable across
torch versions and machines).  ``state_dict_spec`` restates the reference's key/shape
contract (model.py:494-515, attention.py:25-37,291-322, layers.py:8-39);
``tests/golden/make_golden.py`` asserts it equals the reference module's own
``state_dict()`` before any vector is generated.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch

from .tacotron2_oracle import default_hparams


def state_dict_spec(hp: dict) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) in the reference's state_dict order.  kind selects the init."""
    E, S = hp["encoder_embedding_dim"], hp["symbols_embedding_dim"]
    K = hp["encoder_kernel_size"]
    A, Ha, Hd, Pn = hp["attention_dim"], hp["attention_rnn_dim"], hp["decoder_rnn_dim"], hp["prenet_dim"]
    M = hp["n_mel_channels"] * hp["n_frames_per_step"]
    C = hp["BERT_embedding_dim"]
    PE, PK = hp["postnet_embedding_dim"], hp["postnet_kernel_size"]
    spec: List[Tuple[str, Tuple[int, ...], str]] = []
    add = lambda k, s, kind: spec.append((k, tuple(s), kind))

    def bn(prefix, c):
        add(prefix + ".weight", (c,), "bn_w"); add(prefix + ".bias", (c,), "bn_b")
        add(prefix + ".running_mean", (c,), "bn_rm"); add(prefix + ".running_var", (c,), "bn_rv")
        add(prefix + ".num_batches_tracked", (), "bn_n")

    def enc(prefix):
        for i in range(hp["encoder_n_convolutions"]):
            add(f"{prefix}.convolutions.{i}.0.conv.weight", (E, E, K), "conv")
            add(f"{prefix}.convolutions.{i}.0.conv.bias", (E,), "bias")
            bn(f"{prefix}.convolutions.{i}.1", E)
        for sfx in ("", "_reverse"):
            add(f"{prefix}.lstm.weight_ih_l0{sfx}", (4 * (E // 2), E), "lstm")
            add(f"{prefix}.lstm.weight_hh_l0{sfx}", (4 * (E // 2), E // 2), "lstm")
            add(f"{prefix}.lstm.bias_ih_l0{sfx}", (4 * (E // 2),), "lstm")
            add(f"{prefix}.lstm.bias_hh_l0{sfx}", (4 * (E // 2),), "lstm")

    def lstmcell(prefix, i, h):
        add(prefix + ".weight_ih", (4 * h, i), "lstm"); add(prefix + ".weight_hh", (4 * h, h), "lstm")
        add(prefix + ".bias_ih", (4 * h,), "lstm"); add(prefix + ".bias_hh", (4 * h,), "lstm")

    def attention(prefix):
        if hp["attention"] == "StepwiseMonotonicAttention":          # attention.py:302-322
            add(prefix + ".memory_layer.linear_layer.weight", (A, E), "linear")
            add(prefix + ".v.weight", (1, A), "linear")
            add(prefix + ".query_layer.linear_layer.weight", (A, Ha), "linear")
        elif hp["attention"] == "DynamicConvolutionAttention":      # attention.py:199-230 (the P buffer sits between memory_layer and W)
            add(prefix + ".P", (11,), "dca_prior")
            add(prefix + ".memory_layer.linear_layer.weight", (A, E), "linear")
            add(prefix + ".W.weight", (A, Ha), "linear"); add(prefix + ".W.bias", (A,), "bias")
            add(prefix + ".V.weight", (168, A), "linear")
            add(prefix + ".F.weight", (8, 1, 21), "conv")
            add(prefix + ".U.weight", (A, 8), "linear")
            add(prefix + ".T.weight", (A, 8), "linear"); add(prefix + ".T.bias", (A,), "bias")
            add(prefix + ".v.weight", (1, A), "linear")
        elif hp["attention"] == "GMMAttention":                      # attention.py:405-415
            add(prefix + ".memory_layer.linear_layer.weight", (A, E), "linear")
            add(prefix + ".mlp.0.weight", (A, Ha), "linear"); add(prefix + ".mlp.0.bias", (A,), "bias")
            add(prefix + ".mlp.2.weight", (15, A), "linear"); add(prefix + ".mlp.2.bias", (15,), "bias")
        else:                                                        # LSA, attention.py:26-37
            add(prefix + ".query_layer.linear_layer.weight", (A, Ha), "linear")
            add(prefix + ".memory_layer.linear_layer.weight", (A, E), "linear")
            add(prefix + ".v.linear_layer.weight", (1, A), "linear")
            nf, ks = hp["attention_location_n_filters"], hp["attention_location_kernel_size"]
            add(prefix + ".location_layer.location_conv.conv.weight", (nf, 2, ks), "conv")
            add(prefix + ".location_layer.location_dense.linear_layer.weight", (A, nf), "linear")

    add("embedding.weight", (hp["n_symbols"], S), "emb")
    add("embedding_sub.weight", (hp["sub_n_symbols"], S), "emb")
    enc("encoder"); enc("encoder_sub")
    add("linear_converter.linear_layer.weight", (E, E + C), "linear"); add("linear_converter.linear_layer.bias", (E,), "bias")
    add("linear_converter_sub.linear_layer.weight", (E, E + C), "linear"); add("linear_converter_sub.linear_layer.bias", (E,), "bias")
    for pn in ("prenet", "prenet_bert"):
        add(f"decoder.{pn}.layers.0.linear_layer.weight", (Pn, M), "linear")
        add(f"decoder.{pn}.layers.1.linear_layer.weight", (Pn, Pn), "linear")
    lstmcell("decoder.attention_rnn", Pn + E, Ha)
    lstmcell("decoder.attention_rnn_bert", Pn + E, Ha)
    attention("decoder.attention_layer")
    attention("decoder.attention_layer_bert")
    lstmcell("decoder.decoder_rnn", 2 * Ha + 2 * E, Hd)
    lstmcell("decoder.decoder_rnn_bert", Ha + E, Hd)                 # dead module, model.py:197-199
    add("decoder.linear_projection.linear_layer.weight", (M, Hd + 2 * E), "linear")
    add("decoder.linear_projection.linear_layer.bias", (M,), "bias")
    add("decoder.gate_layer.linear_layer.weight", (1, Hd + 2 * E), "linear")
    add("decoder.gate_layer.linear_layer.bias", (1,), "bias")
    n = hp["postnet_n_convolutions"]
    for i in range(n):
        ci = hp["n_mel_channels"] if i == 0 else PE
        co = hp["n_mel_channels"] if i == n - 1 else PE
        add(f"postnet.convolutions.{i}.0.conv.weight", (co, ci, PK), "conv")
        add(f"postnet.convolutions.{i}.0.conv.bias", (co,), "bias")
        bn(f"postnet.convolutions.{i}.1", co)
    return spec


def state_dict_spec_single(hp: dict) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Key/shape list of the classic single-stream Tacotron2 (NVIDIA layout): the dual-stream list minus
    the sub-word stream and the CLS converters, with decoder_rnn / projections on [h|ctx]."""
    E, Ha, Hd = hp["encoder_embedding_dim"], hp["attention_rnn_dim"], hp["decoder_rnn_dim"]
    M = hp["n_mel_channels"] * hp["n_frames_per_step"]
    out = []
    for k, shape, kind in state_dict_spec(hp):
        if "_bert" in k or k.startswith(("embedding_sub", "encoder_sub", "linear_converter")):
            continue
        if k == "decoder.decoder_rnn.weight_ih":
            shape = (4 * Hd, Ha + E)
        if k in ("decoder.linear_projection.linear_layer.weight", "decoder.gate_layer.linear_layer.weight"):
            shape = (shape[0], Hd + E)
        out.append((k, shape, kind))
    return out


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([zlib.crc32(key.encode()), seed]))


def make_weights(hp: dict, seed: int = 1234, single: bool = False) -> Dict[str, torch.Tensor]:
    """Recipe weights: Xavier-like uniform ranges per kind, non-trivial BN statistics."""
    out: Dict[str, torch.Tensor] = {}
    for key, shape, kind in (state_dict_spec_single(hp) if single else state_dict_spec(hp)):
        g = _rng(key, seed)
        if kind == "bn_n":
            out[key] = torch.tensor(int(g.integers(1, 5)), dtype=torch.long)
            continue
        if kind in ("linear", "conv"):
            rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
            bound = math.sqrt(6.0 / ((shape[0] + shape[1]) * rf))
            a = g.uniform(-bound, bound, size=shape)
        elif kind == "lstm":
            bound = 1.0 / math.sqrt(shape[0] // 4)
            a = g.uniform(-bound, bound, size=shape)
        elif kind == "emb":
            bound = math.sqrt(3.0) * math.sqrt(2.0 / (hp["n_symbols"] + hp["symbols_embedding_dim"]))
            a = g.uniform(-bound, bound, size=shape)
        elif kind == "dca_prior":                     # attention.py:219-221: beta-binomial prior, flipped
            from scipy.stats import betabinom
            a = betabinom.pmf(np.arange(shape[0]), shape[0] - 1, 0.1, 0.9)[::-1].copy()
        elif kind == "bias":
            a = g.uniform(-0.05, 0.05, size=shape)
        elif kind == "bn_w":
            a = g.uniform(0.5, 1.5, size=shape)
        elif kind == "bn_b":
            a = g.uniform(-0.1, 0.1, size=shape)
        elif kind == "bn_rm":
            a = g.uniform(-0.1, 0.1, size=shape)
        elif kind == "bn_rv":
            a = g.uniform(0.5, 1.5, size=shape)
        else:
            raise KeyError(kind)
        out[key] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    return out


def make_batch(hp: dict, B: int, Tin: int, Tsub: int, T: int, seed: int = 1234, ragged: bool = True):
    """Synthetic batch in the layout ``parse_batch`` consumes:
    (text, input_lengths, input_lengths_bert, mel [B,80,T], gate [B,T], output_lengths,
     sub_ids, phoneme_cls [B,Tin,768], bert_cls [B,Tsub,768], align)."""
    g = np.random.Generator(np.random.PCG64([seed, B, Tin, Tsub, T]))
    lo = lambda n: max(1, int(0.7 * n))
    def lens(n):
        l = g.integers(lo(n), n + 1, size=B) if ragged else np.full(B, n)
        l[0] = n
        return np.sort(l)[::-1].copy()
    tl, bl, ol = lens(Tin), lens(Tsub), lens(T)
    text = g.integers(1, hp["n_symbols"], size=(B, Tin)); sub = g.integers(1, hp["sub_n_symbols"], size=(B, Tsub))
    mel = np.clip(g.normal(-5.0, 2.0, size=(B, hp["n_mel_channels"], T)), -11.5, 2.0).astype(np.float32)
    gate = np.zeros((B, T), np.float32)
    for b in range(B):
        text[b, tl[b]:] = 0; sub[b, bl[b]:] = 0; mel[b, :, ol[b]:] = 0.0; gate[b, ol[b] - 1:] = 1.0
    C = hp["BERT_embedding_dim"]
    cls = g.normal(0, 1, size=(B, 1, C)).astype(np.float32)
    bcls = g.normal(0, 1, size=(B, 1, C)).astype(np.float32)
    t = torch.from_numpy
    return (t(text).long(), t(tl).long(), t(bl).long(), t(mel), t(gate), t(ol).long(), t(sub).long(),
            t(np.repeat(cls, Tin, 1).copy()), t(np.repeat(bcls, Tsub, 1).copy()), torch.zeros(B, T, Tin))


def parse_batch(batch):
    """BERT_Tacotron2.parse_batch (model.py:517-529), device-agnostic restatement."""
    text, il, ilb, mel, gate, ol, sub, pcls, bcls, align = batch
    max_in = int(torch.max(torch.cat((il, ilb), 0)).item())
    max_out = int(torch.max(ol).item())
    return ((text.long(), il.long(), ilb.long(), mel.float(), (max_in, max_out), ol.long(), sub, pcls, bcls),
            (mel.float(), gate.float(), align.float()))


def make_rnd(hp: dict, B: int, Tin: int, Tsub: int, T: int, seed: int = 7) -> dict:
    """Keep-masks / noise for one training-mode forward, in the reference's tensor layouts."""
    g = np.random.Generator(np.random.PCG64([seed, 99]))
    E, Ha, Hd, Pn, PE, M = (hp["encoder_embedding_dim"], hp["attention_rnn_dim"], hp["decoder_rnn_dim"],
                            hp["prenet_dim"], hp["postnet_embedding_dim"], hp["n_mel_channels"])
    keep = lambda p, *s: torch.from_numpy((g.random(size=s) >= p).astype(np.float32))
    nrm = lambda *s: torch.from_numpy(g.normal(size=s).astype(np.float32))
    n = hp["postnet_n_convolutions"]
    return dict(
        enc_keep=[keep(0.5, B, E, Tin) for _ in range(3)],
        encsub_keep=[keep(0.5, B, E, Tsub) for _ in range(3)],
        prenet_keep=[keep(0.5, T, B, Pn) for _ in range(2)],
        prenet_bert_keep=[keep(0.5, T, B, Pn) for _ in range(2)],
        att_h_keep=keep(0.1, T, B, Ha), att_c_keep=keep(0.1, T, B, Ha),
        att_h_bert_keep=keep(0.1, T, B, Ha), att_c_bert_keep=keep(0.1, T, B, Ha),
        dec_h_keep=keep(0.1, T, B, Hd), dec_c_keep=keep(0.1, T, B, Hd),
        post_keep=[keep(0.5, B, M if i == n - 1 else PE, T) for i in range(n)],
        sma_noise=nrm(T, B, Tin), sma_noise_bert=nrm(T, B, Tsub),
    )


__all__ = ["default_hparams", "state_dict_spec", "state_dict_spec_single", "make_weights", "make_batch", "parse_batch", "make_rnd"]
