"""Pins the CPU oracle (oracle/tacotron2_oracle.py) against golden vectors recorded from
the reference's own model.py (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import tacotron2_oracle as O

SMA, LSA, FA2, GMM, DCA = "StepwiseMonotonicAttention", "LSA", "ForwardAttentionV2", "GMMAttention", "DynamicConvolutionAttention"
TOL = 2e-5      # oracle-vs-reference fp32 CPU, same torch kernels, different op grouping


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _hp(att):
    hp = O.default_hparams()
    hp["attention"] = att
    return hp


def _maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


@pytest.mark.parametrize("att,name", [(SMA, "sma_small_eval"), (LSA, "lsa_small_eval"), (SMA, "sma_baseline_eval"), (FA2, "fa2_small_eval"), (GMM, "gmm_small_eval"), (DCA, "dca_small_eval")])
def test_forward_eval(golden_dir, att, name):
    g = _load(golden_dir, name)
    B, Tin, Tsub, T, _ = g["meta"]
    hp = _hp(att)
    P = recipe.make_weights(hp)
    x, y = recipe.parse_batch(recipe.make_batch(hp, int(B), int(Tin), int(Tsub), int(T)))
    trace = []
    with torch.no_grad():
        out = O.forward(P, hp, x, training=False, trace=trace)
    for k, v in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), out):
        assert _maxabs(v.numpy(), g[k]) < TOL, k
    for key in g.files:
        if key.startswith("step"):
            step, field = key.split("_", 1)
            idx = -1 if step == "steplast" else int(step[4:])
            assert _maxabs(trace[idx][field].numpy(), g[key]) < TOL, key


@pytest.mark.parametrize("att,name", [(SMA, "sma_small_train"), (LSA, "lsa_small_train"), (FA2, "fa2_small_train"), (GMM, "gmm_small_train"), (DCA, "dca_small_train")])
def test_forward_backward_train(golden_dir, att, name):
    g = _load(golden_dir, name)
    B, Tin, Tsub, T, _ = (int(v) for v in g["meta"])
    hp = _hp(att)
    P = recipe.make_weights(hp)
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T))
    rnd = recipe.make_rnd(hp, B, Tin, Tsub, T)
    if att != SMA:
        rnd["sma_noise"] = rnd["sma_noise_bert"] = None
    stats = {}
    out = O.forward(P, hp, x, training=True, rnd=rnd, new_stats=stats)
    for k, v in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), out):
        assert _maxabs(v.detach().numpy(), g[k]) < TOL, k
    loss, mel_loss, gate_loss = O.loss(out, y)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    loss.backward()
    keys = [str(k) for k in g["grad_keys"]]
    for k, st in zip(keys, g["grad_stats"]):
        gr = P[k].grad
        assert gr is not None, k
        assert abs(float(gr.double().norm()) - st[0]) <= 2e-4 * max(st[0], 1e-3), (k, float(gr.double().norm()), st[0])
        if "grad/" + k in g.files:
            ref = g["grad/" + k]
            assert _maxabs(gr.numpy(), ref) <= 1e-5 + 2e-4 * float(np.abs(ref).max()), k
        else:
            ref = g["gradhead/" + k]
            assert _maxabs(gr.reshape(-1)[:256].numpy(), ref) <= 1e-5 + 2e-4 * st[2], k
    # parameters without a gradient in the reference (dead decoder_rnn_bert, model.py:197-199)
    for k, v in P.items():
        if k.startswith("decoder.decoder_rnn_bert"):
            assert v.grad is None
    for key in g.files:
        if key.startswith("bn/"):
            k = key[3:]
            assert _maxabs(stats[k].numpy(), g[key]) < 1e-5, key


@pytest.mark.parametrize("att,name", [(SMA, "sma_infer"), (LSA, "lsa_infer")])
def test_inference(golden_dir, att, name):
    g = _load(golden_dir, name)
    _, Tin, Tsub, steps = (int(v) for v in g["meta"])
    hp = _hp(att)
    P = recipe.make_weights(hp)
    b = recipe.make_batch(hp, 1, Tin, Tsub, 8, seed=4321, ragged=False)
    ids, sub, pcls, bcls = b[0], b[6], b[7], b[8]
    with torch.no_grad():
        r = O.inference(P, hp, ids, sub, pcls, bcls, max_decoder_steps=steps, gate_threshold=2.0)
        assert r[5] is False
        for k, v in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), r[:5]):
            assert v.shape == g["fixed_" + k].shape, k
            assert _maxabs(v.numpy(), g["fixed_" + k]) < TOL, k
        r2 = O.inference(P, hp, ids, sub, pcls, bcls, gate_threshold=float(g["stop_threshold"]))
        assert r2[5] is True
        assert r2[0].shape[2] - 1 == int(g["stop_index"])          # stop frame bit-exact
        assert _maxabs(r2[0].numpy(), g["stop_mel"]) < TOL
