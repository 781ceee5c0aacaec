"""BASELINE.json configs at their own sizes (default 512-dim / 80-mel dims), checked against the CPU oracle on the items
the oracle can do in seconds and through size-independent properties on the rest.

  configs[3]  inference(), B=32, >= 256 decoder steps        model.py:430-492 (oracle: B=1 runs)
  configs[4]  teacher-forced forward (GTA), B=128, T=400     model.py:392-428
  configs[1]  whole BERT_Tacotron2 training step, B=64, T=400, bf16 operands

Tolerances: fp32 mode 1e-4 max-abs (north_star) and bit-exact stop index; bf16 mode against the fp32 HIP path within
the documented bound (mel 0.03, gate 0.014, alignments 0.002 on the golden case; random inputs here)."""
import pytest
import torch

from helpers import SMA, hp_for, maxabs, to_dev
from oracle import recipe
from oracle import tacotron2_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def env():
    from tacotron2_subword_amd import _lib as L, ops
    L.lib()
    L.set_precision("f32")
    yield L, ops
    L.set_precision("f32")


def _memories(hp, P, B, Tin, Tsub, seed):
    b = recipe.make_batch(hp, B, Tin, Tsub, 8, seed=seed, ragged=False)
    with torch.no_grad():
        mem = O.front_end(P, hp, b[0], None, b[7], "phone", False)
        mem_sub = O.front_end(P, hp, b[6], None, b[8], "sub", False)
    return mem, mem_sub


def _sma_alignment_properties(al, tol=2e-4):
    """[B, T, Tin] StepwiseMonotonicAttention alignments (attention.py:337-348): non-negative, mass never created (a row sums to
    at most 1), and the mass only moves forward: the cumulative distribution at every position never grows from one frame to the next."""
    assert bool((al >= -tol).all())
    assert float(al.sum(2).max()) <= 1.0 + tol
    cdf = al.cumsum(2)
    assert float((cdf[:, 1:] - cdf[:, :-1]).max()) <= tol


def test_config4_decode_b32_default_dims(env):
    """BASELINE configs[3] at its own size: B=32, max_decoder_steps=1000, default dims.  fp32: items 0 and 17 equal their own B=1
    oracle runs over the first 256 frames (< 1e-4; the oracle takes seconds for that many), all 1000 frames are finite and keep
    the attention's invariants, and every item's stop index is the bit-exact first frame over a mid-sequence threshold; bf16
    (the persistent decode loop): the persistent kernel really ran, the first 256 frames stay within the bf16-mode bound of
    the fp32 HIP run, all 1000 keep the invariants."""
    L, ops = env
    hp = hp_for(SMA)
    P = recipe.make_weights(hp)
    P["decoder.gate_layer.linear_layer.bias"] = P["decoder.gate_layer.linear_layer.bias"] - 1.0
    B, Tin, Tsub, steps, n_or = 32, 100, 60, 1000, 256
    mem, mem_sub = _memories(hp, P, B, Tin, Tsub, seed=2024)
    dims = L.dims_from_hparams(hp)
    Pd = to_dev(P)
    W = L.decoder_weights(Pd, dims.attention_kind)
    L.set_precision("f32")
    dp, n, stop = ops.decoder_infer(W, dims, mem.cuda(), mem_sub.cuda(), max_steps=steps, gate_threshold=2.0, prenet_dropout=False)
    torch.cuda.synchronize()
    assert n == steps and bool((stop < 0).all())
    assert bool(torch.isfinite(dp.mel).all()) and bool(torch.isfinite(dp.gate).all())
    _sma_alignment_properties(dp.align); _sma_alignment_properties(dp.align_sub)
    gate32 = torch.sigmoid(dp.gate.cpu())
    for i in (0, 17):
        with torch.no_grad():
            mel, gate, al, alb, flag = O.decoder_inference(mem[i:i + 1], mem_sub[i:i + 1], P, hp, max_decoder_steps=n_or, gate_threshold=2.0)
        assert maxabs(dp.mel[i:i + 1, :n_or].cpu().transpose(1, 2), mel) < TOL
        assert maxabs(dp.gate[i:i + 1, :n_or].cpu().unsqueeze(-1), gate) < TOL
        assert maxabs(dp.align[i:i + 1, :n_or].cpu(), al) < TOL and maxabs(dp.align_sub[i:i + 1, :n_or].cpu(), alb) < TOL
    # a threshold that the median item crosses early: the stop rule must fire on exactly the first frame above it
    thr = float(gate32[:, n_or // 2].median())
    dp2, n2, stop2 = ops.decoder_infer(W, dims, mem.cuda(), mem_sub.cuda(), max_steps=steps, gate_threshold=thr, prenet_dropout=False, poll_every=8)
    torch.cuda.synchronize()
    above = gate32 > thr
    want = torch.where(above.any(1), above.float().argmax(1), torch.full((B,), -1))
    got = stop2.cpu().long()
    fired = want >= 0
    assert bool((got[fired][want[fired] < n2] == want[fired][want[fired] < n2]).all()), (got, want)      # (the loop may end before a late item's frame)
    assert int(fired.sum()) >= B // 4                     # the threshold is a mid-sequence one for a good part of the batch
    for i in (0, 17):                                     # and the B=1 oracle with the stop rule on stops on the same frame
        if 0 <= want[i] < n_or:
            with torch.no_grad():
                mel, *_ = O.decoder_inference(mem[i:i + 1], mem_sub[i:i + 1], P, hp, max_decoder_steps=n_or, gate_threshold=thr)
            assert mel.shape[2] == int(want[i]) + 1
    # bf16-operand decode loop (persistent launches of 32 steps) at this size, against the fp32 HIP run
    L.set_precision("bf16")
    try:
        L.prof_enable(256)
        dpb, nb, stopb = ops.decoder_infer(W, dims, mem.cuda(), mem_sub.cuda(), max_steps=steps, gate_threshold=2.0, prenet_dropout=False)
        torch.cuda.synchronize()
        prof = L.prof_collect()
    finally:
        L.set_precision("f32")
    assert prof["chain_dec"][1] == (steps + 31) // 32 and not any(dpb.chain_status()), prof        # the persistent kernel really ran
    assert nb == steps and bool(torch.isfinite(dpb.mel).all())
    _sma_alignment_properties(dpb.align, 2e-3); _sma_alignment_properties(dpb.align_sub, 2e-3)
    f = slice(0, n_or)
    errs = dict(mel=maxabs(dpb.mel[:, f], dp.mel[:, f]), gate=maxabs(dpb.gate[:, f], dp.gate[:, f]), align=maxabs(dpb.align[:, f], dp.align[:, f]),
                align_sub=maxabs(dpb.align_sub[:, f], dp.align_sub[:, f]))
    print("config #4, bf16 vs fp32 HIP (first 256 frames):", errs)
    assert errs["mel"] < 0.05 and errs["gate"] < 0.025 and errs["align"] < 0.005 and errs["align_sub"] < 0.005, errs    # observed 0.022 / 0.010 / 0.0019 / 0.0015


def _tf_inputs(hp, P, B, Tin, Tsub, T, seed):
    batch = recipe.make_batch(hp, B, Tin, Tsub, T, seed=seed)
    x, y = recipe.parse_batch(batch)
    with torch.no_grad():
        mem = O.front_end(P, hp, x[0], x[1], x[7], "phone", False)
        mem_sub = O.front_end(P, hp, x[6], x[2], x[8], "sub", False)
    return x, mem, mem_sub


def test_config5_gta_b128_default_dims(env):
    """Teacher-forced forward at B=128, T=400 (four 32-row tiles per workgroup in the recurrent steps).  fp32: two items
    equal the oracle's forward of those two items (the decoder has no cross-item arithmetic), every item equals itself
    run alone; bf16 (the mode bench.py --workload gta runs, persistent chains): within the bf16-mode bound of fp32."""
    L, ops = env
    hp = hp_for(SMA)
    P = recipe.make_weights(hp)
    B, Tin, Tsub, T = 128, 100, 60, 400
    x, mem, mem_sub = _tf_inputs(hp, P, B, Tin, Tsub, T, seed=77)
    dims = L.dims_from_hparams(hp)
    Pd = to_dev(P)                                       # (the packed pointers do not own the tensors)
    W = L.decoder_weights(Pd, dims.attention_kind)
    args = [t.cuda() for t in (mem, mem_sub, x[1], x[2], x[3])]
    L.set_precision("f32")
    dp = ops.decoder_forward(W, dims, *args, training=False, prenet_dropout=False, seed=0)
    torch.cuda.synchronize()
    idx = [3, 101]
    with torch.no_grad():
        mel, gate, al, alb = O.decoder_forward(mem[idx], mem_sub[idx], x[3][idx], x[1][idx], x[2][idx], P, hp)
    # (the oracle pads the pair to ITS longest memory; compare the common columns)
    assert maxabs(dp.mel[idx].cpu().transpose(1, 2), mel) < TOL
    assert maxabs(dp.gate[idx].cpu(), gate) < TOL
    assert maxabs(dp.align[idx][:, :, :al.shape[2]].cpu(), al) < TOL and maxabs(dp.align_sub[idx][:, :, :alb.shape[2]].cpu(), alb) < TOL
    one = ops.decoder_forward(W, dims, *[t[64:65].contiguous() for t in args], training=False, prenet_dropout=False, seed=0)
    torch.cuda.synchronize()
    assert maxabs(one.mel, dp.mel[64:65]) < TOL and maxabs(one.align, dp.align[64:65]) < TOL
    L.set_precision("bf16")
    try:
        L.prof_enable(64)
        dpb = ops.decoder_forward(W, dims, *args, training=False, prenet_dropout=False, seed=0)
        torch.cuda.synchronize()
        prof = L.prof_collect()
        assert prof["chain_a_fwd"][1] == 1 and prof["chain_b_fwd"][1] == 1 and not any(dpb.chain_status()), prof     # both persistent chains really ran
    finally:
        L.set_precision("f32")
    errs = dict(mel=maxabs(dpb.mel, dp.mel), gate=maxabs(dpb.gate, dp.gate), align=maxabs(dpb.align, dp.align), align_sub=maxabs(dpb.align_sub, dp.align_sub))
    print("config #5, bf16 (persistent chains) vs fp32 HIP:", errs)
    assert errs["mel"] < 0.06 and errs["gate"] < 0.05 and errs["align"] < 0.005 and errs["align_sub"] < 0.006, errs   # observed 0.029 / 0.025 / 0.0019 / 0.0027
    # the persistent forward against the ORACLE directly (not only against the other HIP path): the two items above
    assert maxabs(dpb.mel[idx].cpu().transpose(1, 2), mel) < 0.06 and maxabs(dpb.gate[idx].cpu(), gate) < 0.05
    assert maxabs(dpb.align[idx][:, :, :al.shape[2]].cpu(), al) < 0.005


def test_config2_training_step_b64_bf16(env):
    """One whole training step of BERT_Tacotron2 at B=64, T=400 in the bf16 mode (persistent forward and backward chains),
    next to the same step in the fp32 mode: finite everywhere, the loss within the documented bound, per-parameter
    gradient norms tracking fp32.  Then the decoder alone at this length in fp32: two items' memory gradients and a sample
    of parameter gradients against fp64 autograd through the oracle (T=400, both passes at full length)."""
    L, ops = env
    from tacotron2_subword_amd.hparams import create_hparams
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    from tacotron2_subword_amd.model import BERT_Tacotron2
    from tacotron2_subword_amd import train as T
    hp = create_hparams()
    B, Tin, Tsub, Tn = 64, 100, 60, 400
    res = {}
    for mode in ("f32", "bf16"):
        L.set_precision(mode)
        try:
            torch.manual_seed(1234)
            model = BERT_Tacotron2(hp).cuda().train()
            x, y = model.parse_batch(T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=31))
            L.prof_enable(8 * Tn + 64)
            out = model(x)
            loss = Tacotron2Loss()(out, y, x)[0]
            loss.backward()
            torch.cuda.synchronize()
            prof = L.prof_collect()
            ran = {k: prof[k][1] for k in ("chain_a_fwd", "chain_b_fwd", "chain_a_bwd", "chain_b_bwd")}
            assert all(v == (1 if mode == "bf16" else 0) for v in ran.values()), ran       # bf16: all four persistent chains really ran
            assert all(bool(torch.isfinite(o).all()) for o in out)
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
            assert all(bool(torch.isfinite(g).all()) for g in grads.values())
            res[mode] = (float(loss), grads)
        finally:
            L.set_precision("f32")
    l32, g32 = res["f32"]
    l16, g16 = res["bf16"]
    print("config #2 loss fp32 / bf16:", l32, l16)
    assert abs(l16 - l32) < 2e-3 * abs(l32)                 # observed 3e-6
    # (a conv bias in front of a training-mode BatchNorm has a mathematically zero gradient: what is left there is rounding)
    worst = {k: float((g16[k] - g32[k]).norm()) / (float(g32[k].norm()) + 1e-12) for k in g32 if not k.endswith(".0.conv.bias")}
    print("worst relative gradient deviations bf16 vs fp32:", dict(sorted(worst.items(), key=lambda kv: -kv[1])[:5]))
    assert set(g16) == set(g32) and max(worst.values()) < 0.16, worst      # observed 0.082 (sub-word encoder's first conv: bf16 operands)
    # decoder alone, T=400, fp32, two items: gradients vs fp64 autograd through the oracle
    hpo = hp_for(SMA)
    P = recipe.make_weights(hpo)
    xo, mem, mem_sub = _tf_inputs(hpo, P, 2, Tin, Tsub, Tn, seed=5)
    g = torch.Generator().manual_seed(9)
    dmel, dgate = torch.randn(2, 80, Tn, generator=g), torch.randn(2, Tn, generator=g)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    keys = ["decoder.attention_rnn.weight_hh", "decoder.decoder_rnn.weight_ih", "decoder.attention_layer.query_layer.linear_layer.weight",
            "decoder.linear_projection.linear_layer.weight", "decoder.prenet.layers.0.linear_layer.weight"]
    for k in keys:
        P64[k].requires_grad_(True)
    m64, ms64 = mem.double().requires_grad_(True), mem_sub.double().requires_grad_(True)
    mel, gate, al, alb = O.decoder_forward(m64, ms64, xo[3].double(), xo[1], xo[2], P64, hpo)
    ((mel * dmel.double()).sum() + (gate * dgate.double()).sum()).backward()
    dims = L.dims_from_hparams(hpo)
    Pd = to_dev(P)
    W = L.decoder_weights(Pd, dims.attention_kind)
    args = [t.cuda() for t in (mem, mem_sub, xo[1], xo[2], xo[3])]
    dp = ops.decoder_forward(W, dims, *args, training=False, prenet_dropout=False, seed=0)
    G, dm, dms = ops.decoder_backward(W, Pd, dims, dp, args[0], args[1], dmel.transpose(1, 2).contiguous().cuda(), dgate.cuda(),
                                      training=False, prenet_dropout=False, seed=0)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double().cpu() - b).norm()) / (float(b.norm()) + 1e-30)
    errs = {"d_memory": rel(dm, m64.grad), "d_memory_sub": rel(dms, ms64.grad)}
    errs.update({k: rel(G[k], P64[k].grad) for k in keys})
    print("decoder backward at T=400 vs fp64 oracle autograd (relative L2):", errs)
    assert max(errs.values()) < 2e-3, errs


def test_lsa_training_step_b64_bf16(env):
    """The same whole-step comparison for LocationSensitiveAttention (the other attention SURVEY section 8 names) at B=64 with
    the BASELINE memory lengths and 160 frames: bf16 mode = persistent forward chain + the persistent LSA backward (two
    position splits per item, halo rows and softmax-dot hand-off) against the fp32 mode's per-step launches."""
    L, ops = env
    from tacotron2_subword_amd.hparams import create_hparams
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    from tacotron2_subword_amd.model import BERT_Tacotron2
    from tacotron2_subword_amd import train as T
    hp = create_hparams()
    hp.attention = "LocationSensitiveAttention"
    B, Tin, Tsub, Tn = 64, 100, 60, 160
    res = {}
    for mode in ("f32", "bf16"):
        L.set_precision(mode)
        try:
            torch.manual_seed(1234)
            model = BERT_Tacotron2(hp).cuda().train()
            x, y = model.parse_batch(T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=31))
            L.prof_enable(8 * Tn + 64)
            out = model(x)
            loss = Tacotron2Loss()(out, y, x)[0]
            loss.backward()
            torch.cuda.synchronize()
            prof = L.prof_collect()
            assert prof["chain_a_bwd"][1] == (1 if mode == "bf16" else 0), prof      # the persistent LSA backward really ran
            assert all(bool(torch.isfinite(o).all()) for o in out)
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
            assert all(bool(torch.isfinite(g).all()) for g in grads.values())
            res[mode] = (float(loss.detach()), grads)
        finally:
            L.set_precision("f32")
    l32, g32 = res["f32"]
    l16, g16 = res["bf16"]
    print("LSA B=64 loss fp32 / bf16:", l32, l16)
    assert abs(l16 - l32) < 2e-3 * abs(l32)                 # observed 3e-5
    worst = {k: float((g16[k] - g32[k]).norm()) / (float(g32[k].norm()) + 1e-12) for k in g32 if not k.endswith(".0.conv.bias")}
    print("worst relative gradient deviations bf16 vs fp32 (LSA):", dict(sorted(worst.items(), key=lambda kv: -kv[1])[:5]))
    assert set(g16) == set(g32) and max(worst.values()) < 0.14, worst      # observed 0.068
