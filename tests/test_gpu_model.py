"""GPU parity of the drop-in module API (BERT_Tacotron2.parse_batch / forward / inference,
load_model) against the golden vectors recorded from the reference and the oracle's autograd."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import tacotron2_oracle as O

from helpers import DCA, FA2, GMM, LSA, SMA, hp_for, load_golden, maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build_model(att, train=False):
    from tacotron2_subword_amd.hparams import create_hparams
    from tacotron2_subword_amd.model import BERT_Tacotron2
    hps = create_hparams()
    hps.attention = att
    m = BERT_Tacotron2(hps)
    m.load_state_dict(recipe.make_weights(hp_for(att)))       # reference-keyed state_dict loads unchanged
    m = m.cuda()
    m.train(train)
    m.decoder.prenet_dropout = False                            # deterministic parity set
    return m, hps


@pytest.mark.parametrize("att,name", [(SMA, "sma_small_eval"), (LSA, "lsa_small_eval"), (SMA, "sma_baseline_eval"), (FA2, "fa2_small_eval"), (GMM, "gmm_small_eval"), (DCA, "dca_small_eval")])
def test_forward_eval_vs_golden(att, name):
    g = load_golden(name)
    B, Tin, Tsub, T, _ = (int(v) for v in g["meta"])
    m, hps = build_model(att)
    batch = recipe.make_batch(hp_for(att), B, Tin, Tsub, T)
    x, y = m.parse_batch(batch)
    with torch.no_grad():
        out = m(x)
    for k, v in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), out):
        assert tuple(v.shape) == g[k].shape, k
        assert maxabs(v, g[k]) < TOL, k


@pytest.mark.parametrize("att", [SMA, LSA, FA2, GMM, DCA])
def test_backward_eval_mode_vs_oracle_autograd(att):
    """Whole-model gradients (embeddings, encoders, converters, decoder, postnet) in eval mode (BN
    running statistics, no dropout, no noise): HIP backward of every block vs the oracle's autograd."""
    hp = hp_for(att)
    B, Tin, Tsub, T = 3, 13, 8, 12
    m, hps = build_model(att, train=False)
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    batch = recipe.make_batch(hp, B, Tin, Tsub, T)
    x, y = m.parse_batch(batch)
    out = m(x)
    loss = Tacotron2Loss()(out, y, x)[0]
    loss.backward()
    # oracle: fp32 autograd for the loss value, fp64 autograd as the gradient ground truth.  A parameter whose
    # gradient is a small remainder of cancelling terms (DCA's T.bias: 1e-3 relative between the oracle's own fp32
    # and fp64 results) is held to 3x the fp32 oracle's error instead of the flat bound.
    def oracle_grads(dt):
        P = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in recipe.make_weights(hp).items()}
        for k, v in P.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        cast = lambda ts: tuple(t.to(dt) if torch.is_tensor(t) and t.is_floating_point() else t for t in ts)
        xo, yo = recipe.parse_batch(batch)
        lo = O.loss(O.forward(P, hp, cast(xo), training=False), cast(yo))[0]
        lo.backward()
        return float(lo.detach()), {k: (None if v.grad is None else v.grad.double()) for k, v in P.items() if v.is_floating_point()}
    lo32, g32 = oracle_grads(torch.float32)
    _, g64 = oracle_grads(torch.float64)
    assert abs(float(loss.detach()) - lo32) < 1e-5
    bad = {}
    for k, p in m.named_parameters():
        ref = g64[k]
        if ref is None:
            assert p.grad is None, k                 # dead decoder_rnn_bert
            continue
        scale = max(float(ref.abs().max()), 1e-7)
        err = float((p.grad.double().cpu() - ref).abs().max()) / scale
        noise = float((g32[k] - ref).abs().max()) / scale
        if not err < max(5e-4, 3 * noise):
            bad[k] = (err, noise)
    assert not bad, bad


@pytest.mark.parametrize("att,name", [(SMA, "sma_infer"), (LSA, "lsa_infer")])
def test_inference_vs_golden(att, name):
    g = load_golden(name)
    _, Tin, Tsub, steps = (int(v) for v in g["meta"])
    m, hps = build_model(att)
    b = recipe.make_batch(hp_for(att), 1, Tin, Tsub, 8, seed=4321, ragged=False)
    ids, sub, pcls, bcls = b[0].cuda(), b[6].cuda(), b[7].cuda(), b[8].cuda()
    m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, steps
    r = m.inference(ids, sub, pcls, bcls)
    assert r[5] is False
    for k, v in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), r[:5]):
        assert tuple(v.shape) == g["fixed_" + k].shape, k
        assert maxabs(v, g["fixed_" + k]) < TOL, k
    m.decoder.gate_threshold, m.decoder.max_decoder_steps = float(g["stop_threshold"]), 1000
    r2 = m.inference(ids, sub, pcls, bcls)
    assert r2[5] is True
    assert r2[0].shape[2] - 1 == int(g["stop_index"])            # stop frame bit-exact
    assert maxabs(r2[0], g["stop_mel"]) < TOL
    assert maxabs(r2[1], g["stop_mel_postnet"]) < TOL


def test_training_step_runs_and_is_finite():
    """One full training iteration at a small shape: finite loss, every live parameter moves,
    the dead decoder_rnn_bert does not."""
    from tacotron2_subword_amd import train as T
    from tacotron2_subword_amd.hparams import create_hparams
    hps = create_hparams()
    model, opt, crit = T.make_training_objects(hps)
    model.train()
    x, y = model.parse_batch(T.synthetic_batch(hps, 4, 20, 12, 24))
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    l0 = float(T.train_step(model, crit, opt, x, y, hps, 0))
    l1 = float(T.train_step(model, crit, opt, x, y, hps, 1))
    assert np.isfinite(l0) and np.isfinite(l1)
    for k, v in model.named_parameters():
        moved = not torch.equal(v.detach(), before[k])
        assert moved == (not k.startswith("decoder.decoder_rnn_bert")), k


@pytest.mark.parametrize("att", [SMA, LSA, FA2, GMM, DCA])
def test_training_mode_full_model_vs_oracle(att):
    """Training mode end to end: BN batch statistics, conv/LSTM-state/prenet dropout and SMA noise drawn
    by the HIP RNG (exported through the C ABI and replayed through the oracle), loss, every parameter
    gradient, and the BN running-statistics update."""
    from tacotron2_subword_amd import _lib as L
    from tacotron2_subword_amd import ops
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    hp = hp_for(att)
    B, Tin, Tsub, T = 3, 13, 8, 12
    m, hps = build_model(att, train=True)
    m.decoder.prenet_dropout = True
    m._t2_calls, m.decoder._t2_calls = 0, 0
    seed = ((1234 * 1000003 + 1) * 64) & ((1 << 63) - 1)
    batch = recipe.make_batch(hp, B, Tin, Tsub, T)
    x, y = m.parse_batch(batch)
    out = m(x)
    loss = Tacotron2Loss()(out, y, x)[0]
    loss.backward()
    # the bits the kernels drew, in the oracle's layouts
    S, E, Pn, Ha, Hd, M = L.SITE, 512, 256, 1024, 1024, 80
    km = lambda site, p, *shape: ops.rng_keep_mask(seed, site, int(np.prod(shape)), p).view(*shape).float().cpu()
    bct = lambda t: t.permute(0, 2, 1).contiguous()
    rnd = dict(enc_keep=[bct(km(S["ENC0"] + i, 0.5, B, Tin, E)) for i in range(3)],
               encsub_keep=[bct(km(S["ENCSUB0"] + i, 0.5, B, Tsub, E)) for i in range(3)],
               post_keep=[bct(km(S["POSTNET0"] + i, 0.5, B, T, M if i == 4 else 512)) for i in range(5)],
               prenet_keep=[km(S["PRENET1"], 0.5, T, B, Pn), km(S["PRENET2"], 0.5, T, B, Pn)],
               prenet_bert_keep=[km(S["PRENET1_SUB"], 0.5, T, B, Pn), km(S["PRENET2_SUB"], 0.5, T, B, Pn)],
               att_h_keep=km(S["ATT_H"], 0.1, T, B, Ha), att_c_keep=km(S["ATT_C"], 0.1, T, B, Ha),
               att_h_bert_keep=km(S["ATT_H_SUB"], 0.1, T, B, Ha), att_c_bert_keep=km(S["ATT_C_SUB"], 0.1, T, B, Ha),
               dec_h_keep=km(S["DEC_H"], 0.1, T, B, Hd), dec_c_keep=km(S["DEC_C"], 0.1, T, B, Hd),
               sma_noise=ops.rng_normal(seed, S["NOISE"], B * T * Tin).view(T, B, Tin).cpu(),
               sma_noise_bert=ops.rng_normal(seed, S["NOISE_SUB"], B * T * Tsub).view(T, B, Tsub).cpu())
    P = recipe.make_weights(hp)
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    xo, yo = recipe.parse_batch(batch)
    stats = {}
    oo = O.forward(P, hp, xo, training=True, rnd=rnd, new_stats=stats)
    for k, a, b in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), out, oo):
        assert maxabs(a, b.detach()) < 2e-4, k
    lo = O.loss(oo, yo)[0]
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-5
    bad = {}
    for k, p in m.named_parameters():
        ref = P[k].grad
        if ref is None:
            assert p.grad is None, k
            continue
        # conv biases feeding a train-mode BatchNorm have a mathematically zero gradient (both sides hold rounding
        # noise of the column sums, ~1e-7): absolute bound for them, relative for everything else
        if k.endswith("conv.bias") and float(ref.abs().max()) < 1e-5:
            if not maxabs(p.grad, ref) < 1e-6:
                bad[k] = maxabs(p.grad, ref)
            continue
        err = maxabs(p.grad, ref) / max(float(ref.abs().max()), 1e-4)
        if not err < 1e-3:
            bad[k] = err
    assert not bad, bad
    sd = m.state_dict()
    for k, v in stats.items():
        assert maxabs(sd[k].float(), v.float()) < 1e-5, k


def test_deferred_weight_gradients_and_prologue_are_bitwise_neutral():
    """The stream-level schedule (decoder prologue under the encoders, weight-gradient tail of the decoder backward on
    the library's side stream, joined at the end of backward — off by default) changes no arithmetic: every gradient is bit-identical
    to the one-stream schedule, two iterations in a row (T = 40 takes the two-chain path)."""
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    hp = hp_for(SMA)
    B, Tin, Tsub, T = 4, 21, 13, 40
    m, hps = build_model(SMA, train=True)
    m.decoder.prenet_dropout = True
    batch = recipe.make_batch(hp, B, Tin, Tsub, T)
    x, y = m.parse_batch(batch)
    res = {}
    for mode in (True, False):
        m.overlap_encoders, m.decoder.defer_weight_grads = mode, mode
        m._t2_calls, m.decoder._t2_calls = 0, 0
        out = []
        for it in range(2):
            m.zero_grad()
            loss = Tacotron2Loss()(m(x), y, x)[0]
            loss.backward()
            out.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        res[mode] = out
    torch.cuda.synchronize()
    for it in range(2):
        assert res[True][it].keys() == res[False][it].keys()
        for k in res[True][it]:
            assert torch.equal(res[True][it][k], res[False][it][k]), (it, k)


def test_bf16_operand_mode_stays_close_to_fp32_golden():
    """bf16-operand GEMMs (fp32 accumulate and state) are the throughput mode, not the parity mode: this
    records how far the outputs move on the BASELINE-shaped golden case (400 recurrent frames)."""
    from tacotron2_subword_amd import _lib as L
    g = load_golden("sma_baseline_eval")
    B, Tin, Tsub, T, _ = (int(v) for v in g["meta"])
    m, hps = build_model(SMA)
    x, y = m.parse_batch(recipe.make_batch(hp_for(SMA), B, Tin, Tsub, T))
    L.set_precision("bf16")
    try:
        with torch.no_grad():
            out = m(x)
    finally:
        L.set_precision("f32")
    errs = {k: maxabs(v, g[k]) for k, v in zip(("mel", "mel_postnet", "gate", "align"), out)}
    print("bf16-operand max-abs error vs reference fp32:", errs)
    assert errs["mel"] < 0.06 and errs["mel_postnet"] < 0.07 and errs["gate"] < 0.03 and errs["align"] < 0.004, errs    # observed 0.030 / 0.033 / 0.013 / 0.0019


@pytest.mark.parametrize("att,name", [(SMA, "sma_infer"), (LSA, "lsa_infer")])
def test_bf16_decode_loop_stays_close_to_fp32_golden(att, name):
    """bf16-operand decode loop (whole-cell weight shadows, bf16 input rows): outputs stay near the fp32 golden frames."""
    from tacotron2_subword_amd import _lib as L
    g = load_golden(name)
    _, Tin, Tsub, steps = (int(v) for v in g["meta"])
    m, hps = build_model(att)
    b = recipe.make_batch(hp_for(att), 1, Tin, Tsub, 8, seed=4321, ragged=False)
    ids, sub, pcls, bcls = b[0].cuda(), b[6].cuda(), b[7].cuda(), b[8].cuda()
    m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, steps
    L.set_precision("bf16")
    try:
        r = m.inference(ids, sub, pcls, bcls)
        torch.cuda.synchronize()
    finally:
        L.set_precision("f32")
    assert tuple(r[0].shape) == g["fixed_mel"].shape
    assert maxabs(r[0], g["fixed_mel"]) < 0.05 and maxabs(r[1], g["fixed_mel_postnet"]) < 0.1
    assert maxabs(r[3], g["fixed_align"]) < 0.02


@pytest.mark.parametrize("att", [SMA, LSA, FA2, GMM, DCA])
def test_bf16_mode_gradients_track_fp32(att):
    """Same batch, same RNG seeds, default dims, every attention type: the bf16-operand mode (large GEMMs and the recurrent
    step GEMMs in bf16, fp32 accumulate/state; SMA / LSA / ForwardAttentionV2 through the persistent chains, GMM / DCA through
    the per-step launches) must give outputs and gradients close to the fp32 parity path."""
    from tacotron2_subword_amd import _lib as L
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    hp = hp_for(att)
    B, Tin, Tsub, T = 4, 24, 16, 40
    batch = recipe.make_batch(hp, B, Tin, Tsub, T)
    grads, losses, outs = {}, {}, {}
    for mode in ("f32", "bf16"):
        m, hps = build_model(att, train=True)
        m.decoder.prenet_dropout = True
        m._t2_calls, m.decoder._t2_calls = 0, 0
        x, y = m.parse_batch(batch)
        L.set_precision(mode)
        try:
            out = m(x)
            loss = Tacotron2Loss()(out, y, x)[0]
            loss.backward()
        finally:
            L.set_precision("f32")
        losses[mode] = float(loss.detach())
        outs[mode] = [o.detach().float().cpu() for o in out]
        grads[mode] = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}
    oerr = {n: maxabs(a, b) for n, a, b in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), outs["bf16"], outs["f32"])}
    print(att, "bf16 vs fp32 outputs (training mode, same dropout bits):", oerr)
    assert oerr["mel"] < 0.03 and oerr["gate"] < 0.012 and oerr["align"] < 0.006 and oerr["align_bert"] < 0.004, oerr     # observed <= 0.013 / 0.005 / 0.0024 / 0.0015
    assert set(grads["bf16"]) == set(grads["f32"])
    assert abs(losses["bf16"] - losses["f32"]) < 0.01 * abs(losses["f32"])
    worst = {}
    for k, g32 in grads["f32"].items():
        n = float(g32.norm())
        if n < 1e-6:
            continue
        rel = float((grads["bf16"][k] - g32).norm()) / n
        if rel > 0.15:
            worst[k] = rel
    print("bf16 vs fp32: loss", losses, "worst relative gradient deviations:", dict(sorted(worst.items(), key=lambda kv: -kv[1])[:5]))
    assert not worst, worst


@pytest.mark.parametrize("mode", ["L2", "KL"])
def test_alignment_guide_loss_gradients_reach_the_decoder(mode):
    """alignloss = L2 / KL (loss_function.py:29-54): the loss is a host reduction on the alignments, its gradient
    enters the HIP decoder backward through d_align / d_align_sub; whole-model gradients vs the oracle's autograd."""
    att = SMA
    hp = hp_for(att)
    B, Tin, Tsub, T = 2, 10, 10, 12                            # both branches compare BOTH alignments with the phone-level target
    m, hps = build_model(att, train=False)
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    from tacotron2_subword_amd.utils import Alignment_Generator
    batch = list(recipe.make_batch(hp, B, Tin, Tsub, T, ragged=False))
    dur = torch.ones(B, Tin, dtype=torch.long)
    dur[:, 0] = T - Tin + 1
    batch[9] = Alignment_Generator()(dur)                      # [B, T, Tin] hard alignment
    batch = tuple(batch)
    x, y = m.parse_batch(batch)
    out = m(x)
    res = Tacotron2Loss(mode)(out, y, x, 0)
    res[0].backward()
    P = recipe.make_weights(hp)
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    xo, yo = recipe.parse_batch(batch)
    oo = O.forward(P, hp, xo, training=False)
    ro = O.loss_align(oo, yo, xo, mode, 0)
    ro[0].backward()
    for a, b in zip(res, ro):
        assert abs(float(a.detach()) - float(b.detach())) < 2e-5 * max(1.0, abs(float(b.detach())))
    bad = {}
    for k, p in m.named_parameters():
        ref = P[k].grad
        if ref is None:
            assert p.grad is None, k
            continue
        err = maxabs(p.grad, ref) / max(float(ref.abs().max()), 1e-7)
        if not err < 1e-3:
            bad[k] = err
    assert not bad, bad


def test_bf16_steps_at_batch_96_track_fp32():
    """B in (64, 128]: the bf16-operand recurrent steps use four 32-row tiles per workgroup; outputs and memory
    gradients stay close to the fp32 path (same bound style as the B <= 64 closeness tests)."""
    from tacotron2_subword_amd import _lib as L
    from tacotron2_subword_amd import ops
    from helpers import to_dev
    hp = hp_for(SMA)
    B, Tin, Tsub, T = 96, 24, 17, 20
    P = to_dev(recipe.make_weights(hp, seed=2))
    dims = L.dims_from_hparams(hp)
    W = L.decoder_weights(P, dims.attention_kind)
    g = torch.Generator(device="cuda").manual_seed(3)
    mem = torch.randn(B, Tin, 512, device="cuda", generator=g) * 0.5
    mems = torch.randn(B, Tsub, 512, device="cuda", generator=g) * 0.5
    mels = torch.randn(B, 80, T, device="cuda", generator=g)
    tl = torch.full((B,), Tin, device="cuda"); bl = torch.full((B,), Tsub, device="cuda")
    dmel = torch.randn(B, T, 80, device="cuda", generator=g); dgate = torch.randn(B, T, device="cuda", generator=g)
    res = {}
    for mode in ("f32", "bf16"):
        L.set_precision(mode)
        try:
            dp = ops.decoder_forward(W, dims, mem, mems, tl, bl, mels, training=False, prenet_dropout=False, seed=0)
            G, dm, dms = ops.decoder_backward(W, P, dims, dp, mem, mems, dmel, dgate, training=False, prenet_dropout=False, seed=0)
            torch.cuda.synchronize()
            res[mode] = (dp.mel.clone(), dp.gate.clone(), dp.align.clone(), dm.clone(), G["decoder.decoder_rnn.weight_hh"].clone())
        finally:
            L.set_precision("f32")
    a, b = res["f32"], res["bf16"]
    assert maxabs(a[0], b[0]) < 0.05 and maxabs(a[1], b[1]) < 0.05 and maxabs(a[2], b[2]) < 0.02
    for x, y in zip(a[3:], b[3:]):
        assert float((x - y).norm() / x.norm()) < 0.15
