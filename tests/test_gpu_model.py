"""GPU parity of the drop-in module API (BERT_Tacotron2.parse_batch / forward / inference,
load_model) against the golden vectors recorded from the reference and the oracle's autograd."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import tacotron2_oracle as O

from helpers import LSA, SMA, hp_for, load_golden, maxabs

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


@pytest.mark.parametrize("att,name", [(SMA, "sma_small_eval"), (LSA, "lsa_small_eval"), (SMA, "sma_baseline_eval")])
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


def test_backward_eval_mode_vs_oracle_autograd(monkeypatch):
    """Whole-model gradients (encoders, converters, decoder, postnet) with every source of
    randomness off (BN running statistics, no dropout, no noise): the HIP decoder backward inside
    torch autograd vs the oracle's autograd.  (The interim torch BiLSTM needs train mode for its
    backward, so the module is in train mode with BN / decoder in eval and dropout patched out.)"""
    att = SMA
    hp = hp_for(att)
    B, Tin, Tsub, T = 3, 13, 8, 12
    m, hps = build_model(att, train=True)
    import tacotron2_subword_amd.model as M
    monkeypatch.setattr(M.F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.eval()
    m.decoder.eval()
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    batch = recipe.make_batch(hp, B, Tin, Tsub, T)
    x, y = m.parse_batch(batch)
    out = m(x)
    loss = Tacotron2Loss()(out, y, x)[0]
    loss.backward()
    # oracle
    P = recipe.make_weights(hp)
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    xo, yo = recipe.parse_batch(batch)
    oo = O.forward(P, hp, xo, training=False)
    lo = O.loss(oo, yo)[0]
    lo.backward()
    assert abs(float(loss) - float(lo)) < 1e-5
    bad = {}
    for k, p in m.named_parameters():
        ref = P[k].grad
        if ref is None:
            assert p.grad is None, k                 # dead decoder_rnn_bert
            continue
        err = maxabs(p.grad, ref) / max(float(ref.abs().max()), 1e-7)
        if not err < 5e-4:
            bad[k] = err
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
