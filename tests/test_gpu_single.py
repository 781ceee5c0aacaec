"""GPU parity of the single-stream `Tacotron2` class (the API GTA.py:57-59 / inference.py:302,334 expect).

The reference ships no such class (model.py only defines BERT_Tacotron2), so there is nothing of the
reference's to record: parity here is against the CPU oracle's single-stream restatement only
("parity unpinned by the reference").  The kernels are the dual-stream ones run with n_streams = 1, and
those ARE pinned by the golden fixtures in test_gpu_model.py."""
import pytest
import torch

from oracle import recipe
from oracle import tacotron2_oracle as O

from helpers import LSA, SMA, hp_for, maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build_single(att, train=False):
    from tacotron2_subword_amd.hparams import create_hparams
    from tacotron2_subword_amd.model import Tacotron2
    hps = create_hparams()
    hps.attention = att
    m = Tacotron2(hps)
    m.load_state_dict(recipe.make_weights(hp_for(att), single=True))
    m = m.cuda()
    m.train(train)
    m.decoder.prenet_dropout = False
    return m


def single_batch(hp, B, Tin, T, **kw):
    text, il, _, mel, gate, ol = recipe.make_batch(hp, B, Tin, 4, T, **kw)[:6]
    return (text, il, mel, gate, ol)


@pytest.mark.parametrize("att", [SMA, LSA])
def test_single_forward_eval_vs_oracle(att):
    hp = hp_for(att)
    B, Tin, T = 3, 14, 11
    m = build_single(att)
    batch = single_batch(hp, B, Tin, T)
    x, y = m.parse_batch(batch)
    with torch.no_grad():
        out = m(x)
    P = recipe.make_weights(hp, single=True)
    text, il, mel, gate, ol = batch
    with torch.no_grad():
        ref = O.forward_single(P, hp, (text, il, mel, int(il.max()), ol))
    assert len(out) == 4
    for k, a, b in zip(("mel", "mel_postnet", "gate", "align"), out, ref):
        assert maxabs(a, b) < TOL, k


@pytest.mark.parametrize("att", [SMA, LSA])
def test_single_backward_eval_vs_oracle_autograd(att):
    hp = hp_for(att)
    B, Tin, T = 3, 12, 10
    m = build_single(att)
    batch = single_batch(hp, B, Tin, T)
    x, y = m.parse_batch(batch)
    out = m(x)
    # Tacotron2Loss's default branch only reads mel / postnet / gate (loss_function.py:12-22,65-66)
    mel_t, gate_t = y
    F = torch.nn.functional
    loss = F.mse_loss(out[0], mel_t) + F.mse_loss(out[1], mel_t) + F.binary_cross_entropy_with_logits(out[2].reshape(-1, 1),
                                                                                                      gate_t.reshape(-1, 1))
    loss.backward()
    P = recipe.make_weights(hp, single=True)
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    text, il, mel, gate, ol = batch
    oo = O.forward_single(P, hp, (text, il, mel, int(il.max()), ol))
    lo = O.loss(oo, (mel, gate))[0]
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-5
    bad = {}
    for k, p in m.named_parameters():
        ref = P[k].grad
        assert ref is not None and p.grad is not None, k          # no dead parameters in the single-stream model
        err = maxabs(p.grad, ref) / max(float(ref.abs().max()), 1e-7)
        if not err < 5e-4:
            bad[k] = err
    assert not bad, bad


@pytest.mark.parametrize("att", [SMA, LSA])
def test_single_inference_vs_oracle(att):
    hp = hp_for(att)
    Tin, steps = 15, 12
    m = build_single(att)
    ids = recipe.make_batch(hp, 1, Tin, 4, 8, seed=4321, ragged=False)[0]
    P = recipe.make_weights(hp, single=True)
    with torch.no_grad():
        ref = O.inference_single(P, hp, ids, max_decoder_steps=steps, gate_threshold=2.0)
    assert ref[4] is False
    m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, steps
    r = m.inference(ids.cuda())
    assert len(r) == 4
    for k, a, b in zip(("mel", "mel_postnet", "gate", "align"), r, ref):
        assert maxabs(a, b) < TOL, k
    # stop rule: threshold between two recorded gate values -> same stop frame as the oracle
    sg = torch.sigmoid(ref[2].flatten())
    order = torch.argsort(sg, descending=True)
    i = int(order[0])
    if i > 0 and float(sg[:i].max()) < float(sg[i]):
        thr = 0.5 * (float(sg[:i].max()) + float(sg[i]))
        with torch.no_grad():
            ref2 = O.inference_single(P, hp, ids, max_decoder_steps=1000, gate_threshold=thr)
        m.decoder.gate_threshold, m.decoder.max_decoder_steps = thr, 1000
        r2 = m.inference(ids.cuda())
        assert r2[0].shape == ref2[0].shape                        # stop frame bit-exact
        assert maxabs(r2[0], ref2[0]) < TOL and maxabs(r2[1], ref2[1]) < TOL
