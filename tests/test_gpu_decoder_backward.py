"""GPU parity of the hand-written decoder backward (BPTT through both attention LSTMs, the
stepwise-monotonic attention recurrence, the decoder LSTM, prenets and projections) against
torch.autograd run on the CPU oracle — the reference has no backward source of its own
(it is autograd of model.py:392-428), so this is the oracle for gradients."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import tacotron2_oracle as O

from helpers import SMA, hp_for, maxabs, oracle_memories, tiny_hp, to_dev

pytestmark = pytest.mark.gpu
RTOL = 3e-4          # max-abs error relative to the largest reference entry of each gradient tensor


@pytest.fixture(scope="module")
def env():
    from tacotron2_subword_amd import _lib as L
    from tacotron2_subword_amd import ops
    return L, ops


def hip_rnd(ops, L, hp, seed, B, T, Tin, Tsub):
    Pn, Ha, Hd = hp["prenet_dim"], hp["attention_rnn_dim"], hp["decoder_rnn_dim"]
    S = L.SITE

    def km(site, p, *shape):
        return ops.rng_keep_mask(seed, S[site], int(np.prod(shape)), p).view(*shape).float().cpu()

    return dict(prenet_keep=[km("PRENET1", 0.5, T, B, Pn), km("PRENET2", 0.5, T, B, Pn)],
                prenet_bert_keep=[km("PRENET1_SUB", 0.5, T, B, Pn), km("PRENET2_SUB", 0.5, T, B, Pn)],
                att_h_keep=km("ATT_H", 0.1, T, B, Ha), att_c_keep=km("ATT_C", 0.1, T, B, Ha),
                att_h_bert_keep=km("ATT_H_SUB", 0.1, T, B, Ha), att_c_bert_keep=km("ATT_C_SUB", 0.1, T, B, Ha),
                dec_h_keep=km("DEC_H", 0.1, T, B, Hd), dec_c_keep=km("DEC_C", 0.1, T, B, Hd),
                sma_noise=ops.rng_normal(seed, S["NOISE"], B * T * Tin).view(T, B, Tin).cpu(),
                sma_noise_bert=ops.rng_normal(seed, S["NOISE_SUB"], B * T * Tsub).view(T, B, Tsub).cpu())


@pytest.mark.parametrize("cfg", ["tiny_eval", "tiny_train", "tiny_b33", "default_train", "default_align"])
def test_decoder_backward_vs_autograd(env, cfg):
    L, ops = env
    training = cfg in ("tiny_train", "default_train", "default_align", "tiny_b33")
    with_align = cfg in ("default_align", "tiny_b33")
    if cfg.startswith("tiny"):
        hp = tiny_hp(SMA)
        B, Tin, Tsub, T = (33 if cfg == "tiny_b33" else 5), 11, 7, 9
    else:
        hp = hp_for(SMA)
        B, Tin, Tsub, T = 3, 13, 8, 12
    seed = 99173
    P = recipe.make_weights(hp, seed=5)
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T, seed=B))
    mem, mem_sub = oracle_memories(P, hp, x)
    rnd = hip_rnd(ops, L, hp, seed, B, T, Tin, Tsub) if training else None

    # ---- oracle: autograd
    dec_keys = ["decoder." + k for k in L.DECODER_PARAM_KEYS_SMA]
    Pg = {k: (v.clone().requires_grad_(True) if k in dec_keys else v) for k, v in P.items()}
    mem_g, mem_sub_g = mem.clone().requires_grad_(True), mem_sub.clone().requires_grad_(True)
    mel, gate, al, alb = O.decoder_forward(mem_g, mem_sub_g, x[3], x[1], x[2], Pg, hp, rnd)
    g = torch.Generator().manual_seed(17)
    R_mel, R_gate = torch.randn(mel.shape, generator=g), torch.randn(gate.shape, generator=g)
    R_al, R_alb = torch.randn(al.shape, generator=g), torch.randn(alb.shape, generator=g)
    loss = (mel * R_mel).sum() + (gate * R_gate).sum()
    if with_align:
        loss = loss + (al * R_al).sum() + (alb * R_alb).sum()
    loss.backward()

    # ---- HIP
    dims = L.dims_from_hparams(hp)
    Pd = to_dev(P)
    W = L.decoder_weights(Pd, dims.attention_kind)
    memd, memsd = mem.cuda().contiguous(), mem_sub.cuda().contiguous()
    dp = ops.decoder_forward(W, dims, memd, memsd, x[1].cuda(), x[2].cuda(), x[3].cuda().contiguous(),
                             training=training, prenet_dropout=training, seed=seed)
    assert maxabs(dp.mel.cpu().transpose(1, 2), mel) < 1e-4
    G, dmem, dmems = ops.decoder_backward(W, Pd, dims, dp, memd, memsd,
                                          R_mel.transpose(1, 2).contiguous().cuda(), R_gate.cuda().contiguous(),
                                          training=training, prenet_dropout=training, seed=seed,
                                          d_align=R_al.cuda().contiguous() if with_align else None,
                                          d_align_sub=R_alb.cuda().contiguous() if with_align else None)
    torch.cuda.synchronize()

    def rel(a, ref):
        return maxabs(a, ref) / max(float(ref.abs().max()), 1e-6)

    errs = {"d_memory": rel(dmem, mem_g.grad), "d_memory_sub": rel(dmems, mem_sub_g.grad)}
    for k in dec_keys:
        errs[k] = rel(G[k], Pg[k].grad)
    bad = {k: v for k, v in errs.items() if not v < RTOL}
    assert not bad, bad
