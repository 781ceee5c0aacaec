"""GPU parity of the hand-written decoder backward (BPTT through both attention LSTMs, the
stepwise-monotonic / location-sensitive attention recurrence, the decoder LSTM, prenets and projections) against
torch.autograd run on the CPU oracle — the reference has no backward source of its own
(it is autograd of model.py:392-428), so this is the oracle for gradients."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import tacotron2_oracle as O

from helpers import DCA, FA2, GMM, LSA, SMA, hp_for, maxabs, oracle_memories, tiny_hp, to_dev

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


@pytest.mark.parametrize("cfg", ["tiny_eval", "tiny_train", "tiny_b33", "tiny_long", "tiny_T40", "tiny_one_frame", "tiny_long_memory", "default_train", "default_align"])
@pytest.mark.parametrize("att", [SMA, LSA, FA2, GMM, DCA])
def test_decoder_backward_vs_autograd(env, cfg, att):
    L, ops = env
    training = cfg in ("tiny_train", "default_train", "default_align", "tiny_b33", "tiny_long", "tiny_T40", "tiny_long_memory")
    with_align = cfg in ("default_align", "tiny_b33", "tiny_long")
    if cfg == "tiny_long":                       # several 32-position chunks per attention step, ragged tails
        hp = tiny_hp(att)
        B, Tin, Tsub, T = 2, 70, 37, 6
    elif cfg == "tiny_one_frame":                # T = 1: no recurrent gradient at all
        hp = tiny_hp(att)
        B, Tin, Tsub, T = 3, 9, 6, 1
    elif cfg == "tiny_long_memory":              # memories past the LDS-resident fast paths (LSA: scalar fallback kernel)
        hp = tiny_hp(att)
        B, Tin, Tsub, T = 2, 300, 170, 3
    elif cfg == "tiny_T40":                      # long enough for the chunked two-stream schedule (3 chunks of 16 steps)
        hp = tiny_hp(att)
        B, Tin, Tsub, T = 3, 12, 9, 40
    elif cfg.startswith("tiny"):
        hp = tiny_hp(att)
        B, Tin, Tsub, T = (33 if cfg == "tiny_b33" else 5), 11, 7, 9
    else:
        hp = hp_for(att)
        B, Tin, Tsub, T = 3, 13, 8, 12
    seed = 99173
    P = recipe.make_weights(hp, seed=5)
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T, seed=B))
    mem, mem_sub = oracle_memories(P, hp, x)
    rnd = hip_rnd(ops, L, hp, seed, B, T, Tin, Tsub) if training else None

    # ---- oracle: autograd, in fp32 (the reference's arithmetic) and in fp64 (ground truth for the tolerance)
    dec_keys = ["decoder." + k for k in L.decoder_param_keys({SMA: L.ATTN_SMA, GMM: L.ATTN_GMM, DCA: L.ATTN_DCA}.get(att, L.ATTN_LSA))]
    g = torch.Generator().manual_seed(17)
    R_mel, R_gate = torch.randn(B, hp["n_mel_channels"], T, generator=g), torch.randn(B, T, generator=g)
    R_al, R_alb = torch.randn(B, T, Tin, generator=g), torch.randn(B, T, Tsub, generator=g)

    def autograd(dt):
        cv = lambda v: v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v
        Pg = {k: (cv(v).clone().requires_grad_(True) if k in dec_keys else cv(v)) for k, v in P.items()}
        r = None if rnd is None else {k: ([cv(a) for a in v] if isinstance(v, list) else cv(v)) for k, v in rnd.items()}
        mem_g, mem_sub_g = cv(mem).clone().requires_grad_(True), cv(mem_sub).clone().requires_grad_(True)
        mel, gate, al, alb = O.decoder_forward(mem_g, mem_sub_g, cv(x[3]), x[1], x[2], Pg, hp, r)
        loss = (mel * cv(R_mel)).sum() + (gate * cv(R_gate)).sum()
        if with_align:
            loss = loss + (al * cv(R_al)).sum() + (alb * cv(R_alb)).sum()
        loss.backward()
        out = {k: Pg[k].grad for k in dec_keys}
        out["d_memory"], out["d_memory_sub"] = mem_g.grad, mem_sub_g.grad
        return mel.detach(), out

    mel, g32 = autograd(torch.float32)
    _, g64 = autograd(torch.float64)

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
        return maxabs(a.double(), ref.double()) / max(float(ref.abs().max()), 1e-6)

    # The softmax / sigmoid-recurrence gradients cancel heavily (sum_j de_j = 0 for a softmax), so even torch's fp32
    # autograd is only good to a few 1e-4 .. 1e-3 on the attention parameters, and which tensor gets the larger error
    # depends on summation order (scripts/grad_check_case.py).  A tensor passes when it is within RTOL of the fp64
    # gradient, or no worse than 3x the fp32 oracle's own rounding error against fp64 — for the attention parameters
    # the oracle's worst error over that group is the yardstick.
    G = dict(G)
    G["d_memory"], G["d_memory_sub"] = dmem, dmems
    att_noise = max(rel(g32[k], g64[k]) for k in g64 if "attention_layer" in k)
    bad = {}
    for k, ref in g64.items():
        noise = att_noise if "attention_layer" in k else rel(g32[k], ref)
        err, tol = rel(G[k], ref), max(RTOL, 3.0 * noise)
        if not err < tol:
            bad[k] = (err, tol)
    assert not bad, bad


def test_two_stream_schedule_is_bitwise_identical(env):
    """The side-stream schedule only reorders independent launches: outputs and gradients equal the one-stream run bit for bit."""
    L, ops = env
    hp = tiny_hp(SMA)
    B, Tin, Tsub, T = 4, 12, 9, 50
    P = recipe.make_weights(hp, seed=5)
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T, seed=B))
    mem, mem_sub = oracle_memories(P, hp, x)
    dims = L.dims_from_hparams(hp)
    Pd = to_dev(P)
    W = L.decoder_weights(Pd, dims.attention_kind)
    memd, memsd = mem.cuda().contiguous(), mem_sub.cuda().contiguous()
    g = torch.Generator().manual_seed(3)
    R_mel, R_gate = torch.randn(B, T, hp["n_mel_channels"], generator=g).cuda(), torch.randn(B, T, generator=g).cuda()
    res = []
    for on in (1, 0, 1):
        L.check(L.lib().t2_set_overlap(on))
        dp = ops.decoder_forward(W, dims, memd, memsd, x[1].cuda(), x[2].cuda(), x[3].cuda().contiguous(),
                                 training=True, prenet_dropout=True, seed=7)
        G, dmem, dmems = ops.decoder_backward(W, Pd, dims, dp, memd, memsd, R_mel, R_gate, training=True, prenet_dropout=True, seed=7)
        torch.cuda.synchronize()
        res.append((dp.mel.clone(), dp.gate.clone(), dmem.clone(), dmems.clone(), {k: v.clone() for k, v in G.items()}))
    L.check(L.lib().t2_set_overlap(1))
    for other in res[1:]:
        for a, b in zip(res[0][:4], other[:4]):
            assert torch.equal(a, b)
        for k in res[0][4]:
            assert torch.equal(res[0][4][k], other[4][k]), k


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_shapes_forward_and_backward(env, seed):
    """Seeded random sweep over batch / memory / frame counts and attention kinds (tiny dims): forward within 1e-4 of
    the oracle, gradients within the rule above.  Shapes deliberately avoid multiples of the tile sizes."""
    L, ops = env
    rs = np.random.RandomState(1000 + seed)
    att = [SMA, LSA, FA2, GMM, DCA][seed % 5]
    B, Tin, Tsub, T = int(rs.randint(1, 71)), int(rs.randint(1, 91)), int(rs.randint(1, 51)), int(rs.randint(1, 46))
    hp = tiny_hp(att)
    P = recipe.make_weights(hp, seed=40 + seed)
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T, seed=seed))
    mem, mem_sub = oracle_memories(P, hp, x)
    kind = {SMA: L.ATTN_SMA, GMM: L.ATTN_GMM, DCA: L.ATTN_DCA}.get(att, L.ATTN_LSA)
    dec_keys = ["decoder." + k for k in L.decoder_param_keys(kind)]
    g = torch.Generator().manual_seed(seed)
    R_mel, R_gate = torch.randn(B, hp["n_mel_channels"], T, generator=g), torch.randn(B, T, generator=g)
    R_al = torch.randn(B, T, Tin, generator=g)

    def autograd(dt):
        cv = lambda v: v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v
        Pg = {k: (cv(v).clone().requires_grad_(True) if k in dec_keys else cv(v)) for k, v in P.items()}
        m1, m2 = cv(mem).clone().requires_grad_(True), cv(mem_sub).clone().requires_grad_(True)
        mel, gate, al, alb = O.decoder_forward(m1, m2, cv(x[3]), x[1], x[2], Pg, hp, None)
        ((mel * cv(R_mel)).sum() + (gate * cv(R_gate)).sum() + (al * cv(R_al)).sum()).backward()
        out = {k: Pg[k].grad for k in dec_keys}
        out["d_memory"], out["d_memory_sub"] = m1.grad, m2.grad
        return (mel.detach(), gate.detach(), al.detach(), alb.detach()), out

    fwd32, g32 = autograd(torch.float32)
    _, g64 = autograd(torch.float64)
    dims = L.dims_from_hparams(hp)
    Pd = to_dev(P)
    W = L.decoder_weights(Pd, dims.attention_kind)
    memd, memsd = mem.cuda().contiguous(), mem_sub.cuda().contiguous()
    dp = ops.decoder_forward(W, dims, memd, memsd, x[1].cuda(), x[2].cuda(), x[3].cuda().contiguous(), training=False, prenet_dropout=False, seed=0)
    for got, ref in zip((dp.mel.transpose(1, 2), dp.gate, dp.align, dp.align_sub), fwd32):
        assert maxabs(got, ref) < 1e-4, (att, B, Tin, Tsub, T)
    G, dmem, dmems = ops.decoder_backward(W, Pd, dims, dp, memd, memsd, R_mel.transpose(1, 2).contiguous().cuda(), R_gate.cuda().contiguous(),
                                          training=False, prenet_dropout=False, seed=0, d_align=R_al.cuda().contiguous())
    torch.cuda.synchronize()
    G = dict(G)
    G["d_memory"], G["d_memory_sub"] = dmem, dmems
    rel = lambda a, ref: maxabs(a.double(), ref.double()) / max(float(ref.abs().max()), 1e-6)
    att_noise = max(rel(g32[k], g64[k]) for k in g64 if "attention_layer" in k)
    bad = {}
    for k, ref in g64.items():
        noise = att_noise if "attention_layer" in k else rel(g32[k], ref)
        err, tol = rel(G[k], ref), max(RTOL, 3.0 * noise)
        if not err < tol:
            bad[k] = (err, tol)
    assert not bad, (att, B, Tin, Tsub, T, bad)
