"""GPU parity of the HIP decoder (teacher-forced and autoregressive) against the CPU oracle and
against the golden vectors recorded from the reference.  Tolerance is the north-star contract:
mel / gate max-abs error < 1e-4 in fp32, stop-frame index bit-exact."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import tacotron2_oracle as O

from helpers import DCA, FA2, GMM, LSA, SMA, hp_for, load_golden, maxabs, oracle_memories, tiny_hp, to_dev

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def env():
    from tacotron2_subword_amd import _lib as L
    from tacotron2_subword_amd import ops
    return L, ops


def run_hip_decoder(env, P, hp, mem, mem_sub, tl, bl, mels, training=False, prenet_dropout=False, seed=0):
    L, ops = env
    dims = L.dims_from_hparams(hp)
    Pd = to_dev(P)
    W = L.decoder_weights(Pd, dims.attention_kind)
    dp = ops.decoder_forward(W, dims, mem.cuda().contiguous(), mem_sub.cuda().contiguous(), tl.cuda(), bl.cuda(),
                             mels.cuda().contiguous(), training=training, prenet_dropout=prenet_dropout, seed=seed, keep=Pd)
    torch.cuda.synchronize()
    return dp


@pytest.mark.parametrize("att,name", [(SMA, "sma_small_eval"), (LSA, "lsa_small_eval"), (SMA, "sma_baseline_eval"), (FA2, "fa2_small_eval"), (GMM, "gmm_small_eval"), (DCA, "dca_small_eval")])
def test_teacher_forced_vs_golden(env, att, name):
    g = load_golden(name)
    B, Tin, Tsub, T, _ = (int(v) for v in g["meta"])
    hp = hp_for(att)
    P = recipe.make_weights(hp)
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T))
    mem, mem_sub = torch.from_numpy(g["memory"]), torch.from_numpy(g["memory_sub"])
    dp = run_hip_decoder(env, P, hp, mem, mem_sub, x[1], x[2], x[3])
    # golden mel/gate are masked by parse_output; compare the valid frames
    ol = x[5]
    mel = dp.mel.cpu().transpose(1, 2)          # [B,80,T]
    valid = O.get_mask_from_lengths(ol, T)
    assert maxabs(mel * valid[:, None, :], g["mel"]) < TOL
    gate = dp.gate.cpu()
    assert maxabs(torch.where(valid, gate, torch.full_like(gate, 1e3)), g["gate"]) < TOL
    assert maxabs(dp.align.cpu(), g["align"]) < TOL
    assert maxabs(dp.align_sub.cpu(), g["align_bert"]) < TOL
    # recurrent state of the first / last step (nothing drifts over T steps)
    Ha, E, Hd = hp["attention_rnn_dim"], hp["encoder_embedding_dim"], hp["decoder_rnn_dim"]
    din = dp.view("din", T, B, 2 * Ha + 2 * E).cpu()             # time-major workspace
    assert maxabs(din[T - 1, :, :Ha], g["steplast_att_h"]) < TOL
    assert maxabs(din[T - 1, :, Ha:Ha + E], g["steplast_ctx"]) < TOL
    assert maxabs(din[0, :, Ha + E:2 * Ha + E], g["step0_att_h_bert"]) < TOL
    dout = dp.view("dout", T, B, Hd + 2 * E).cpu()
    assert maxabs(dout[T - 1, :, :Hd], g["steplast_dec_h"]) < TOL


@pytest.mark.parametrize("att", [SMA, LSA, FA2, GMM, DCA])
@pytest.mark.parametrize("B", [1, 5, 33])
def test_teacher_forced_tiny_vs_oracle(env, att, B):
    """Small dims, ragged lengths, batch sizes around the 32-row MFMA tile edge."""
    hp = tiny_hp(att)
    P = recipe.make_weights(hp, seed=3)
    Tin, Tsub, T = 11, 7, 9
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T, seed=B))
    mem, mem_sub = oracle_memories(P, hp, x)
    with torch.no_grad():
        mel, gate, al, alb = O.decoder_forward(mem, mem_sub, x[3], x[1], x[2], P, hp)
    dp = run_hip_decoder(env, P, hp, mem, mem_sub, x[1], x[2], x[3])
    assert maxabs(dp.mel.cpu().transpose(1, 2), mel) < TOL
    assert maxabs(dp.gate.cpu(), gate) < TOL
    assert maxabs(dp.align.cpu(), al) < TOL
    assert maxabs(dp.align_sub.cpu(), alb) < TOL


@pytest.mark.parametrize("att", [SMA, LSA])
def test_teacher_forced_training_mode_replay(env, att):
    """Training mode: dropout keep-bits and SMA noise drawn by the HIP RNG are exported through
    the C ABI and replayed through the oracle."""
    L, ops = env
    hp = hp_for(att)
    P = recipe.make_weights(hp)
    B, Tin, Tsub, T = 3, 13, 8, 12
    seed = 20240607
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T))
    mem, mem_sub = oracle_memories(P, hp, x)
    Pn, Ha, Hd = hp["prenet_dim"], hp["attention_rnn_dim"], hp["decoder_rnn_dim"]
    S = L.SITE

    def km(site, p, *shape):        # HIP index order is [T,B,*], the oracle's layout
        return ops.rng_keep_mask(seed, S[site], int(np.prod(shape)), p).view(*shape).float().cpu()

    rnd = dict(prenet_keep=[km("PRENET1", 0.5, T, B, Pn), km("PRENET2", 0.5, T, B, Pn)],
               prenet_bert_keep=[km("PRENET1_SUB", 0.5, T, B, Pn), km("PRENET2_SUB", 0.5, T, B, Pn)],
               att_h_keep=km("ATT_H", 0.1, T, B, Ha), att_c_keep=km("ATT_C", 0.1, T, B, Ha),
               att_h_bert_keep=km("ATT_H_SUB", 0.1, T, B, Ha), att_c_bert_keep=km("ATT_C_SUB", 0.1, T, B, Ha),
               dec_h_keep=km("DEC_H", 0.1, T, B, Hd), dec_c_keep=km("DEC_C", 0.1, T, B, Hd))
    if att == SMA:
        rnd["sma_noise"] = ops.rng_normal(seed, S["NOISE"], B * T * Tin).view(T, B, Tin).cpu()
        rnd["sma_noise_bert"] = ops.rng_normal(seed, S["NOISE_SUB"], B * T * Tsub).view(T, B, Tsub).cpu()
    with torch.no_grad():
        mel, gate, al, alb = O.decoder_forward(mem, mem_sub, x[3], x[1], x[2], P, hp, rnd)
    dp = run_hip_decoder(env, P, hp, mem, mem_sub, x[1], x[2], x[3], training=True, prenet_dropout=True, seed=seed)
    assert maxabs(dp.mel.cpu().transpose(1, 2), mel) < TOL
    assert maxabs(dp.gate.cpu(), gate) < TOL
    assert maxabs(dp.align.cpu(), al) < TOL
    assert maxabs(dp.align_sub.cpu(), alb) < TOL


@pytest.mark.parametrize("att,name", [(SMA, "sma_infer"), (LSA, "lsa_infer")])
def test_inference_vs_golden(env, att, name):
    L, ops = env
    g = load_golden(name)
    _, Tin, Tsub, steps = (int(v) for v in g["meta"])
    hp = hp_for(att)
    P = recipe.make_weights(hp)
    b = recipe.make_batch(hp, 1, Tin, Tsub, 8, seed=4321, ragged=False)
    with torch.no_grad():
        mem = O.front_end(P, hp, b[0], None, b[7], "phone", False)
        mem_sub = O.front_end(P, hp, b[6], None, b[8], "sub", False)
    dims = L.dims_from_hparams(hp)
    Pd = to_dev(P)
    W = L.decoder_weights(Pd, dims.attention_kind)
    # fixed-length run: the stop never fires
    dp, n, stop = ops.decoder_infer(W, dims, mem.cuda(), mem_sub.cuda(), max_steps=steps, gate_threshold=2.0, prenet_dropout=False)
    torch.cuda.synchronize()
    assert n == steps and int(stop[0]) == -1
    assert maxabs(dp.mel.cpu().transpose(1, 2), g["fixed_mel"]) < TOL
    assert maxabs(dp.gate.cpu().unsqueeze(-1), g["fixed_gate"]) < TOL
    assert maxabs(dp.align.cpu(), g["fixed_align"]) < TOL
    assert maxabs(dp.align_sub.cpu(), g["fixed_align_bert"]) < TOL
    # stop rule: index bit-exact
    dp2, n2, stop2 = ops.decoder_infer(W, dims, mem.cuda(), mem_sub.cuda(), max_steps=1000,
                                       gate_threshold=float(g["stop_threshold"]), prenet_dropout=False, poll_every=4)
    torch.cuda.synchronize()
    k = int(g["stop_index"])
    assert int(stop2[0]) == k
    assert k + 1 <= n2 <= k + 1 + 3 * 4                  # the loop may run up to 3 polls past the stop (t2amd.h)
    assert maxabs(dp2.mel[:, :k + 1].cpu().transpose(1, 2), g["stop_mel"]) < TOL


def test_inference_batch_equals_single_items(env):
    """B > 1 has no reference behaviour (model.py:461 breaks); the rule is: every item behaves
    exactly like its own B == 1 run."""
    L, ops = env
    hp = tiny_hp(SMA)
    P = recipe.make_weights(hp, seed=11)
    B, Tin, Tsub, steps = 4, 9, 6, 12
    b = recipe.make_batch(hp, B, Tin, Tsub, 8, seed=77, ragged=False)
    with torch.no_grad():
        mem = O.front_end(P, hp, b[0], None, b[7], "phone", False)
        mem_sub = O.front_end(P, hp, b[6], None, b[8], "sub", False)
    dims = L.dims_from_hparams(hp)
    Pd = to_dev(P)
    W = L.decoder_weights(Pd, dims.attention_kind)
    dp, n, stop = ops.decoder_infer(W, dims, mem.cuda(), mem_sub.cuda(), max_steps=steps, gate_threshold=2.0, prenet_dropout=False)
    torch.cuda.synchronize()
    for i in range(B):
        with torch.no_grad():
            mel, gate, al, alb, flag = O.decoder_inference(mem[i:i + 1], mem_sub[i:i + 1], P, hp, max_decoder_steps=steps, gate_threshold=2.0)
        assert maxabs(dp.mel[i:i + 1].cpu().transpose(1, 2), mel) < TOL
        assert maxabs(dp.align[i:i + 1].cpu(), al) < TOL


def _edge_batch(hp, B, Tin, Tsub, T, lens_in, lens_sub):
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T, seed=11 * B + T, ragged=False))
    x = list(x)
    x[1] = torch.tensor(lens_in, dtype=torch.long)
    x[2] = torch.tensor(lens_sub, dtype=torch.long)
    return x


@pytest.mark.parametrize("att", [SMA, LSA, FA2, GMM, DCA])
@pytest.mark.parametrize("case", ["one_frame", "one_position", "length_one_items", "long_memory", "batch_130"])
def test_teacher_forced_edge_shapes(env, att, case):
    """Edge shapes of the decoder pass (tiny dims): a single frame, a single memory position, items whose valid
    length is 1, memories long enough to leave the LDS-resident fast paths (LSA falls back to the scalar location
    layer), a batch spanning five 32-row tiles."""
    hp = tiny_hp(att)
    P = recipe.make_weights(hp, seed=4)
    if case == "one_frame":
        B, Tin, Tsub, T, li, ls = 3, 9, 6, 1, [9, 7, 5], [6, 6, 2]
    elif case == "one_position":
        B, Tin, Tsub, T, li, ls = 2, 1, 1, 5, [1, 1], [1, 1]
    elif case == "length_one_items":
        B, Tin, Tsub, T, li, ls = 3, 10, 8, 6, [10, 1, 1], [8, 1, 3]
    elif case == "long_memory":
        B, Tin, Tsub, T, li, ls = 2, 300, 170, 4, [300, 211], [170, 95]
    else:
        B, Tin, Tsub, T = 130, 6, 5, 3
        li, ls = [6] * 100 + [3] * 30, [5] * 90 + [2] * 40
    x = _edge_batch(hp, B, Tin, Tsub, T, li, ls)
    mem, mem_sub = oracle_memories(P, hp, x)
    with torch.no_grad():
        mel, gate, al, alb = O.decoder_forward(mem, mem_sub, x[3], x[1], x[2], P, hp)
    dp = run_hip_decoder(env, P, hp, mem, mem_sub, x[1], x[2], x[3])
    assert maxabs(dp.mel.cpu().transpose(1, 2), mel) < TOL
    assert maxabs(dp.gate.cpu(), gate) < TOL
    assert maxabs(dp.align.cpu(), al) < TOL
    assert maxabs(dp.align_sub.cpu(), alb) < TOL


@pytest.mark.parametrize("att", [SMA, LSA])
def test_full_size_properties(env, att):
    """BASELINE-sized pass (B=64, 100/60 positions, 400 frames, default dims) checked through size-independent
    properties, since the oracle cannot run this size in seconds: every item's outputs and memory gradients equal
    those of the same item run alone (items are independent: no cross-item arithmetic in the decoder), permuting the
    batch permutes the outputs, alignment rows are probability-like."""
    L, ops = env
    hp = hp_for(att)
    B, Tin, Tsub, T = 64, 100, 60, 400
    P = to_dev(recipe.make_weights(hp, seed=9))
    dims = L.dims_from_hparams(hp)
    W = L.decoder_weights(P, dims.attention_kind)
    g = torch.Generator(device="cuda").manual_seed(5)
    mem = torch.randn(B, Tin, 512, device="cuda", generator=g) * 0.5
    mems = torch.randn(B, Tsub, 512, device="cuda", generator=g) * 0.5
    mels = torch.randn(B, 80, T, device="cuda", generator=g)
    tl = torch.randint(70, Tin + 1, (B,), device="cuda", generator=g); tl[0] = Tin
    bl = torch.randint(40, Tsub + 1, (B,), device="cuda", generator=g); bl[0] = Tsub
    dmel = torch.randn(B, T, 80, device="cuda", generator=g)
    dgate = torch.randn(B, T, device="cuda", generator=g)

    def run(idx):
        sel = lambda t: t[idx].contiguous()
        dp = ops.decoder_forward(W, dims, sel(mem), sel(mems), sel(tl), sel(bl), sel(mels), training=False, prenet_dropout=False, seed=0)
        G, dm, dms = ops.decoder_backward(W, P, dims, dp, sel(mem), sel(mems), sel(dmel), sel(dgate), training=False, prenet_dropout=False, seed=0)
        torch.cuda.synchronize()
        return dp.mel.clone(), dp.gate.clone(), dp.align.clone(), dp.align_sub.clone(), dm.clone(), dms.clone()

    full = run(torch.arange(B, device="cuda"))
    assert all(bool(torch.isfinite(t).all()) for t in full)
    al, als = full[2], full[3]
    assert float(al.min()) >= 0.0 and float(als.min()) >= 0.0
    rows, rows_s = al.sum(-1), als.sum(-1)
    if att == LSA:
        assert float((rows - 1).abs().max()) < 1e-4 and float((rows_s - 1).abs().max()) < 1e-4      # softmax rows
    else:
        assert float(rows.max()) < 1 + 1e-4 and float(rows_s.max()) < 1 + 1e-4                        # SMA mass only leaks past the end
    for b_idx in (0, 37):                               # an item alone (MT = 1 tile) vs inside the batch of 64 (MT = 2)
        alone = run(torch.tensor([b_idx], device="cuda"))
        for a, f in zip(alone, full):
            ref = f[b_idx:b_idx + 1]
            assert float((a - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    perm = torch.randperm(B, device="cuda", generator=g)
    shuffled = run(perm)
    for s, f in zip(shuffled, full):
        assert float((s - f[perm]).abs().max()) <= 1e-5 * max(1.0, float(f.abs().max()))
