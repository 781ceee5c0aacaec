"""GPU parity of the encoder / postnet building blocks (conv+BN+act+dropout stack, BiLSTM with
packed-sequence semantics, embedding, linear) against the CPU oracle functions and their autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tacotron2_oracle as O

from helpers import maxabs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from tacotron2_subword_amd import _lib as L
    from tacotron2_subword_amd import blocks, ops
    return L, blocks, ops


def rel(a, ref):
    return maxabs(a, ref) / max(float(ref.abs().max()), 1e-6)


def test_linear_and_embedding(env):
    L, blocks, ops = env
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 7, 40, generator=g, requires_grad=True)
    W = torch.randn(24, 40, generator=g, requires_grad=True)
    b = torch.randn(24, generator=g, requires_grad=True)
    R = torch.randn(3, 7, 24, generator=g)
    (F.linear(x, W, b) * R).sum().backward()
    xd, Wd, bd = (t.detach().cuda().requires_grad_(True) for t in (x, W, b))
    y = blocks.linear(xd, Wd, bd)
    assert maxabs(y, F.linear(x, W, b)) < 1e-4
    (y * R.cuda()).sum().backward()
    assert rel(xd.grad, x.grad) < 1e-4 and rel(Wd.grad, W.grad) < 1e-4 and rel(bd.grad, b.grad) < 1e-4
    ids = torch.randint(0, 11, (4, 9), generator=g)
    tab = torch.randn(11, 16, generator=g, requires_grad=True)
    R2 = torch.randn(4, 9, 16, generator=g)
    (F.embedding(ids, tab) * R2).sum().backward()
    tabd = tab.detach().cuda().requires_grad_(True)
    e = blocks.embedding(ids.cuda(), tabd)
    assert maxabs(e, F.embedding(ids, tab)) == 0.0
    (e * R2.cuda()).sum().backward()
    assert rel(tabd.grad, tab.grad) < 1e-5


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("kind", ["postnet", "encoder"])
def test_conv_bn_stack_vs_oracle(env, kind, training):
    L, blocks, ops = env
    g = torch.Generator().manual_seed(3)
    B, T = 3, 17
    if kind == "postnet":
        chans, acts, tacts, site0, res = [8, 32, 32, 8], [2, 2, 0], [torch.tanh, torch.tanh, None], L.SITE["POSTNET0"], True
    else:
        chans, acts, tacts, site0, res = [32, 32, 32], [1, 1], [F.relu, F.relu], L.SITE["ENC0"], False
    n = len(acts)
    convs = [torch.nn.Conv1d(chans[i], chans[i + 1], 5, padding=2) for i in range(n)]
    bns = [torch.nn.BatchNorm1d(chans[i + 1]) for i in range(n)]
    for bn in bns:
        bn.weight.data.uniform_(0.5, 1.5, generator=g); bn.bias.data.uniform_(-0.2, 0.2, generator=g)
        bn.running_mean.uniform_(-0.1, 0.1, generator=g); bn.running_var.uniform_(0.5, 1.5, generator=g)
    x = torch.randn(B, chans[0], T, generator=g, requires_grad=True)             # reference layout [B,C,T]
    seed = 4242
    keep = [ops.rng_keep_mask(seed, site0 + i, B * T * chans[i + 1], 0.5).view(B, T, chans[i + 1]).float().cpu().permute(0, 2, 1)
            for i in range(n)] if training else [None] * n
    # oracle: the same op sequence as model.py:65-70 / :98-99 with replayed masks
    P, stats = {}, {}
    for i in range(n):
        P[f"s.{i}.0.conv.weight"], P[f"s.{i}.0.conv.bias"] = convs[i].weight, convs[i].bias
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            P[f"s.{i}.1.{k}"] = getattr(bns[i], k)
    h = x
    for i in range(n):
        h = O.conv_bn(h, P, f"s.{i}.0", f"s.{i}.1", training, stats)
        if tacts[i] is not None:
            h = tacts[i](h)
        h = O._drop(h, keep[i], 0.5)
    if res:
        h = h + x
    R = torch.randn(h.shape, generator=g)
    (h * R).sum().backward()
    # HIP
    import copy
    dconvs = [copy.deepcopy(c).cuda() for c in convs]
    dbns = [copy.deepcopy(b).cuda() for b in bns]
    for m_ in dconvs + dbns:
        m_.zero_grad()
    xd = x.detach().permute(0, 2, 1).contiguous().cuda().requires_grad_(True)   # [B,T,C]
    y = blocks.conv_bn_stack(xd, list(zip(dconvs, dbns)), acts, training=training, drop_p=0.5, seed=seed, site0=site0, residual=res)
    assert maxabs(y.permute(0, 2, 1), h) < 2e-4
    (y * R.permute(0, 2, 1).contiguous().cuda()).sum().backward()
    assert rel(xd.grad.permute(0, 2, 1), x.grad) < 5e-4
    for i in range(n):
        assert rel(dconvs[i].weight.grad, convs[i].weight.grad) < 5e-4, i
        assert rel(dconvs[i].bias.grad, convs[i].bias.grad) < 5e-4 or float(convs[i].bias.grad.abs().max()) < 1e-4, i
        assert rel(dbns[i].weight.grad, bns[i].weight.grad) < 5e-4, i
        assert rel(dbns[i].bias.grad, bns[i].bias.grad) < 5e-4, i
        if training:
            assert maxabs(dbns[i].running_mean, stats[f"s.{i}.1.running_mean"]) < 1e-5
            assert maxabs(dbns[i].running_var, stats[f"s.{i}.1.running_var"]) < 1e-5
            assert int(dbns[i].num_batches_tracked) == int(bns[i].num_batches_tracked) + 1


@pytest.mark.parametrize("training", [False, True])
def test_conv_bn_stack_bf16_staged_vs_converting(env, training):
    """bf16 mode, whole-tile shapes: the conv GEMMs (forward, d(input), d(weight)) on staged bf16 operands (frames staged
    once, taps as row shifts; im2col transpose written out for d(weight)) against the same GEMMs on the converting
    kernel, utterance edges included.  Both round the same fp32 operands to bf16, but their K splits differ (the 256-tile
    kernel keeps K whole where the converting kernel splits it), so layer 1 agrees to summation order (1e-6) and a few
    elements of everything computed FROM it then round to the neighbouring bf16 value: 1e-3 of the largest element bounds
    that; the independent check is test_conv_bf16_implicit_gemm_vs_fp64_on_rounded_operands below."""
    L, blocks, ops = env
    g = torch.Generator().manual_seed(11)
    B, T = 2, 128
    chans, acts = [128, 256, 256], [1, 2]
    convs = [torch.nn.Conv1d(chans[i], chans[i + 1], 5, padding=2).cuda() for i in range(2)]
    bns = [torch.nn.BatchNorm1d(chans[i + 1]).cuda() for i in range(2)]
    x0 = torch.randn(B, T, chans[0], generator=g).cuda()
    R = torch.randn(B, T, chans[-1], generator=g).cuda()
    res = {}
    L.set_precision("bf16")
    try:
        for on in (1, 0):
            L.check(L.lib().t2_set_gemm_staging(on))
            for m_ in convs + bns:
                m_.zero_grad()
            for bn in bns:
                bn.running_mean.zero_(); bn.running_var.fill_(1.0)
            xd = x0.clone().requires_grad_(True)
            y = blocks.conv_bn_stack(xd, list(zip(convs, bns)), acts, training=training, drop_p=0.5, seed=77, site0=L.SITE["ENC0"])
            (y * R).sum().backward()
            res[on] = [y.detach().clone(), xd.grad.clone()] + [c.weight.grad.clone() for c in convs] + [c.bias.grad.clone() for c in convs]
    finally:
        L.check(L.lib().t2_set_gemm_staging(1))
        L.set_precision("f32")
    for a, b in zip(res[1], res[0]):
        assert torch.isfinite(a).all()
        if float(b.abs().max()) < 1e-4:                              # d(conv bias) in front of a training-mode BatchNorm: exactly 0, rounding noise
            assert float((a - b).abs().max()) < 1e-4
        else:
            assert rel(a, b) < 1e-3


@pytest.mark.parametrize("B,T,Cin,Cout", [(2, 128, 128, 256), (3, 256, 256, 512)])
def test_conv_bf16_implicit_gemm_vs_fp64_on_rounded_operands(env, B, T, Cin, Cout):
    """The bf16-source implicit-conv path (frames staged once as bf16, taps as row shifts, im2col transpose written out
    for d(weight)) against an INDEPENDENT reference: F.conv1d in fp64 on bf16-rounded x, w (forward) and bf16-rounded dz
    (both backward products) — the yardstick test_gemm_bf16_staged_operands uses for the plain GEMM.  Whole-tile shape
    (B*T = 256 rows, 128 -> 256 channels, k = 5), utterance edges checked on their own.  Reference semantics: Conv1d of
    model.py:34-70 via layers.py:21-39.  BatchNorm in eval mode with unit statistics is the identity, so y = conv + bias."""
    import torch.nn.functional as F
    L, blocks, ops = env
    g = torch.Generator().manual_seed(23)
    conv = torch.nn.Conv1d(Cin, Cout, 5, padding=2).cuda()
    bn = torch.nn.BatchNorm1d(Cout).cuda().eval()
    bn.running_mean.zero_(); bn.running_var.fill_(1.0 - bn.eps)
    x0 = torch.randn(B, T, Cin, generator=g).cuda()
    R = torch.randn(B, T, Cout, generator=g).cuda()
    L.set_precision("bf16")
    try:
        xd = x0.clone().requires_grad_(True)
        y = blocks.conv_bn_stack(xd, [(conv, bn)], [blocks.ACT_NONE], training=False, drop_p=0.0, seed=1, site0=L.SITE["ENC0"])
        (y * R).sum().backward()
        torch.cuda.synchronize()
    finally:
        L.set_precision("f32")
    rb = lambda t: t.detach().bfloat16().double().cpu()
    xb = rb(x0).permute(0, 2, 1).requires_grad_(True)                    # [B,Cin,T], bf16-rounded values in fp64
    wb = rb(conv.weight).requires_grad_(True)
    z = F.conv1d(xb, wb, conv.bias.detach().double().cpu(), padding=2)    # [B,Cout,T]
    inv = 1.0 / torch.sqrt(bn.running_var.double().cpu() + bn.eps)
    y_ref = (z * inv[None, :, None]).permute(0, 2, 1)
    K = 5 * Cin
    tol = 2e-3 * max(1.0, K ** 0.5 / 8)
    err = (y.double().cpu() - y_ref).abs()
    edges = torch.tensor([0, 1, T - 2, T - 1])
    assert float(err.max()) < tol and float(err[:, edges].max()) < tol, (float(err.max()), float(err[:, edges].max()))
    # backward: both products read bf16(dz); dz = dy * gamma * invstd (eval-mode BatchNorm)
    dz = rb(R * (bn.weight.detach() * (1.0 / torch.sqrt(bn.running_var + bn.eps)))[None, None, :]).permute(0, 2, 1)
    gx, gw = torch.autograd.grad(z, (xb, wb), dz)
    ex = (xd.grad.double().cpu() - gx.permute(0, 2, 1)).abs()
    assert float(ex.max()) < 2e-3 * max(1.0, (5 * Cout) ** 0.5 / 8) and float(ex[:, edges].max()) < 2e-3 * max(1.0, (5 * Cout) ** 0.5 / 8)
    ew = (conv.weight.grad.double().cpu() - gw).abs()
    assert float(ew.max()) < 2e-3 * max(1.0, (B * T) ** 0.5 / 8), float(ew.max())
    # the edge taps of d(weight) see the zero padding: a tap-wise check that no frame of the neighbouring utterance leaks in
    for tap in (0, 4):
        assert float(ew[:, :, tap].max()) < 2e-3 * max(1.0, (B * T) ** 0.5 / 8)


@pytest.mark.parametrize("packed", [True, False])
def test_bilstm_vs_oracle(env, packed):
    L, blocks, ops = env
    g = torch.Generator().manual_seed(5)
    B, T, E = 5, 11, 128                      # H = 64: the LSTM kernels need multiples of 64
    lstm = torch.nn.LSTM(E, E // 2, 1, batch_first=True, bidirectional=True)
    x = torch.randn(B, T, E, generator=g, requires_grad=True)
    lengths = torch.tensor([11, 9, 9, 4, 1]) if packed else None
    P = {"l." + k: v for k, v in lstm.named_parameters()}
    out = O.bilstm(x, lengths, P, "l")
    R = torch.randn(out.shape, generator=g)
    (out * R).sum().backward()
    import copy
    dl = copy.deepcopy(lstm).cuda()
    dl.zero_grad()
    xd = x.detach().cuda().requires_grad_(True)
    y = blocks.bilstm(xd, None if lengths is None else lengths.cuda(), dl)
    assert maxabs(y, out) < 1e-4
    (y * R.cuda()).sum().backward()
    assert rel(xd.grad, x.grad) < 3e-4
    for (k, p), (_, q) in zip(lstm.named_parameters(), dl.named_parameters()):
        assert rel(q.grad, p.grad) < 3e-4, k


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("B,T", [(5, 11), (64, 23), (33, 40), (128, 9)])
def test_bilstm_persistent_chain_vs_oracle(env, B, T, packed, mode):
    """The encoder BiLSTM at the reference's size (H = 256, model.py:93-112) runs ALL its steps, both directions, in one
    persistent launch forward and one backward (csrc/chain_enc.hip, exact fp32 in both precision modes; the hoisted input
    GEMM follows the mode).  Against the oracle's nn.LSTM restatement and against the per-step launch path."""
    L, blocks, ops = env
    if not L.lib().t2_chain_claimed():
        pytest.skip("another process holds this GPU's persistent-kernel claim")
    g = torch.Generator().manual_seed(B * 100 + T)
    E = 512
    lstm = torch.nn.LSTM(E, E // 2, 1, batch_first=True, bidirectional=True)
    x = torch.randn(B, T, E, generator=g, requires_grad=True)
    lengths = None
    if packed:
        lengths = torch.randint(1, T + 1, (B,), generator=g).sort(descending=True).values
        lengths[0] = T
    P = {"l." + k: v for k, v in lstm.named_parameters()}
    out = O.bilstm(x, lengths, P, "l")
    R = torch.randn(out.shape, generator=g)
    (out * R).sum().backward()
    import copy
    res = {}
    L.set_precision(mode)
    try:
        for chain in (True, False):
            L.set_chain(chain)
            dl = copy.deepcopy(lstm).cuda()
            dl.zero_grad()
            xd = x.detach().cuda().requires_grad_(True)
            y = blocks.bilstm(xd, None if lengths is None else lengths.cuda(), dl)
            (y * R.cuda()).sum().backward()
            torch.cuda.synchronize()
            res[chain] = (y.detach(), xd.grad, {k: q.grad for k, q in dl.named_parameters()})
    finally:
        L.set_precision("f32"); L.set_chain(True)
    ops.check_chain_status(block=True)
    tol_y, tol_g = (1e-4, 3e-4) if mode == "f32" else (3e-2, 3e-2)     # bf16 mode: the input GEMM's operands are bf16
    y, dx, gw = res[True]
    assert maxabs(y, out) < tol_y
    assert rel(dx, x.grad) < tol_g
    for k, p in lstm.named_parameters():
        assert rel(gw[k], p.grad) < tol_g, k
    # chain vs launches: the same arithmetic up to summation order (and the accurate exp in both)
    y0, dx0, gw0 = res[False]
    tol_c = 1e-4 if mode == "f32" else 2e-3       # (bf16 mode: the GEMMs behind dpre round their operands: 1e-7 apart can become one bf16 ulp)
    assert maxabs(y, y0) < 2e-5 and rel(dx, dx0) < tol_c
    for k in gw:
        assert rel(gw[k], gw0[k]) < tol_c, k
    if packed:                                                        # frames past an item's length are exactly zero
        for b in range(B):
            assert not y[b, int(lengths[b]):].any()


def test_fused_adam_matches_clip_plus_torch_adam():
    """optim.FusedAdam.step(max_norm) vs torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step over odd-sized tensors,
    several steps, with and without clipping being active; state_dicts interchange."""
    from tacotron2_subword_amd.optim import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [(4096, 768), (1, 128), (8193,), (33, 7, 5), (80,), (3,)]
    ref_p = [torch.randn(s, device="cuda", generator=g).requires_grad_(True) for s in shapes]
    fus_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref_p.append(torch.zeros(5, device="cuda", requires_grad=True)); fus_p.append(torch.zeros(5, device="cuda", requires_grad=True))   # never gets a grad
    ref = torch.optim.Adam(ref_p, lr=1e-3, weight_decay=1e-6)
    fus = FusedAdam(fus_p, lr=1e-3, weight_decay=1e-6)
    for step in range(4):
        scale = 0.01 if step == 2 else 1.0                          # step 2: norm below the threshold -> coefficient 1
        for a, b in zip(ref_p[:-1], fus_p[:-1]):
            grad = torch.randn(a.shape, device="cuda", generator=g) * scale
            a.grad, b.grad = grad.clone(), grad.clone()
        want_norm = torch.nn.utils.clip_grad_norm_(ref_p, 1.0)
        ref.step()
        got_norm = fus.step(max_norm=1.0)
        assert abs(float(got_norm) - float(want_norm)) < 1e-4 * float(want_norm)
        for a, b in zip(ref_p, fus_p):
            assert float((a - b).abs().max()) < 2e-6, step
    sd = fus.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0
    ref2 = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in fus_p], lr=1e-3, weight_decay=1e-6)
    ref2.load_state_dict(sd)                                        # FusedAdam state loads into torch.optim.Adam
    assert fus_p[-1].grad is None and len(fus.state[fus_p[-1]]) == 0
