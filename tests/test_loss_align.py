"""Alignment-guide branches of Tacotron2Loss and Alignment_Generator against vectors recorded
from the reference's own loss_function.py / utils.py (tests/golden/make_golden_loss.py): the oracle restatement and
the host module (plain torch reductions, so the comparison runs on the CPU)."""
import numpy as np
import pytest
import torch

from oracle import tacotron2_oracle as O

from helpers import load_golden


def _case(g):
    t = lambda k: torch.from_numpy(np.asarray(g[k]))
    x = (None, t("tl"), t("tl"), t("mel_t"), tuple(int(v) for v in g["max_lens"]), t("ol"), None, None, None)
    y = (t("mel_t"), t("gate_t"), t("align_t"))
    return t, x, y


@pytest.mark.parametrize("mode", ["", "L2", "KL"])
@pytest.mark.parametrize("iters", [0, 50000])
@pytest.mark.parametrize("impl", ["oracle", "module"])
def test_loss_branches_vs_reference(mode, iters, impl):
    g = load_golden("loss_align")
    t, x, y = _case(g)
    leaves = [t(k).clone().requires_grad_(True) for k in ("mel_o", "post_o", "gate_o", "al", "alb")]
    outs = [l * 1.0 for l in leaves]
    if impl == "oracle":
        res = O.loss_align(outs, y, x, mode, iters)
    else:
        from tacotron2_subword_amd.loss_function import Tacotron2Loss
        res = Tacotron2Loss(mode)(outs, y, x, iters)
    res[0].backward()
    tag = f"{mode or 'none'}_{iters}"
    want = g[f"loss_{tag}"]
    for r, w in zip(res, want):
        if np.isnan(w):
            assert r is None
        else:
            assert abs(float(r) - float(w)) < 1e-5 * max(1.0, abs(float(w)))
    for name, l in zip(("mel_o", "post_o", "gate_o", "al", "alb"), leaves):
        ref = torch.from_numpy(g[f"grad_{tag}_{name}"])
        got = l.grad if l.grad is not None else torch.zeros_like(l)
        assert float((got - ref).abs().max()) < 1e-6 + 1e-5 * float(ref.abs().max()), name
    # the model outputs handed in are left untouched (the reference edits them in place; documented difference)
    assert float((outs[3].detach() == 0).sum()) > 0


def test_kl_branch_indexes_the_max_len_pair_like_the_reference():
    from tacotron2_subword_amd.loss_function import Tacotron2Loss
    g = load_golden("loss_align")
    t, x, y = _case(g)
    rep = lambda v: torch.cat([v, v[:1]], 0)                 # B = 3: x[4] has only two entries
    outs = [rep(t(k)) for k in ("mel_o", "post_o", "gate_o", "al", "alb")]
    y3 = tuple(rep(v) for v in y)
    x3 = (None, rep(x[1]), rep(x[2]), rep(x[3]), x[4], rep(x[5]), None, None, None)
    with pytest.raises(IndexError):
        Tacotron2Loss("KL")(outs, y3, x3, 0)


def test_alignment_generator_vs_reference():
    from tacotron2_subword_amd.utils import Alignment_Generator
    g = load_golden("loss_align")
    dur = torch.from_numpy(g["dur"])
    a = Alignment_Generator()(dur)
    assert a.shape == g["align_t"].shape and torch.equal(a, torch.from_numpy(g["align_t"]))
    ragged = torch.tensor([[3, 0, 2], [1, 1, 1]])            # a zero-length phone, items of different total length
    b = Alignment_Generator()(ragged)
    want = torch.zeros(2, 5, 3)
    want[0, 0:3, 0] = 1; want[0, 3:5, 2] = 1; want[1, 0, 0] = 1; want[1, 1, 1] = 1; want[1, 2, 2] = 1
    assert torch.equal(b, want)
