"""Experiment (test infrastructure, not collected by pytest): how accurate would a persistent decoder chain be whose
recurrent LSTM products use SPLIT-bf16 operands (x = hi + lo, both bf16; x.w ~ hi.hi + hi.lo + lo.hi, fp32 accumulation on
the bf16 matrix cores) instead of exact fp32?  The oracle's LSTM cells are re-run on the BASELINE-shaped case (B=2, 100 / 60
positions, 400 frames) with their matrix products emulated term by term, and compared with the exact fp32 oracle.
VERDICT r2 item 5: is a parity-grade (1e-4) fast path feasible?   python tests/tools/split_bf16_accuracy.py"""
import os, sys
import torch
import torch.nn.functional as F
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import recipe
from oracle import tacotron2_oracle as O

def split(x, terms):
    parts, r = [], x
    for _ in range(terms):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p); r = r - p
    return parts

def make_linear(nx, nw, keep):
    """x.W^T with x in nx bf16 parts, W in nw parts, keeping the cross terms (i, j) with i + j < keep"""
    def lin(x, w, b=None):
        xs, ws = split(x, nx), split(w, nw)
        acc = None
        for i, xp in enumerate(xs):
            for j, wp in enumerate(ws):
                if i + j < keep:
                    t = xp @ wp.t()
                    acc = t if acc is None else acc + t
        return acc if b is None else acc + b
    return lin

def run(lin, hp, P, x):
    orig = O.lstm_cell
    def cell(xx, h, c, w_ih, w_hh, b_ih, b_hh):
        gates = lin(xx, w_ih, b_ih) + lin(h, w_hh, b_hh)
        i, f, g, o = gates.chunk(4, 1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c2), c2
    if lin is not None:
        O.lstm_cell = cell
    try:
        with torch.no_grad():
            return O.forward(P, hp, x, training=False)
    finally:
        O.lstm_cell = orig

hp = O.default_hparams()
P = recipe.make_weights(hp)
x, y = recipe.parse_batch(recipe.make_batch(hp, 2, 100, 60, 400))
ref = run(None, hp, P, x)
names = ("mel", "mel_postnet", "gate", "align", "align_bert")
for label, lin in (("bf16 operands (1 term: the current bf16 mode's LSTM products)", make_linear(1, 1, 1)),
                   ("split-bf16, 3 terms (hi.hi + hi.lo + lo.hi)", make_linear(2, 2, 2)),
                   ("split-bf16, 4 terms (+ lo.lo)", make_linear(2, 2, 3)),
                   ("3-way split, 6 terms (hh hm mh hl lh mm)", make_linear(3, 3, 3))):
    out = run(lin, hp, P, x)
    errs = {n: float((a - b).abs().max()) for n, a, b in zip(names, out, ref)}
    print(f"{label:70s} " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()))
