#!/usr/bin/env python3
"""Golden vectors for the alignment-guide branches of Tacotron2Loss (loss_function.py:24-66) and for
Alignment_Generator / create_alignment (utils.py:92-117), recorded from the reference's own modules (imported with
the harness shims of make_golden.py).  Run in the build container only; writes loss_align.npz next to this file."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    ref_model, ref_attention, ref_hparams, ref_loss = import_reference()
    import utils as ref_utils  # noqa  (the reference's utils.py: /root/reference is on sys.path now)
    g = torch.Generator().manual_seed(2024)
    B, T, Tin, M = 2, 9, 7, 80                      # B = 2: the KL branch indexes x[4] = (max_in, max_out) by batch item
    mel_t = torch.randn(B, M, T, generator=g)
    gate_t = (torch.rand(B, T, generator=g) > 0.7).float()
    dur = torch.tensor([[2, 1, 1, 2, 1, 1, 1], [1, 1, 3, 1, 1, 1, 1]])
    align_t = ref_utils.Alignment_Generator()(dur)                      # [B, 9, 7] hard alignment
    assert tuple(align_t.shape) == (B, T, Tin)
    mel_o, post_o = torch.randn(B, M, T, generator=g), torch.randn(B, M, T, generator=g)
    gate_o = torch.randn(B, T, generator=g)
    al = torch.softmax(torch.randn(B, T, Tin, generator=g), -1)
    alb = torch.softmax(torch.randn(B, T, Tin, generator=g), -1)
    al[:, :, -1] = 0.0                                                   # exact zeros: the KL branch replaces them by 1e-6
    alb[0, 2, 3] = 0.0
    tl, ol = torch.tensor([7, 6]), torch.tensor([9, 8])
    x = (None, tl, tl, mel_t, (7, 9), ol, None, None, None)
    out = dict(mel_t=mel_t, gate_t=gate_t, dur=dur, align_t=align_t, mel_o=mel_o, post_o=post_o, gate_o=gate_o, al=al, alb=alb,
               tl=tl, ol=ol, max_lens=np.array([7, 9]))
    for mode in ("", "L2", "KL"):
        for iters in (0, 50000):
            leaves = [t.clone().requires_grad_(True) for t in (mel_o, post_o, gate_o, al, alb)]
            # the reference edits its inputs in place (KL branch): hand it non-leaf copies, like real model outputs
            outs = [l * 1.0 for l in leaves]
            res = ref_loss.Tacotron2Loss(mode)(outs, (mel_t.clone(), gate_t.clone(), align_t.clone()), x, iters)
            res[0].backward()
            tag = f"{mode or 'none'}_{iters}"
            out[f"loss_{tag}"] = np.array([float(r) if r is not None else np.nan for r in res])
            for name, l in zip(("mel_o", "post_o", "gate_o", "al", "alb"), leaves):
                out[f"grad_{tag}_{name}"] = (l.grad if l.grad is not None else torch.zeros_like(l)).numpy()
    np.savez_compressed(os.path.join(HERE, "loss_align.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print("wrote loss_align.npz", {k: out[k] for k in out if k.startswith("loss_")})


if __name__ == "__main__":
    main()
