"""Dev tool: where a step of the persistent attention-chain backward kernel spends its time (diagnostic build only).
  T2_EXTRA_HIPCC_FLAGS=-DT2_STAMPS=1 python -c "from tacotron2_subword_amd import build; build.build(force=True)"
  python scripts/chain_bwd_stamps.py [--B 64] [--T 400]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import SMA, LSA, decoder_setup
from tacotron2_subword_amd import _lib as L, ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--T", type=int, default=400)
ap.add_argument("--att", default="sma")
a = ap.parse_args()
L.set_precision("bf16")
hp, P, dims, W = decoder_setup(SMA if a.att == "sma" else LSA)
g = torch.Generator(device="cuda").manual_seed(1)
mem = torch.randn(a.B, 100, 512, device="cuda", generator=g) * 0.5
mems = torch.randn(a.B, 60, 512, device="cuda", generator=g) * 0.5
mels = torch.randn(a.B, 80, a.T, device="cuda", generator=g)
tl = torch.full((a.B,), 100, device="cuda")
bl = torch.full((a.B,), 60, device="cuda")
dmel = torch.randn(a.B, a.T, 80, device="cuda", generator=g)
dgate = torch.randn(a.B, a.T, device="cuda", generator=g)
lib = L.lib()
NAMES = ["A wait ctx partials", "A dctx + operands", "A g / recurrence", "A energies bwd + dq", "A publish", "P wait dq / h partials",
         "P dq.Wq + gates + frags", "P publish", "G wait dg", "G loads + MFMA + reduce + stores", "G publish", "LSA: halo rows + operands -> LDS", "LSA: loc conv | dWc | carried gradients", "LSA: pa + tanh tile", "-",
         "(G publish -> next A start: stores, prefetch)"]
for it in range(2):
    dp = ops.decoder_forward(W, dims, mem, mems, tl, bl, mels, training=True, prenet_dropout=True, seed=it)
    lib.t2_debug_clear_chain_bwd_stamps()
    ops.decoder_backward(W, P, dims, dp, mem, mems, dmel, dgate, training=True, prenet_dropout=True, seed=it)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 16))()
lib.t2_debug_read_chain_bwd_stamps(buf, 256 * 16)
tot = [sum(buf[w * 16 + i] for w in range(256)) / 256 / a.T / 100.0 for i in range(16)]
print("attention-chain backward, mean over workgroups, us per step:")
for n, v in zip(NAMES, tot):
    if n != "-":
        print(f"  {n:44s} {v:6.2f}")
print("  sum", round(sum(tot), 2))
# per class of workgroup (the G item decides: context-column tiles, h-column tiles, no G item): who waits where
cls = {"G ctx tiles": [w for w in range(256) if w < 192 and (w % 96) % 24 < 8], "G h tiles": [w for w in range(256) if w < 192 and (w % 96) % 24 >= 8],
       "no G item": list(range(192, 256))}
print(f"{'':46s}" + "".join(f"{k:>14s}" for k in cls))
for i, n in enumerate(NAMES):
    if n != "-":
        print(f"  {n:44s}" + "".join(f"{sum(buf[w * 16 + i] for w in ws) / len(ws) / a.T / 100.0:14.2f}" for ws in cls.values()))
