"""Dev tool: where a step of the persistent chain kernels spends its time (diagnostic build only).
  T2_EXTRA_HIPCC_FLAGS=-DT2_STAMPS=1 python -c "from tacotron2_subword_amd import build; build.build(force=True)"
  python scripts/chain_stamps.py [--B 64] [--T 400]
Segments (thread 0 of each workgroup, realtime counter): see the T2_CSTAMP calls in csrc/chain.hip."""
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
ap.add_argument("--Tin", type=int, default=100)
ap.add_argument("--Tsub", type=int, default=60)
ap.add_argument("--att", default="sma")
ap.add_argument("--decode", type=int, default=0, help="decoder steps of an inference() run instead of the teacher-forced pass")
a = ap.parse_args()
L.set_precision("bf16")
hp, P, dims, W = decoder_setup(SMA if a.att == "sma" else LSA)
g = torch.Generator(device="cuda").manual_seed(1)
mem = torch.randn(a.B, a.Tin, 512, device="cuda", generator=g) * 0.5
mems = torch.randn(a.B, a.Tsub, 512, device="cuda", generator=g) * 0.5
mels = torch.randn(a.B, 80, a.T, device="cuda", generator=g)
tl = torch.full((a.B,), a.Tin, device="cuda")
bl = torch.full((a.B,), a.Tsub, device="cuda")
lib = L.lib()
NAMES = ["L wait h", "L h-part GEMM", "L wait ctx", "L reduce + gates", "L h/q stores", "L publish", "A (idle->start)", "A wait h",
         "A query sum", "A energies", "A recurrence", "A context", "A publish", "A saved stores", "L ctx loads + MFMA issue", "L saved stores + pre loads"]
if a.decode:
    NAMES[1], NAMES[6], NAMES[13] = "D wait h / ctx / dec_h", "D loads + MFMA + gates", "(A publish -> D start)"
for it in range(3):
    lib.t2_debug_clear_chain_stamps()
    if a.decode:
        a.T = a.decode
        dp, n, stop = ops.decoder_infer(W, dims, mem, mems, max_steps=a.decode, gate_threshold=2.0, prenet_dropout=True, seed=it)
    else:
        dp = ops.decoder_forward(W, dims, mem, mems, tl, bl, mels, training=True, prenet_dropout=True, seed=it)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 16))()
lib.t2_debug_read_chain_stamps(buf, 256 * 16)
# -DT2_STAMPS=1 records the attention chain, =2 the decoder-LSTM chain (L segments only)
tot = [sum(buf[w * 16 + i] for w in range(256)) / 256 / a.T / 100.0 for i in range(16)]
print("mean over workgroups, us per step:")
for n, v in zip(NAMES, tot):
    print(f"  {n:28s} {v:6.2f}")
print("  sum", round(sum(tot), 2))
if not a.decode:
    # per class of workgroup: the A item's stream (B*CS workgroups per stream; the phone stream has the longer memory)
    half = 128
    cls = {"A item: phone stream": range(0, half), "A item: sub-word stream": range(half, 256)}
    print(f"{'':30s}" + "".join(f"{k:>26s}" for k in cls))
    for i, n in enumerate(NAMES):
        print(f"  {n:28s}" + "".join(f"{sum(buf[w * 16 + i] for w in ws) / len(ws) / a.T / 100.0:26.2f}" for ws in cls.values()))
