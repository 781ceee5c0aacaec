"""Dev tool: time the HIP decoder forward / backward at a given shape (random weights)."""
import argparse
import sys
import os
import time

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
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--bwd", type=int, default=1)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--overlap", type=int, default=1)
ap.add_argument("--prof", type=int, default=1, help="print per-kernel HIP-event averages of one extra pass")
ap.add_argument("--att", default="sma", help="sma | lsa")
a = ap.parse_args()
L.set_precision(a.dtype)
L.check(L.lib().t2_set_overlap(a.overlap))
hp, P, dims, W = decoder_setup(SMA if a.att == "sma" else LSA)
g = torch.Generator(device="cuda").manual_seed(1)
mem = torch.randn(a.B, a.Tin, 512, device="cuda", generator=g) * 0.5
mems = torch.randn(a.B, a.Tsub, 512, device="cuda", generator=g) * 0.5
mels = torch.randn(a.B, 80, a.T, device="cuda", generator=g)
tl = torch.full((a.B,), a.Tin, device="cuda")
bl = torch.full((a.B,), a.Tsub, device="cuda")
dmel = torch.randn(a.B, a.T, 80, device="cuda", generator=g)
dgate = torch.randn(a.B, a.T, device="cuda", generator=g)
for it in range(a.iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dp = ops.decoder_forward(W, dims, mem, mems, tl, bl, mels, training=True, prenet_dropout=True, seed=it)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if a.bwd:
        G, dm, dms = ops.decoder_backward(W, P, dims, dp, mem, mems, dmel, dgate, training=True, prenet_dropout=True, seed=it)
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iter {it}: fwd {1e3*(t1-t0):.2f} ms  bwd {1e3*(t2-t1):.2f} ms  frames/s fwd+bwd {a.B*a.T/(t2-t0):.0f}", flush=True)
    del dp
print("finite:", bool(torch.isfinite(dm).all()) if a.bwd else True)
if a.prof:
    L.prof_enable(8 * a.T + 64)
    dp = ops.decoder_forward(W, dims, mem, mems, tl, bl, mels, training=True, prenet_dropout=True, seed=99)
    if a.bwd:
        ops.decoder_backward(W, P, dims, dp, mem, mems, dmel, dgate, training=True, prenet_dropout=True, seed=99)
    for k, (ms, n) in L.prof_collect().items():
        if n:
            print(f"  {k:26s} {n:5d} launches  avg {1e3 * ms / n:7.2f} us  total {ms:7.2f} ms")
