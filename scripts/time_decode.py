"""Dev tool: decoder steps per second of the decode loop at a given batch / polling interval (random weights)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import SMA, LSA, decoder_setup
from tacotron2_subword_amd import _lib as L, ops
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--steps", type=int, default=1024)
ap.add_argument("--polls", default="16,32,64,128")
ap.add_argument("--att", default="sma")
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
L.set_precision(a.dtype)
hp, P, dims, W = decoder_setup(SMA if a.att == "sma" else LSA)
g = torch.Generator(device="cuda").manual_seed(1)
mem = torch.randn(a.B, 100, 512, device="cuda", generator=g) * 0.5
mems = torch.randn(a.B, 60, 512, device="cuda", generator=g) * 0.5
for chain in (True, False):
    L.set_chain(chain)
    for poll in [int(p) for p in a.polls.split(",")]:
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dp, n, stop = ops.decoder_infer(W, dims, mem, mems, max_steps=a.steps, gate_threshold=2.0, prenet_dropout=True, seed=it, poll_every=poll)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"chain={int(chain)} poll={poll:4d}: {1e6 * dt / a.steps:7.2f} us/step  {a.steps / dt:9.0f} steps/s  status {dp.chain_status()}", flush=True)
