"""Dev tool: the 256x256 LDS-DMA GEMM against the 128x128 register-staged one (same bf16 operands), repeated for races.
Run twice:  T2_GEMM_256=0 python scripts/check_gemm256.py --save /tmp/g.pt ;  python scripts/check_gemm256.py --load /tmp/g.pt"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_subword_amd import _lib as L, ops
ap = argparse.ArgumentParser(); ap.add_argument("--save"); ap.add_argument("--load"); ap.add_argument("--reps", type=int, default=6); ap.add_argument("--notest", action="store_true"); ap.add_argument("--only")
a = ap.parse_args()
shapes = [("tiny", 256, 256, 128, False, True), ("k6", 512, 256, 384, False, True), ("wide", 256, 1024, 256, False, False),
          ("PRED NT", 25600, 4096, 3072, False, True), ("dDIN NN", 25600, 3072, 4096, False, False),
          ("dW_dec TN", 4096, 3072, 25600, True, False), ("dW_hh TN", 4096, 1024, 25600, True, False), ("dW_ctx TN", 4096, 512, 25600, True, False),
          ("chunk NT", 3328, 4096, 3072, False, True), ("sq4096 NT", 4096, 4096, 4096, False, True), ("sq8192 NT", 8192, 8192, 8192, False, True)]
ws = torch.empty(400 << 20, device="cuda")
L.set_precision("bf16")
ref = torch.load(a.load) if a.load else {}
out_all = {}
g = torch.Generator(device="cuda").manual_seed(3)
for name, M, N, K, ta, tb in shapes:
    if a.only and a.only not in name: continue
    A = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g)
    B = torch.randn((N, K) if tb else (K, N), device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda")
    first = None
    for rep in range(a.reps):
        out.zero_()
        ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out, ws=ws)
        torch.cuda.synchronize()
        if first is None: first = out.clone()
        elif not a.notest: assert torch.equal(first, out), (name, rep, (first - out).abs().max().item())
    t0 = time.perf_counter()
    for _ in range(5): ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out, ws=ws)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    msg = ""
    if a.load:
        r = ref[name].cuda()
        f2 = first if r.shape == first.shape else first[:2048]
        err = ((f2 - r).abs().max() / r.abs().max()).item()
        msg = f" rel err vs 128-tile {err:.2e}"
        assert err < 2e-5, (name, err)
    if a.notest:
        print(f"{name:10s} {dt*1e3:7.3f} ms {2*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True); continue
    # sampled fp64 check on bf16-rounded operands
    Ar = (A.t() if ta else A).bfloat16().double(); Br = (B if tb else B.t()).bfloat16().double()
    rows = torch.randint(0, M, (64,), device="cuda", generator=g)
    want = Ar[rows] @ Br.t()
    e64 = ((first[rows].double() - want).abs().max() / want.abs().max()).item()
    assert e64 < 1e-5, (name, e64)
    print(f"{name:10s} M={M} N={N} K={K}: {dt*1e3:7.3f} ms {2*M*N*K/dt/1e12:7.1f} TFLOP/s (incl. staging casts){msg} fp64 {e64:.1e}", flush=True)
    out_all[name] = first.cpu() if M * N <= 4096 * 4096 else first[:2048].cpu()
if a.save:
    torch.save(out_all, a.save)
