"""Dev tool: the ragged / small products of one training iteration (the ones the dispatcher routes to the converting kernels,
profiles/r03_gemm_inventory.txt) timed in isolation in bf16 mode, against a memory floor (operands + result once at 4 TB/s) and the
matrix floor at 1 PFLOP/s: which of them are worth a kernel of their own."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_subword_amd import _lib as L, ops
# (name, M, N, K, trans_a (A stored [K][M]), trans_b (B stored [N][K]))
shapes = [("prenet1 fwd NT", 25600, 256, 80, False, True), ("prenet2 fwd NT", 25600, 256, 256, False, True),
          ("prenet dX NN", 25600, 256, 256, False, False), ("mel proj fwd NT", 25600, 80, 2048, False, True),
          ("mel proj dX NN", 25600, 2048, 80, False, False), ("prenet1 dW TN", 256, 80, 25600, True, False),
          ("dWq TN", 128, 1024, 25600, True, False), ("proj dW TN", 80, 2048, 25600, True, False),
          ("pm NT", 6400, 128, 512, False, True), ("dpm.Wm NN", 6400, 512, 128, False, False), ("dWm TN", 128, 512, 6400, True, False)]
ws = torch.empty(160 << 20, device="cuda")
L.set_precision("bf16")
tot = 0.0
for name, M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out, ws=ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out, ws=ws)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    mem = 4.0 * (M * K + N * K + M * N) / 4e12
    mat = 2.0 * M * N * K / 1e15
    tot += dt
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:6d}: {dt*1e6:7.1f} us   floors: memory {mem*1e6:6.1f} us, matrix {mat*1e6:6.1f} us   x{dt/max(mem, mat):5.1f}", flush=True)
print(f"sum {tot*1e3:.3f} ms")
L.set_precision("f32")
