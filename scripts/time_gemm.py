"""Dev tool: TFLOP/s of the GEMM kernels on the shapes of one training iteration (fp32 vs bf16 operands)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_subword_amd import _lib as L, ops
shapes = [("PRED fwd NT", 25600, 4096, 3072, False, True), ("dDIN NN", 25600, 3072, 4096, False, False),
          ("dW_dec_ih TN", 4096, 3072, 25600, True, False), ("dW_att_hh TN", 4096, 1024, 25600, True, False),
          ("PREA fwd NT", 25600, 4096, 256, False, True), ("postnet conv NT-like", 25600, 512, 2560, False, True),
          ("dWq TN", 128, 1024, 25600, True, False),
          ("PRED chunk NT", 3200, 4096, 3072, False, True), ("dDIN chunk NN", 3200, 3072, 4096, False, False),
          ("dW_att_ctx TN", 4096, 512, 25600, True, False)]
ws = torch.empty(160 << 20, device="cuda")
for name, M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    for mode in ("f32", "bf16"):
        L.set_precision(mode)
        for _ in range(2):
            ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out, ws=ws)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            ops.gemm(A, B, trans_a=ta, trans_b=tb, out=out, ws=ws)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        if mode == "f32":
            ref = out.clone()
        else:
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            assert err < 2e-2, (name, err)
        print(f"{name:24s} {mode:5s} M={M} N={N} K={K}: {dt*1e3:7.3f} ms  {2*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
L.set_precision("f32")
