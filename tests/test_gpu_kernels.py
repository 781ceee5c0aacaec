"""GPU parity tests of the individual HIP kernels (GEMM, RNG) through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tacotron2_subword_amd import ops as _ops
    return _ops


def _ref(A, B, ta, tb):
    a = A.double().cpu()
    b = B.double().cpu()
    a = a.t() if ta else a
    b = b.t() if tb else b
    return a @ b


@pytest.mark.parametrize("M,N,K", [(64, 128, 1024), (200, 80, 2048), (37, 1, 2048), (256, 512, 80), (130, 257, 100),
                                   (1, 4096, 768), (2560, 512, 400)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, True), (True, False)])
def test_gemm_layouts(ops, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
    B = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
    C = ops.gemm(A, B, trans_a=ta, trans_b=tb)
    ref = _ref(A, B, ta, tb)
    err = (C.double().cpu() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, K ** 0.5 / 8), err      # fp32 fma chain vs fp64


def test_gemm_epilogue_and_splitk(ops):
    g = torch.Generator().manual_seed(5)
    A = torch.randn(64, 4096, generator=g).cuda()
    B = torch.randn(128, 4096, generator=g).cuda()
    bias = torch.randn(128, generator=g).cuda()
    C0 = torch.randn(64, 128, generator=g).cuda()
    ref = torch.relu(0.5 * (A.double().cpu() @ B.double().cpu().t()) + bias.double().cpu())
    out = ops.gemm(A, B, bias=bias, act=1, alpha=0.5)
    assert (out.double().cpu() - ref).abs().max().item() < 5e-4
    ws = torch.empty(8 * 64 * 128, device="cuda")
    out2 = ops.gemm(A, B, bias=bias, act=1, alpha=0.5, ws=ws, splitk=8)
    assert (out2.double().cpu() - ref).abs().max().item() < 5e-4
    out3 = C0.clone()
    ops.gemm(A, B, alpha=1.0, beta=2.0, out=out3)
    ref3 = A.double().cpu() @ B.double().cpu().t() + 2.0 * C0.double().cpu()
    assert (out3.double().cpu() - ref3).abs().max().item() < 1e-3
    t = ops.gemm(A[:, :256].contiguous(), B[:, :256].contiguous(), act=2)
    assert (t.double().cpu() - torch.tanh(A[:, :256].double().cpu() @ B[:, :256].double().cpu().t())).abs().max().item() < 1e-5


def test_gemm_strided_rows(ops):
    """Rows of A and C taken with a stride (the per-step views of [B,T,W] buffers)."""
    g = torch.Generator().manual_seed(9)
    big = torch.randn(8, 5, 192, generator=g).cuda()          # [B,T,W]
    W = torch.randn(48, 192, generator=g).cuda()
    out = torch.zeros(8, 5, 48).cuda()
    A = big[:, 3, :]                                             # stride 5*192
    Cv = out[:, 3, :]
    from tacotron2_subword_amd import _lib as L
    L.check(L.lib().t2_gemm(A.data_ptr(), W.data_ptr(), Cv.data_ptr(), 8, 48, 192, A.stride(0), 1, W.stride(0), 1,
                            Cv.stride(0), None, 0, 1.0, 0.0, None, 0, 0, L.stream()))
    ref = big[:, 3, :].double().cpu() @ W.double().cpu().t()
    assert (out[:, 3].double().cpu() - ref).abs().max().item() < 1e-4
    assert out[:, 2].abs().max().item() == 0.0


def test_rng_statistics_and_determinism(ops):
    n = 1 << 20
    m1 = ops.rng_keep_mask(1234, 5, n, 0.1)
    m2 = ops.rng_keep_mask(1234, 5, n, 0.1)
    m3 = ops.rng_keep_mask(1234, 6, n, 0.1)
    assert torch.equal(m1, m2)
    assert not torch.equal(m1, m3)
    assert abs(m1.float().mean().item() - 0.9) < 2e-3
    assert abs(ops.rng_keep_mask(99, 1, n, 0.5).float().mean().item() - 0.5) < 3e-3
    z = ops.rng_normal(7, 11, n)
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1.0) < 5e-3
    assert torch.isfinite(z).all()
    zc = z.cpu()
    assert abs(float((zc[:-1] * zc[1:]).mean())) < 5e-3       # neighbouring indices decorrelated


@pytest.mark.parametrize("M,N,K", [(256, 512, 1024), (200, 130, 328), (2560, 512, 400), (4096, 256, 6400)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, True), (True, False)])
def test_gemm_bf16_operands(ops, M, N, K, ta, tb):
    """bf16-operand mode: same layouts, fp32 accumulate; error bounded by bf16 rounding of the operands."""
    from tacotron2_subword_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
    B = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
    ws = torch.empty(16 * M * N, device="cuda")
    L.set_precision("bf16")
    try:
        C = ops.gemm(A, B, trans_a=ta, trans_b=tb)
        C2 = ops.gemm(A, B, trans_a=ta, trans_b=tb, ws=ws, splitk=4)
    finally:
        L.set_precision("f32")
    # reference on bf16-rounded operands (exact products, fp64 accumulate)
    ref = _ref(A.bfloat16().float(), B.bfloat16().float(), ta, tb)
    assert (C.double().cpu() - ref).abs().max().item() < 2e-3 * max(1.0, K ** 0.5 / 8)
    assert (C2.double().cpu() - ref).abs().max().item() < 2e-3 * max(1.0, K ** 0.5 / 8)
    assert L.get_precision() == "f32"


@pytest.mark.parametrize("M,N,K", [(256, 384, 192), (512, 256, 4096), (1024, 640, 64), (512, 512, 320), (256, 256, 128), (768, 256, 4160)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, True), (True, False)])
def test_gemm_bf16_staged_operands(ops, M, N, K, ta, tb):
    """bf16 mode with scratch: whole-tile shapes take the bf16-source kernels (operands staged as bf16 copies, split-K for
    the long K) — whole 256-tiles the 256 x 256 LDS-DMA kernel (even, odd and minimal K-tile counts, with and without
    split-K), the rest the 128 x 128 one; same contract as the converting kernel, and the two agree to summation order."""
    from tacotron2_subword_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = torch.randn((K, M) if ta else (M, K), generator=g).cuda()
    B = torch.randn((N, K) if tb else (K, N), generator=g).cuda()
    ws = torch.empty(8 * M * N + (M + N) * K, device="cuda")
    bias = torch.randn(N, generator=g).cuda()
    L.set_precision("bf16")
    try:
        C1 = ops.gemm(A, B, trans_a=ta, trans_b=tb, ws=ws, bias=bias)
        L.check(L.lib().t2_set_gemm_staging(0))
        C0 = ops.gemm(A, B, trans_a=ta, trans_b=tb, ws=ws, bias=bias)
    finally:
        L.check(L.lib().t2_set_gemm_staging(1))
        L.set_precision("f32")
    ref = _ref(A.bfloat16().float(), B.bfloat16().float(), ta, tb) + bias.double().cpu()
    assert (C1.double().cpu() - ref).abs().max().item() < 2e-3 * max(1.0, K ** 0.5 / 8)
    assert (C1 - C0).abs().max().item() < 1e-4 * max(1.0, K ** 0.5)


def test_gemm_bf16_tile256_epilogue(ops):
    """The 256 x 256 kernel's LDS-staged epilogue: bias + activation + alpha, accumulate into C (beta), a strided C,
    an explicit split-K — against fp64 on bf16-rounded operands."""
    from tacotron2_subword_amd import _lib as L
    g = torch.Generator().manual_seed(77)
    M, N, K = 512, 768, 1088                                     # 17 K-tiles (odd)
    A = torch.randn(M, K, generator=g).cuda(); B = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda(); C0 = torch.randn(M, N, generator=g).cuda()
    ws = torch.empty(8 * M * N + (M + N) * K, device="cuda")
    prod = A.bfloat16().double().cpu() @ B.bfloat16().double().cpu().t()
    tol = 2e-3 * max(1.0, K ** 0.5 / 8)
    L.set_precision("bf16")
    try:
        o1 = ops.gemm(A, B, bias=bias, act=1, alpha=0.5, ws=ws)
        o2 = C0.clone(); ops.gemm(A, B, alpha=1.0, beta=2.0, out=o2, ws=ws)
        big = torch.zeros(M, N + 64, device="cuda"); o3 = big[:, 32:32 + N]          # a strided C whose rows stay 16-byte aligned
        raw = lambda out, b: L.check(L.lib().t2_gemm(A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, K, 1, K, 1, out.stride(0),
                                                     b.data_ptr() if b is not None else None, 0, 1.0, 0.0, ws.data_ptr(), ws.numel() * 4, 0, L.stream()))
        raw(o3, bias)
        odd = torch.zeros(M, N + 3, device="cuda"); o4 = odd[:, 1:1 + N]             # rows that are NOT 16-byte aligned
        raw(o4, None)
        o5 = ops.gemm(A, B, ws=ws, splitk=3)
    finally:
        L.set_precision("f32")
    assert (o1.double().cpu() - torch.relu(0.5 * prod + bias.double().cpu())).abs().max().item() < tol
    assert (o2.double().cpu() - (prod + 2.0 * C0.double().cpu())).abs().max().item() < tol
    assert (o3.double().cpu() - (prod + bias.double().cpu())).abs().max().item() < tol
    assert big[:, :32].abs().max().item() == 0.0 and big[:, 32 + N:].abs().max().item() == 0.0
    assert (o4.double().cpu() - prod).abs().max().item() < tol
    assert odd[:, 0].abs().max().item() == 0.0 and odd[:, 1 + N:].abs().max().item() == 0.0
    assert (o5.double().cpu() - prod).abs().max().item() < tol


def test_prof_gemm_times_and_result(ops):
    """t2_prof_gemm (bench.py's GEMM figure): both timed variants leave the product in C and report positive times;
    the pre-staged variant is not slower than the one that casts its operands on every call."""
    import ctypes as C
    from tacotron2_subword_amd import _lib as L
    g = torch.Generator().manual_seed(9)
    M, N, K = 1024, 512, 768
    A = torch.randn(M, K, generator=g).cuda(); B = torch.randn(N, K, generator=g).cuda()
    out = torch.zeros(M, N, device="cuda")
    ws = torch.empty(8 << 20, device="cuda")
    a = L.GemmArgs()
    a.A, a.B, a.C, a.M, a.N, a.K = A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K
    a.sam, a.sak, a.sbn, a.sbk, a.ldc, a.batch, a.alpha, a.beta = K, 1, K, 1, N, 1, 1.0, 0.0
    a.ws, a.ws_bytes, a.splitk = ws.data_ptr(), ws.numel() * 4, 0
    mt, mk = C.c_float(), C.c_float()
    L.set_precision("bf16")
    try:
        L.check(L.lib().t2_prof_gemm(C.byref(a), 5, C.byref(mt), C.byref(mk), L.stream()))
    finally:
        L.set_precision("f32")
    ref = A.bfloat16().double().cpu() @ B.bfloat16().double().cpu().t()
    assert (out.double().cpu() - ref).abs().max().item() < 2e-3 * max(1.0, K ** 0.5 / 8)
    assert 0.0 < mk.value <= mt.value * 1.5 and mt.value < 50.0
    with pytest.raises(RuntimeError):                                   # fp32 mode: refused, not silently timed
        L.check(L.lib().t2_prof_gemm(C.byref(a), 5, C.byref(mt), C.byref(mk), L.stream()))
