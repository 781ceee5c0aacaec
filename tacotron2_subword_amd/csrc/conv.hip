// Conv1d(k) + BatchNorm1d + activation + dropout stacks on channels-last frames X[B*T, C]
// (Postnet.forward, model.py:65-70; the conv part of Encoder.forward, model.py:97-99), forward and
// backward, plus the embedding gather / gradient.
//
// The convolution is an implicit-im2col GEMM on the fp32 matrix cores (gemm.hip, ConvAddr): no
// column matrix is materialised.  Weights are re-laid-out per call from the reference's
// [Cout, Cin, k] to [Cout, k*Cin] (and flipped/transposed for the data gradient) — 5 MB per layer.
// BatchNorm statistics use two passes (mean, then centred sum of squares) over two-stage
// fixed-order column reductions, so results are reproducible run to run.
#include "kernels.h"

namespace t2 {

namespace {

inline int grid_for(size_t n, int block = 256, int cap = 8192) {
    size_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

// wp[co][dk*Ci + ci] = w[co][ci][dk]
__global__ void permute_w_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int K) {
    const size_t n = (size_t)Co * Ci * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci); const size_t r = i / Ci; const int dk = (int)(r % K), co = (int)(r / K);
        wp[i] = w[((size_t)co * Ci + ci) * K + dk];
    }
}
// wt[ci][dk*Co + co] = w[co][ci][K-1-dk]     (data gradient = correlation with the flipped kernel)
__global__ void permute_w_bwd_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co, int Ci, int K) {
    const size_t n = (size_t)Co * Ci * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co); const size_t r = i / Co; const int dk = (int)(r % K), ci = (int)(r / K);
        wt[i] = w[((size_t)co * Ci + ci) * K + (K - 1 - dk)];
    }
}
// dw[co][ci][dk] = dwp[co][dk*Ci + ci]
__global__ void unpermute_dw_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int K) {
    const size_t n = (size_t)Co * Ci * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int dk = (int)(i % K); const size_t r = i / K; const int ci = (int)(r % Ci), co = (int)(r / Ci);
        dw[i] = dwp[((size_t)co * K + dk) * Ci + ci];
    }
}

struct BnElem {
    const float* z; const float* mean; const float* invstd; const float* gamma; const float* beta;
    int C; int act; float drop_p; RngKey key;
};
__device__ __forceinline__ float act_fwd(float u, int act) {
    return act == ACT_RELU ? fmaxf(u, 0.f) : (act == ACT_TANH ? tanhf(u) : u);
}

// Two-stage column reductions.  MODE 0: sum z.  MODE 1: sum (z-mean)^2.
// MODE 2 (backward): du = dy * keep/(1-p) * act'(u), written to `du`; s0 = sum du, s1 = sum du*xhat.
template <int MODE>
__global__ void colreduce_stage1_kernel(BnElem e, const float* __restrict__ dy, float* __restrict__ du, int M, int slabs,
                                        float* __restrict__ scratch) {
    const int C = e.C;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int rows = (M + slabs - 1) / slabs;
    const int m0 = blockIdx.y * rows, m1 = min(M, m0 + rows);
    __shared__ float p0[4][64], p1[4][64];
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        const float mean = MODE >= 1 ? e.mean[c] : 0.f;
        const float inv = MODE == 2 ? e.invstd[c] : 0.f, ga = MODE == 2 ? e.gamma[c] : 0.f, be = MODE == 2 ? e.beta[c] : 0.f;
        const float scale = e.drop_p > 0.f ? 1.0f / (1.0f - e.drop_p) : 1.0f;
        for (int m = m0 + sub; m < m1; m += 4) {
            const size_t i = (size_t)m * C + c;
            const float z = e.z[i];
            if (MODE == 0) a0 += z;
            else if (MODE == 1) { const float d = z - mean; a0 += d * d; }
            else {
                const float xh = (z - mean) * inv, u = xh * ga + be;
                float g = dy[i];
                if (e.drop_p > 0.f) g = rng_keep(e.key, (uint32_t)i, e.drop_p) ? g * scale : 0.f;
                if (e.act == ACT_RELU) g = u > 0.f ? g : 0.f;
                else if (e.act == ACT_TANH) { const float t = tanhf(u); g *= 1.0f - t * t; }
                du[i] = g;
                a0 += g; a1 += g * xh;
            }
        }
    }
    p0[sub][threadIdx.x & 63] = a0; p1[sub][threadIdx.x & 63] = a1;
    __syncthreads();
    if (sub == 0 && c < C) {
        const int l = threadIdx.x;
        scratch[(size_t)blockIdx.y * C + c] = p0[0][l] + p0[1][l] + p0[2][l] + p0[3][l];
        if (MODE == 2) scratch[(size_t)(slabs + blockIdx.y) * C + c] = p1[0][l] + p1[1][l] + p1[2][l] + p1[3][l];
    }
}
// out0[c] = f(sum over slabs); MODE 0: mean = s/M ; MODE 1: invstd = rsqrt(s/M + eps), var_out = s/M ; MODE 2: raw sums
__global__ void colreduce_stage2_kernel(const float* __restrict__ scratch, int C, int slabs, int M, int mode, float eps,
                                        float* out0, float* out1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    // the slab partials are requested 16 at a time and added in slab order (one dependent load per slab cost 0.3 us each)
    auto ordered_sum = [&](const float* __restrict__ p) {
        float acc = 0.f;
        int s = 0;
        for (; s + 16 <= slabs; s += 16) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(s + j) * C];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j];
        }
        for (; s < slabs; ++s) acc += p[(size_t)s * C];
        return acc;
    };
    const float s0 = ordered_sum(scratch + c);
    const float s1 = mode == 2 ? ordered_sum(scratch + (size_t)slabs * C + c) : 0.f;
    if (mode == 0) out0[c] = s0 / (float)M;
    else if (mode == 1) { const float var = s0 / (float)M; out0[c] = 1.0f / sqrtf(var + eps); if (out1) out1[c] = var; }
    else { out0[c] = s0; out1[c] = s1; }
}

// y = dropout(act((z-mean)*invstd*gamma + beta)) (+ residual)
__global__ void bn_apply_kernel(BnElem e, const float* __restrict__ residual, float* __restrict__ y, size_t n) {
    const float scale = e.drop_p > 0.f ? 1.0f / (1.0f - e.drop_p) : 1.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % e.C);
        float v = act_fwd((e.z[i] - e.mean[c]) * e.invstd[c] * e.gamma[c] + e.beta[c], e.act);
        if (e.drop_p > 0.f) v = rng_keep(e.key, (uint32_t)i, e.drop_p) ? v * scale : 0.f;
        if (residual) v += residual[i];
        y[i] = v;
    }
}
// training: dz = gamma*invstd*(du - sum_du/M - xhat*sum_duxh/M) ; eval: dz = du*gamma*invstd
__global__ void bn_bwd_dz_kernel(BnElem e, const float* __restrict__ du, const float* __restrict__ sdu, const float* __restrict__ sduxh,
                                 int M, int training, float* __restrict__ dz, size_t n) {
    const float invM = 1.0f / (float)M;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % e.C);
        const float inv = e.invstd[c], ga = e.gamma[c];
        float g = du[i];
        if (training) {
            const float xh = (e.z[i] - e.mean[c]) * inv;
            g = g - sdu[c] * invM - xh * sduxh[c] * invM;
        }
        dz[i] = g * ga * inv;
    }
}
// running statistics (momentum 0.1, unbiased variance), model.py:42 nn.BatchNorm1d defaults
__global__ void bn_running_kernel(const float* mean, const float* var, int C, int M, float momentum, float* rm, float* rv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float unb = var[c] * ((float)M / (float)(M > 1 ? M - 1 : 1));
    rm[c] = (1.0f - momentum) * rm[c] + momentum * mean[c];
    rv[c] = (1.0f - momentum) * rv[c] + momentum * unb;
}
__global__ void invstd_from_var_kernel(const float* var, int C, float eps, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = 1.0f / sqrtf(var[c] + eps);
}

// out[row, :] = table[ids[row], :]
__global__ void embedding_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ table, float* __restrict__ out, int rows, int D) {
    const size_t n = (size_t)rows * (D / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / (D / 4)), q = (int)(i % (D / 4));
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(table + ids[r] * (long)D)[q];
    }
}
// dtable[v, :] = sum over rows with ids[row] == v of dout[row, :]   (one workgroup per vocabulary entry: fixed order)
__global__ void embedding_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ dout, float* __restrict__ dtable, int rows, int D) {
    const int v = blockIdx.x;
    extern __shared__ int hits[];
    __shared__ int wave_cnt[2][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    // ordered compaction of the rows that hit v (keeps the summation order independent of scheduling): every thread
    // carries the running count, the next block of ids is requested before the current one is processed, one barrier
    // per block of rows
    int run = 0;
    long idn = (int)threadIdx.x < rows ? ids[threadIdx.x] : -1;
    for (int r0 = 0, it = 0; r0 < rows; r0 += blockDim.x, ++it) {
        const int r = r0 + threadIdx.x, rn = r + blockDim.x;
        const long id = idn;
        idn = rn < rows ? ids[rn] : -1;
        const bool hit = r < rows && id == v;
        const unsigned long long m = __ballot(hit);
        int* wc = wave_cnt[it & 1];
        if (lane == 0) wc[wave] = __popcll(m);
        __syncthreads();
        int base = run, tot = 0;
        for (int w = 0; w < nw; ++w) { const int cw = wc[w]; if (w < wave) base += cw; tot += cw; }
        if (hit) hits[base + __popcll(m & ((1ull << lane) - 1ull))] = r;
        run += tot;
    }
    __syncthreads();
    const int nhit = run;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float acc = 0.f;                                   // rows requested 8 at a time, added in row order
        int h = 0;
        for (; h + 8 <= nhit; h += 8) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = dout[(size_t)hits[h + j] * D + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += x[j];
        }
        for (; h < nhit; ++h) acc += dout[(size_t)hits[h] * D + c];
        dtable[(size_t)v * D + c] = acc;
    }
}

template <int MODE>
int colreduce(const BnElem& e, const float* dy, float* du, int M, float eps, float* out0, float* out1, float* scratch, hipStream_t s) {
    const int slabs = M >= 64 * 64 ? 64 : (M >= 64 ? M / 64 : 1);
    hipLaunchKernelGGL(colreduce_stage1_kernel<MODE>, dim3((e.C + 63) / 64, slabs), dim3(256), 0, s, e, dy, du, M, slabs, scratch);
    T2_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_stage2_kernel, dim3((e.C + 255) / 256), dim3(256), 0, s, scratch, e.C, slabs, M, MODE, eps, out0, out1);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int conv_bn_fwd(const ConvBnFwd& a, hipStream_t s) {
    const int M = a.B * a.T;
    T2_REQUIRE(a.Cin % 4 == 0 && a.K % 2 == 1, "conv_bn_fwd: Cin=%d must be a multiple of 4 and the kernel size odd (%d)", a.Cin, a.K);
    T2_REQUIRE((size_t)M * a.Cout < (1ull << 32), "conv_bn_fwd: B*T*Cout too large for 32-bit RNG indices");
    const size_t nw = (size_t)a.Cout * a.Cin * a.K;
    hipLaunchKernelGGL(permute_w_fwd_kernel, dim3(grid_for(nw)), dim3(256), 0, s, a.w, a.wperm, a.Cout, a.Cin, a.K);
    T2_LAUNCH_CHECK();
    GemmDesc g = gemm_desc();
    g.A = a.x; g.conv_a = 1; g.conv_T = a.T; g.conv_C = a.Cin; g.conv_pad = (a.K - 1) / 2;
    g.B = a.wperm; g.sbn = (long)a.K * a.Cin; g.sbk = 1;
    g.C = a.z; g.ldc = a.Cout; g.M = M; g.N = a.Cout; g.K = a.K * a.Cin; g.bias1 = a.bias;
    g.ws = a.gemm_ws; g.ws_bytes = a.gemm_ws_bytes;
    T2_TRY_RC(gemm(g, s));
    BnElem e{a.z, a.mean, a.invstd, a.gamma, a.beta, a.Cout, a.act, a.drop_p, rng_key(a.seed, a.site)};
    if (a.training) {
        T2_TRY_RC(colreduce<0>(e, nullptr, nullptr, M, a.eps, a.mean, nullptr, a.scratch, s));
        T2_TRY_RC(colreduce<1>(e, nullptr, nullptr, M, a.eps, a.invstd, a.var, a.scratch, s));
        if (a.run_mean) {
            hipLaunchKernelGGL(bn_running_kernel, dim3((a.Cout + 255) / 256), dim3(256), 0, s, a.mean, a.var, a.Cout, M, 0.1f, a.run_mean, a.run_var);
            T2_LAUNCH_CHECK();
        }
    } else {
        T2_CHECK_HIP(hipMemcpyAsync(a.mean, a.run_mean, sizeof(float) * a.Cout, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(invstd_from_var_kernel, dim3((a.Cout + 255) / 256), dim3(256), 0, s, a.run_var, a.Cout, a.eps, a.invstd);
        T2_LAUNCH_CHECK();
    }
    const size_t n = (size_t)M * a.Cout;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(n)), dim3(256), 0, s, e, a.residual, a.y, n);
    T2_LAUNCH_CHECK();
    return 0;
}

int conv_bn_bwd(const ConvBnBwd& a, hipStream_t s) {
    const int M = a.B * a.T;
    const size_t n = (size_t)M * a.Cout;
    BnElem e{a.z, a.mean, a.invstd, a.gamma, a.beta, a.Cout, a.act, a.drop_p, rng_key(a.seed, a.site)};
    // du (into a.dz) + the two column sums; d(gamma) = sum du*xhat, d(beta) = sum du
    T2_TRY_RC(colreduce<2>(e, a.dy, a.dz, M, a.eps, a.dbeta, a.dgamma, a.scratch, s));
    hipLaunchKernelGGL(bn_bwd_dz_kernel, dim3(grid_for(n)), dim3(256), 0, s, e, a.dz, a.dbeta, a.dgamma, M, a.training, a.dz, n);
    T2_LAUNCH_CHECK();
    T2_TRY_RC(colsum(a.dz, a.Cout, M, a.Cout, a.dbias, nullptr, a.scratch, s));
    // d(weight)[co][dk*Ci+ci] = sum_m dz[m,co] * V[m, dk*Ci+ci]   (implicit im2col on the B side)
    GemmDesc g = gemm_desc();
    g.A = a.dz; g.sam = 1; g.sak = a.Cout;
    g.B = a.x; g.conv_b = 1; g.conv_T = a.T; g.conv_C = a.Cin; g.conv_pad = (a.K - 1) / 2;
    g.C = a.wperm; g.ldc = (long)a.K * a.Cin; g.M = a.Cout; g.N = a.K * a.Cin; g.K = M;
    g.ws = a.gemm_ws; g.ws_bytes = a.gemm_ws_bytes;
    T2_TRY_RC(gemm(g, s));
    const size_t nw = (size_t)a.Cout * a.Cin * a.K;
    hipLaunchKernelGGL(unpermute_dw_kernel, dim3(grid_for(nw)), dim3(256), 0, s, a.wperm, a.dw, a.Cout, a.Cin, a.K);
    T2_LAUNCH_CHECK();
    if (a.dx) {
        // d(input) = correlation of dz with the flipped, transposed kernel
        T2_REQUIRE(a.Cout % 4 == 0, "conv_bn_bwd: Cout=%d must be a multiple of 4", a.Cout);
        hipLaunchKernelGGL(permute_w_bwd_kernel, dim3(grid_for(nw)), dim3(256), 0, s, a.w, a.wperm, a.Cout, a.Cin, a.K);
        T2_LAUNCH_CHECK();
        GemmDesc h = gemm_desc();
        h.A = a.dz; h.conv_a = 1; h.conv_T = a.T; h.conv_C = a.Cout; h.conv_pad = (a.K - 1) / 2;
        h.B = a.wperm; h.sbn = (long)a.K * a.Cout; h.sbk = 1;
        h.C = a.dx; h.ldc = a.Cin; h.M = M; h.N = a.Cin; h.K = a.K * a.Cout;
        h.beta = a.dx_accumulate ? 1.f : 0.f;
        h.ws = a.gemm_ws; h.ws_bytes = a.gemm_ws_bytes;
        T2_TRY_RC(gemm(h, s));
    }
    return 0;
}

int embedding_fwd(const long* ids, const float* table, float* out, int rows, int D, hipStream_t s) {
    T2_REQUIRE(D % 4 == 0, "embedding: dim %d must be a multiple of 4", D);
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for((size_t)rows * (D / 4))), dim3(256), 0, s, ids, table, out, rows, D);
    T2_LAUNCH_CHECK();
    return 0;
}
int embedding_bwd(const long* ids, const float* dout, float* dtable, int rows, int D, int vocab, hipStream_t s) {
    const size_t smem = (size_t)rows * sizeof(int);
    T2_REQUIRE(smem <= 150 * 1024, "embedding_bwd: %d tokens do not fit the LDS hit list", rows);
    T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(embedding_bwd_kernel), smem));
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(vocab), dim3(256), smem, s, ids, dout, dtable, rows, D);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
