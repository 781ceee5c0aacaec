// One LSTM time step for up to 4 independent cells in one launch ("streams": the phone and
// sub-word attention LSTMs of Decoder.decode, or the two directions of an encoder BiLSTM).
//
//   gates = pre[b,:] (+bias1+bias2) + sum_seg x_seg[b,:] . W_seg[n,:]^T        (skinny GEMM, M = B)
//   i,f,g,o = sigmoid, sigmoid, tanh, sigmoid ; c' = f*c + i*g ; h' = o*tanh(c')
//   h_out = dropout(h'), c_out = dropout(c')                                   (model.py:340-346,371-373)
//
// Work split: one workgroup owns 8 hidden units = 32 gate rows of W for ALL batch rows, so every
// weight element is read exactly once per step chip-wide (the step is weight-streaming bound).
// Inside the workgroup the K range of every segment is split over the NW waves; each wave runs
// v_mfma_f32_32x32x2_f32 on operands loaded straight from global memory (16 B per lane along K:
// lane (r, hk) takes k = k0 + 4*hk + j for MFMA j — A and B use the same permutation, so the
// sum is unchanged), then the NW partial tiles are summed through LDS in a fixed order.
// Optionally the workgroup also emits its 8-unit partial of the attention query projection
// W_q h (attention.py:368) so that no separate launch is needed between LSTM and attention.
#include "kernels.h"

namespace t2 {

namespace {

constexpr int HU = 8;        // hidden units per workgroup
constexpr int NW = 8;        // waves per workgroup (split-K)
constexpr int PP = 33;       // LDS pitch of a 32-wide partial tile

template <int MT>
__global__ __launch_bounds__(NW * 64) void lstm_step_fwd_kernel(LstmStepDesc d) {
    const LstmStream& st = d.st[blockIdx.y];
    const int B = d.B, H = d.H;
    const int u0 = blockIdx.x * HU;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hk = lane >> 5;
    const int wrow = (r >> 3) * H + u0 + (r & 7);    // column n = gate*8 + unit  ->  row of W

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* part = smem;                              // [NW][32][PP]
    float* hs = smem + NW * 32 * PP;                 // [MT*32][HU]   post-dropout h of this group

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    // The pointwise tail needs pre[b, g*H+u] (streamed once from HBM) and c_prev: request them
    // now so that their latency hides under the GEMM instead of being paid after it.
    constexpr bool kPrefetch = MT <= 4;
    float pre_v[kPrefetch ? MT : 1][4];
    float cp_v[kPrefetch ? MT : 1];
    if (kPrefetch && threadIdx.x < 32 * HU) {
        const int bl = threadIdx.x >> 3, u = u0 + (threadIdx.x & 7);
#pragma unroll
        for (int m = 0; m < (kPrefetch ? MT : 1); ++m) {
            const int b = m * 32 + bl;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = 0.f;
                if (b < B) {
                    if (st.pre) v += st.pre[(long)b * st.ldpre + g * H + u];
                    if (st.bias1) v += st.bias1[g * H + u];
                    if (st.bias2) v += st.bias2[g * H + u];
                }
                pre_v[m][g] = v;
            }
            cp_v[m] = (b < B && st.c_prev) ? st.c_prev[(long)b * st.ldc_prev + u] : 0.f;
        }
    }

    for (int s = 0; s < st.nseg; ++s) {
        const LstmSeg sg = st.seg[s];
        const int kq = sg.k / NW;
        const float* wp = sg.w + (long)wrow * sg.ldw + wave * kq + 4 * hk;
        const float* xp[MT];
        bool xv[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m * 32 + r;
            xv[m] = row < B;
            xp[m] = sg.x + (long)(xv[m] ? row : 0) * sg.ldx + wave * kq + 4 * hk;
        }
        // Keep many 16-byte loads in flight per wave: at one workgroup per CU the loop is bound by
        // L2/MALL latency, not by the MFMA rate, unless ~100 KB per CU are outstanding.  So the
        // operands of U k-steps are requested first and only then fed to the matrix core.
        constexpr int U = MT <= 2 ? 8 : (MT <= 4 ? 4 : 1);
        int k = 0;
        for (; k + 8 * U <= kq; k += 8 * U) {
            f32x4 w4[U], x4[U][MT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                w4[u] = *reinterpret_cast<const f32x4*>(wp + k + 8 * u);
#pragma unroll
                for (int m = 0; m < MT; ++m) x4[u][m] = *reinterpret_cast<const f32x4*>(xp[m] + k + 8 * u);
            }
            __builtin_amdgcn_sched_barrier(0);      // all U*(1+MT) loads are issued before the first MFMA
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[m] ? x4[u][m][j] : 0.f, w4[u][j], acc[m], 0, 0, 0);
        }
        for (; k < kq; k += 8) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + k);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(xp[m] + k);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[m] ? x4[j] : 0.f, w4[j], acc[m], 0, 0, 0);
            }
        }
    }

    const RngKey kh = rng_key(d.seed, st.site_h), kc = rng_key(d.seed, st.site_c);
    const float scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m > 0) __syncthreads();
        // lane holds column r, rows (e&3) + 8*(e>>2) + 4*hk
#pragma unroll
        for (int e = 0; e < 16; ++e)
            part[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PP + r] = acc[m][e];
        __syncthreads();
        if (threadIdx.x < 32 * HU) {
            const int bl = threadIdx.x >> 3, uu = threadIdx.x & 7;
            const int b = m * 32 + bl, u = u0 + uu;
            if (b < B) {
                float g4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sum += part[(w * 32 + bl) * PP + g * 8 + uu];
                    if (kPrefetch) {
                        sum += pre_v[kPrefetch ? m : 0][g];
                    } else {
                        if (st.pre) sum += st.pre[(long)b * st.ldpre + g * H + u];
                        if (st.bias1) sum += st.bias1[g * H + u];
                        if (st.bias2) sum += st.bias2[g * H + u];
                    }
                    g4[g] = sum;
                }
                const bool active = !st.lengths || st.t < st.lengths[b];
                float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf(g4[2]), og = sigmoidf_(g4[3]);
                const float cp = kPrefetch ? cp_v[kPrefetch ? m : 0] : (st.c_prev ? st.c_prev[(long)b * st.ldc_prev + u] : 0.f);
                float cn = fg * cp + ig * gg;
                float hn = og * tanhf(cn);
                if (!active) { ig = fg = gg = og = 0.f; cn = 0.f; hn = 0.f; }
                if (st.gates) {
                    float* gp = st.gates + (long)b * st.ldgates + u;
                    gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
                }
                if (st.c_new) st.c_new[(long)b * st.ldc_new + u] = cn;
                float ho = hn, co = cn;
                if (d.drop_p > 0.f) {
                    const uint32_t idx = st.idx_base + (uint32_t)b * st.idx_bstride + (uint32_t)u;
                    ho = rng_keep(kh, idx, d.drop_p) ? hn * scale : 0.f;
                    co = rng_keep(kc, idx, d.drop_p) ? cn * scale : 0.f;
                }
                st.h_out[(long)b * st.ldh_out + u] = ho;
                if (st.h_out2) st.h_out2[(long)b * st.ldh_out2 + u] = ho;
                st.c_out[(long)b * st.ldc_out + u] = co;
                hs[b * HU + uu] = ho;
            }
        }
    }

    if (st.wq) {
        // partial query projection of this unit group: qpart[group][b][a] = sum_uu h[b,u0+uu] * Wq[a,u0+uu]
        __syncthreads();
        float* wqs = part;                            // [A][HU]   (A*HU <= NW*32*PP)
        const int A = st.A;
        for (int i = threadIdx.x; i < A * HU; i += NW * 64) wqs[i] = st.wq[(long)(i / HU) * H + u0 + (i % HU)];
        __syncthreads();
        float* qp = st.qpart + (long)blockIdx.x * B * A;
        for (int i = threadIdx.x; i < B * A; i += NW * 64) {
            const int b = i / A, a = i % A;
            float sum = 0.f;
#pragma unroll
            for (int uu = 0; uu < HU; ++uu) sum += hs[b * HU + uu] * wqs[a * HU + uu];
            qp[i] = sum;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// backward, part 1: pointwise.  One thread per (b, u).
//   dh  = direct sources + sum of recurrent partials (+ dq . Wq)          gradient on h_out(t)
//   dhn = dh * keep_h/(1-p) ;  dcn = dc_out * keep_c/(1-p) + dhn * o * (1 - tanh(cn)^2)
//   d(pre-activations) = { dcn*g*i(1-i), dcn*c_prev*f(1-f), dcn*i*(1-g^2), dhn*tanh(cn)*o(1-o) }
//   dc_state <- dcn * f                                                    gradient on c_out(t-1)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_bwd_pointwise_kernel(LstmBwdPointDesc d) {
    const LstmBwdStream& st = d.st[blockIdx.y];
    const int B = d.B, H = d.H;
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= (long)B * H) return;
    const int b = (int)(i / H), u = (int)(i % H);
    float dh = 0.f;
    if (st.dh1) dh += st.dh1[(long)b * st.lddh1 + u];
    if (st.dh2) dh += st.dh2[(long)b * st.lddh2 + u];
    if (st.part && !d.first) {
        const float* p = st.part + (long)b * st.ldpart + st.part_col + u;
        float acc = 0.f;
        for (int z = 0; z < st.nparts; ++z) acc += p[(long)z * st.part_stride];
        dh += acc;
    }
    if (st.dq) {
        const float* q = st.dq + (long)b * st.lddq;
        const float* w = st.wq + u;
        float acc = 0.f;
#pragma unroll 16
        for (int a = 0; a < st.A; ++a) acc += q[a] * w[(long)a * H];
        dh += acc;
    }
    float dc = d.first ? 0.f : st.dc_state[i];
    if (d.drop_p > 0.f) {
        const uint32_t idx = st.idx_base + (uint32_t)b * st.idx_bstride + (uint32_t)u;
        const float scale = 1.0f / (1.0f - d.drop_p);
        dh = rng_keep(rng_key(d.seed, st.site_h), idx, d.drop_p) ? dh * scale : 0.f;
        dc = rng_keep(rng_key(d.seed, st.site_c), idx, d.drop_p) ? dc * scale : 0.f;
    }
    const float* gp = st.gates + (long)b * st.ldgates + u;
    const float ig = gp[0], fg = gp[H], gg = gp[2 * H], og = gp[3 * H];
    const float cn = st.c_new[(long)b * st.ldc_new + u];
    const float cp = st.c_prev ? st.c_prev[(long)b * st.ldc_prev + u] : 0.f;
    const float tc = tanhf(cn);
    const float dcn = dc + dh * og * (1.0f - tc * tc);
    float* dg = st.dg + (long)b * st.lddg + u;
    dg[0] = dcn * gg * ig * (1.0f - ig);
    dg[H] = dcn * cp * fg * (1.0f - fg);
    dg[2 * H] = dcn * ig * (1.0f - gg * gg);
    dg[3 * H] = dh * tc * og * (1.0f - og);
    st.dc_state[i] = dcn * fg;
}

// ---------------------------------------------------------------------------------------------
// backward, part 2: part[z][b][n] = sum_{k in K-split z} dg[b,k] * W[k,n]   for the recurrent
// input columns n (ctx | h for the attention LSTMs, h for the decoder / encoder LSTMs).
// grid = (column tiles of 32, K-splits, streams); inside a workgroup the K range is split again
// over NW waves (v_mfma_f32_32x32x2_f32, operands from global memory) and summed through LDS.
// ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_gemm_kernel(LstmBwdGemmDesc d) {
    const LstmBwdGemmStream& st = d.st[blockIdx.z];
    const int B = d.B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hk = lane >> 5;
    // locate this column tile
    int col0 = blockIdx.x * 32, sidx = 0, cbase = 0;
    while (sidx < st.nseg - 1 && col0 >= cbase + st.seg[sidx].ncols) { cbase += st.seg[sidx].ncols; ++sidx; }
    const LstmBwdSeg sg = st.seg[sidx];
    const int kspan = d.H4 / d.KS, kw = kspan / NW;
    const int kbeg = blockIdx.y * kspan + wave * kw;
    const float* wp = sg.w + (long)(kbeg + 4 * hk) * sg.ldw + (col0 - cbase) + r;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* part = smem;                              // [NW][32][PP]

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    const float* xp[MT];
    bool xv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m * 32 + r;
        xv[m] = row < B;
        xp[m] = st.dg + (long)(xv[m] ? row : 0) * st.lddg + kbeg + 4 * hk;
    }
    constexpr int U = MT <= 2 ? 4 : 2;
    int k = 0;
    for (; k + 8 * U <= kw; k += 8 * U) {
        float w4[U][4];
        f32x4 x4[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w4[u][j] = wp[(long)(k + 8 * u + j) * sg.ldw];
#pragma unroll
            for (int m = 0; m < MT; ++m) x4[u][m] = *reinterpret_cast<const f32x4*>(xp[m] + k + 8 * u);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[m] ? x4[u][m][j] : 0.f, w4[u][j], acc[m], 0, 0, 0);
    }
    for (; k < kw; k += 8) {
        float w4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w4[j] = wp[(long)(k + j) * sg.ldw];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(xp[m] + k);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[m] ? x4[j] : 0.f, w4[j], acc[m], 0, 0, 0);
        }
    }
    float* out = st.part + (long)blockIdx.y * B * d.NC;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m > 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e)
            part[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PP + r] = acc[m][e];
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * 32; i += NW * 64) {
            const int bl = i >> 5, c = i & 31, b = m * 32 + bl;
            if (b < B) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += part[(w * 32 + bl) * PP + c];
                out[(long)b * d.NC + col0 + c] = sum;
            }
        }
    }
}

}  // namespace

int lstm_step_fwd(const LstmStepDesc& d, hipStream_t s) {
    T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= kMaxLstmStreams, "lstm_step: nstreams=%d", d.nstreams);
    T2_REQUIRE(d.B >= 1 && d.B <= 256, "lstm_step: batch %d not in [1,256]", d.B);
    T2_REQUIRE(d.H % HU == 0, "lstm_step: H=%d must be a multiple of %d", d.H, HU);
    for (int i = 0; i < d.nstreams; ++i) {
        const LstmStream& st = d.st[i];
        T2_REQUIRE(st.nseg >= 0 && st.nseg <= kMaxSeg, "lstm_step: nseg=%d", st.nseg);
        for (int j = 0; j < st.nseg; ++j) {
            const LstmSeg& g = st.seg[j];
            T2_REQUIRE(g.k % (8 * NW) == 0, "lstm_step: segment width %d must be a multiple of %d", g.k, 8 * NW);
            T2_REQUIRE(g.ldx % 4 == 0 && g.ldw % 4 == 0 && ((uintptr_t)g.x & 15) == 0 && ((uintptr_t)g.w & 15) == 0,
                       "lstm_step: segment %d operands must be 16-byte aligned (ldx=%ld ldw=%ld)", j, g.ldx, g.ldw);
        }
        T2_REQUIRE(!st.wq || st.A * HU <= NW * 32 * PP, "lstm_step: attention dim %d too large", st.A);
    }
    const int MT = (d.B + 31) / 32;
    dim3 grid(d.H / HU, d.nstreams), block(NW * 64);
    auto smem = [&](int mt) { return (size_t)(NW * 32 * PP + mt * 32 * HU) * sizeof(float); };
    if (MT <= 1) hipLaunchKernelGGL(lstm_step_fwd_kernel<1>, grid, block, smem(1), s, d);
    else if (MT <= 2) hipLaunchKernelGGL(lstm_step_fwd_kernel<2>, grid, block, smem(2), s, d);
    else if (MT <= 4) hipLaunchKernelGGL(lstm_step_fwd_kernel<4>, grid, block, smem(4), s, d);
    else hipLaunchKernelGGL(lstm_step_fwd_kernel<8>, grid, block, smem(8), s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2

namespace t2 {

int lstm_bwd_pointwise(const LstmBwdPointDesc& d, hipStream_t s) {
    T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= kMaxLstmStreams, "lstm_bwd_pointwise: nstreams=%d", d.nstreams);
    const long n = (long)d.B * d.H;
    hipLaunchKernelGGL(lstm_bwd_pointwise_kernel, dim3((unsigned)((n + 255) / 256), d.nstreams), dim3(256), 0, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

int lstm_bwd_ksplit(int H4) {
    int ks = 8;
    while (ks > 1 && H4 % (ks * NW * 8) != 0) ks >>= 1;
    return ks;
}

int lstm_bwd_gemm(const LstmBwdGemmDesc& d, hipStream_t s) {
    T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= kMaxLstmStreams, "lstm_bwd_gemm: nstreams=%d", d.nstreams);
    T2_REQUIRE(d.B >= 1 && d.B <= 256, "lstm_bwd_gemm: batch %d", d.B);
    T2_REQUIRE(d.KS >= 1 && d.H4 % (d.KS * NW * 8) == 0, "lstm_bwd_gemm: 4H=%d not divisible by KS*64 (KS=%d)", d.H4, d.KS);
    int nc = 0;
    for (int j = 0; j < d.st[0].nseg; ++j) nc += d.st[0].seg[j].ncols;
    T2_REQUIRE(nc == d.NC && nc % 32 == 0, "lstm_bwd_gemm: column count %d (NC=%d) must be a multiple of 32", nc, d.NC);
    for (int i = 0; i < d.nstreams; ++i) {
        T2_REQUIRE(d.st[i].lddg % 4 == 0 && ((uintptr_t)d.st[i].dg & 15) == 0, "lstm_bwd_gemm: dg must be 16-byte aligned");
        for (int j = 0; j < d.st[i].nseg; ++j) T2_REQUIRE(d.st[i].seg[j].ncols % 32 == 0, "lstm_bwd_gemm: segment cols %d", d.st[i].seg[j].ncols);
    }
    const int MT = (d.B + 31) / 32;
    dim3 grid(d.NC / 32, d.KS, d.nstreams), block(NW * 64);
    const size_t smem = (size_t)NW * 32 * PP * sizeof(float);
    if (MT <= 1) hipLaunchKernelGGL(lstm_bwd_gemm_kernel<1>, grid, block, smem, s, d);
    else if (MT <= 2) hipLaunchKernelGGL(lstm_bwd_gemm_kernel<2>, grid, block, smem, s, d);
    else if (MT <= 4) hipLaunchKernelGGL(lstm_bwd_gemm_kernel<4>, grid, block, smem, s, d);
    else hipLaunchKernelGGL(lstm_bwd_gemm_kernel<8>, grid, block, smem, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
