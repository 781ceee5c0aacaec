// One attention step for both decoder streams (phone / sub-word) in one launch:
// one workgroup per (batch item, stream).
//
//   kind 0  StepwiseMonotonicAttention.forward   (attention.py:374-398)
//           e_j = v . tanh(q + pm_j) ; mask -> -inf ; (+ noise*std in training) ; p = sigmoid(e)
//           a_t[j] = a_{t-1}[j] p_j + a_{t-1}[j-1] (1 - p_{j-1})   ; ctx = a_t . memory
//   kind 1  LocationSensitiveAttention.forward   (attention.py:64-85, LocationLayer :7-23)
//           e_j = v . tanh(q + dense(conv([w_prev; w_cum]))_j + pm_j) ; mask ; softmax ; ctx = w . memory
//
// The processed-memory rows (A floats) are read 16 B per lane by groups of 16 lanes per
// position j and reduced with wave shuffles; the encoder memory rows (E floats) are read
// 16 B per lane, fully coalesced, once per step.  Positions whose weight is exactly 0 are
// skipped in the context sum (SMA alignments are sparse early on; 0*x contributes nothing).
#include <algorithm>

#include "kernels.h"

#ifdef T2_STAMPS
__device__ unsigned long long t2_stamps_attn[32];
#define T2_ASTAMP(o, i)                                                                          \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (blockIdx.x == 7 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) t2_stamps_attn[(o) + (i)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
extern "C" int t2_debug_read_stamps_attn(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(t2_stamps_attn), sizeof(unsigned long long) * n);
}
#else
#define T2_ASTAMP(o, i)
#endif

namespace t2 {

namespace {

constexpr int NT = 1024;     // 16 waves: the kernels are latency-bound row streams, more waves = more loads in flight

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    // red: >= 4 floats of LDS; result broadcast to all threads
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

__global__ __launch_bounds__(NT) void attention_step_fwd_kernel(AttnStepDesc d) {
    const AttnStream& st = d.st[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Tin = st.Tin, A = d.A, E = d.E, F = d.F, Kc = d.Kc;
    const int Tp = (Tin + 3) & ~3;

    T2_ASTAMP(0, 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* q = smem;                 // [A]
    float* vs = q + A;               // [A]
    float* e = vs + A;               // [Tp]
    float* ap = e + Tp;              // [Tp]
    float* an = ap + Tp;             // [Tp]
    float* red = an + Tp;            // [4*NT]  (query partial groups: (NT/(A/4)) * A floats)
    float* cred = red + 4 * NT;      // [nh*E]
    const int nd = E / 4, nh = NT / nd;
    float* lsa = cred + nh * E;      // LSA only: convw[F*2*Kc] dense[A*(F+1)] loc[Tin*(F+1)] wpad[2][Tin+Kc-1]

    // Processed-memory rows of the first 2 x NT/16 positions (2 x 64 channels per lane) are requested now: they do
    // not depend on the query, and the kernel is a chain of dependent L2/MALL round trips
    constexpr int PFJ = 2, PFA = 2;
    f32x4 pmv[PFJ][PFA];
    {
        const int gid = tid >> 4, sub = tid & 15;
#pragma unroll
        for (int i = 0; i < PFJ; ++i)
#pragma unroll
            for (int k = 0; k < PFA; ++k)
                pmv[i][k] = *reinterpret_cast<const f32x4*>(st.pm + ((long)b * Tin + min(gid + i * (NT / 16), Tin - 1)) * A + min(sub * 4 + 64 * k, A - 4));
    }

    // ---- query: direct, or ordered sum of the partials emitted by lstm_step_fwd
    if (st.qpart) {
        // groups of A/4 lanes read one partial row (16 B per lane); NT/(A/4) rows in flight per pass
        const int a4n = A / 4, ng = NT / a4n;
        const int pg = tid / a4n, a4 = (tid % a4n) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* p = st.qpart + (long)b * A + a4;
        const long ps = (long)d.B * A;
#pragma unroll 4
        for (int i = pg; i < st.nparts; i += ng) acc += *reinterpret_cast<const f32x4*>(p + (long)i * ps);
        *reinterpret_cast<f32x4*>(red + pg * A + a4) = acc;
        __syncthreads();
        if (tid < A) {
            float sum = 0.f;
            const int used = st.nparts < ng ? st.nparts : ng;
            for (int h2 = 0; h2 < used; ++h2) sum += red[h2 * A + tid];
            q[tid] = sum;
            if (st.q_out) st.q_out[(long)b * st.ldq_out + tid] = sum;
        }
    } else {
        for (int a = tid; a < A; a += NT) q[a] = st.query[(long)b * st.ldq + a];
    }
    for (int a = tid; a < A; a += NT) vs[a] = st.v[a];

    // previous alignment / weights
    for (int j = tid; j < Tin; j += NT)
        ap[j] = st.a_prev ? st.a_prev[(long)b * st.lda_prev + j] : ((d.kind == 0 && j == 0) ? 1.f : 0.f);

    float* loc = nullptr; float* dense = nullptr; float* paS = nullptr;
    const int F1 = F + 1, PA = A + 8;            // PA % 16 == 8: the two row-halves of an MFMA tile store to disjoint banks
    if (d.kind == 1) {
        float* convw = lsa;
        dense = convw + F * 2 * Kc;
        loc = dense + A * F1;
        float* wpad = loc + ((Tin * F1 + 3) & ~3);
        const int pad = (Kc - 1) / 2, Tw = Tin + Kc - 1, TwP = (Tw + 4 + 3) & ~3;
        if (d.lsa_pa) paS = wpad + 2 * TwP;
        for (int i = tid; i < F * 2 * Kc; i += NT) convw[i] = st.loc_conv[i];
        for (int i = tid; i < A * F1; i += NT) dense[i] = (i % F1) < F ? st.loc_dense[(i / F1) * F + (i % F1)] : 0.f;
        for (int i = tid; i < 2 * TwP; i += NT) {
            const int c = i / TwP, j = i % TwP - pad;
            float v = 0.f;
            if (j >= 0 && j < Tin) v = c == 0 ? (st.a_prev ? st.a_prev[(long)b * st.lda_prev + j] : 0.f)
                                              : (st.wcum_prev ? st.wcum_prev[(long)b * st.ldwcum_prev + j] : 0.f);
            wpad[i] = v;
        }
        __syncthreads();
        // location conv: loc[j][f] = sum_c sum_k convw[f][c][k] * wcat[c][j + k - pad]; one thread per (f, 4 positions):
        // the sliding window lives in registers, 2 LDS reads per 4 MACs
        const int nj4 = (Tin + 3) / 4;
        for (int i = tid; i < nj4 * F; i += NT) {
            const int f = i % F, j0 = (i / F) * 4;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int c = 0; c < 2; ++c) {
                const float* w = convw + (f * 2 + c) * Kc;
                const float* x = wpad + c * TwP + j0;
                float x0 = x[0], x1 = x[1], x2 = x[2];
                for (int k = 0; k < Kc; ++k) {
                    const float x3 = x[k + 3], wk = w[k];
                    a0 += wk * x0; a1 += wk * x1; a2 += wk * x2; a3 += wk * x3;
                    x0 = x1; x1 = x2; x2 = x3;
                }
            }
            loc[j0 * F1 + f] = a0;
            if (j0 + 1 < Tin) loc[(j0 + 1) * F1 + f] = a1;
            if (j0 + 2 < Tin) loc[(j0 + 2) * F1 + f] = a2;
            if (j0 + 3 < Tin) loc[(j0 + 3) * F1 + f] = a3;
        }
        for (int j = tid; j < Tin; j += NT) loc[j * F1 + F] = 0.f;      // pad column (K rounded up to even for the MFMA)
        if (paS) {
            // location features through the dense layer on the matrix cores: pa[j][a] = sum_f loc[j][f] * Wd[a][f]
            // (exact fp32 fma chains, v_mfma_f32_32x32x2_f32), one 32x32 tile per wave
            __syncthreads();
            const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
            const int njt = (Tin + 31) / 32, nat = A / 32, Ke = (F + 1) & ~1;
            for (int tile = wave; tile < njt * nat; tile += NT / 64) {
                const int jt = tile / nat, at = tile % nat;
                const float* lr = loc + min(jt * 32 + r, Tin - 1) * F1 + h;
                const float* dr = dense + (at * 32 + r) * F1 + h;
                f32x16 acc;
#pragma unroll
                for (int e2 = 0; e2 < 16; ++e2) acc[e2] = 0.f;
                for (int kk = 0; kk < Ke; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lr[kk], dr[kk], acc, 0, 0, 0);
#pragma unroll
                for (int e2 = 0; e2 < 16; ++e2) {
                    const int row = jt * 32 + (e2 & 3) + 8 * (e2 >> 2) + 4 * h;
                    if (row < Tin) paS[row * PA + at * 32 + r] = acc[e2];
                }
            }
        }
    }
    __syncthreads();

    T2_ASTAMP(0, 1);
    // ---- energies: 16 lanes per position j
    {
        const int gid = tid >> 4, sub = tid & 15;
        auto chunk = [&](int j, int a, const f32x4 pv) {      // 4 channels of position j
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float u = q[a + c] + pv[c];
                if (d.kind == 1) {
                    if (paS) u += paS[j * PA + a + c];
                    else {
                        float pa = 0.f;
                        const float* dr = dense + (a + c) * F1;
                        const float* lr = loc + j * F1;
                        for (int f = 0; f < F; ++f) pa += dr[f] * lr[f];
                        u += pa;
                    }
                }
                part += vs[a + c] * tanhf(u);
            }
            return part;
        };
        auto finish = [&](int j, float sum) {
            sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 1, 64);
            if (sub == 0 && j < Tin) e[j] = sum;
        };
#pragma unroll
        for (int i = 0; i < PFJ; ++i) {                       // prefetched positions (whole 16-lane groups stay converged)
            const int j = gid + i * (NT / 16);
            if (j < Tp) {
                float sum = 0.f;
                if (j < Tin) {
#pragma unroll
                    for (int k = 0; k < PFA; ++k) if (sub * 4 + 64 * k < A) sum += chunk(j, sub * 4 + 64 * k, pmv[i][k]);
                    const float* pmr = st.pm + ((long)b * Tin + j) * A;
                    for (int a = sub * 4 + 64 * PFA; a < A; a += 64) sum += chunk(j, a, *reinterpret_cast<const f32x4*>(pmr + a));
                }
                finish(j, sum);
            }
        }
        for (int j = gid + PFJ * (NT / 16); j < Tp; j += NT / 16) {
            float sum = 0.f;
            if (j < Tin) {
                const float* pmr = st.pm + ((long)b * Tin + j) * A;
                for (int a = sub * 4; a < A; a += 64) sum += chunk(j, a, *reinterpret_cast<const f32x4*>(pmr + a));
            }
            finish(j, sum);
        }
    }
    __syncthreads();

    T2_ASTAMP(0, 2);
    int len = st.lengths ? st.lengths[b] : Tin;
    if (d.max_pos > 0) len = min(len, d.max_pos);
    if (d.kind == 0) {
        const RngKey key = rng_key(d.seed, st.site_noise);
        for (int j = tid; j < Tin; j += NT) {
            float ev = e[j];
            if (j >= len) ev = st.mask_value;
            if (d.noise_std > 0.f) ev += d.noise_std * rng_normal(key, st.idx_base + (uint32_t)b * st.idx_bstride + (uint32_t)j);
            const float p = sigmoidf_(ev);
            e[j] = p;
            if (st.p_out) st.p_out[(long)b * st.ldp_out + j] = p;
        }
        __syncthreads();
        for (int j = tid; j < Tin; j += NT) {
            float a = ap[j] * e[j];
            if (j > 0) a += ap[j - 1] * (1.0f - e[j - 1]);
            an[j] = a;
            st.a_out[(long)b * st.lda_out + j] = a;
        }
    } else {
        float mx = -INFINITY;
        for (int j = tid; j < Tin; j += NT) {
            float ev = e[j];
            if (j >= len) ev = st.mask_value;
            e[j] = ev;
            mx = fmaxf(mx, ev);
        }
        mx = block_reduce(mx, red, true);
        float sum = 0.f;
        for (int j = tid; j < Tin; j += NT) { const float x = expf(e[j] - mx); e[j] = x; sum += x; }
        sum = block_reduce(sum, red, false);
        const float inv = 1.0f / sum;
        for (int j = tid; j < Tin; j += NT) {
            const float w = e[j] * inv;
            an[j] = w;
            st.a_out[(long)b * st.lda_out + j] = w;
            if (st.wcum_out)
                st.wcum_out[(long)b * st.ldwcum_out + j] = (st.wcum_prev ? st.wcum_prev[(long)b * st.ldwcum_prev + j] : 0.f) + w;
        }
    }
    __syncthreads();

    T2_ASTAMP(0, 3);
    // ---- context: nh groups of nd lanes, each lane 4 channels; group h takes j = jlo+h, jlo+h+nh, ...
    // Only the band [jlo, jhi) of non-zero weights is read (an SMA alignment is a narrow band that
    // starts one-hot; exact zeros contribute nothing), with no branch inside the unrolled loop.
    {
        int lo = Tin, hi = 0;
        for (int j = tid; j < Tin; j += NT) if (an[j] != 0.f) { lo = min(lo, j); hi = max(hi, j + 1); }
        lo = -(int)block_reduce(-(float)lo, red, true);
        hi = (int)block_reduce((float)hi, red, true);
        const int h = tid / nd, dd = (tid % nd) * 4;
        if (h < nh) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* mp = st.memory + (long)b * Tin * E + dd;
#pragma unroll 8
            for (int j = lo + h; j < hi; j += nh) {
                const f32x4 mv = *reinterpret_cast<const f32x4*>(mp + (long)j * E);
                acc += an[j] * mv;
            }
            *reinterpret_cast<f32x4*>(cred + h * E + dd) = acc;
        }
        __syncthreads();
        T2_ASTAMP(0, 4);
        for (int c = tid; c < E; c += NT) {
            float sum = 0.f;
            for (int h2 = 0; h2 < nh; ++h2) sum += cred[h2 * E + c];
            st.ctx1[(long)b * st.ldctx1 + c] = sum;
            if (st.ctx2) st.ctx2[(long)b * st.ldctx2 + c] = sum;
            if (st.ctx16) st.ctx16[(long)b * st.ldctx16 + c] = (__bf16)sum;
            if (st.ctx16b) st.ctx16b[(long)b * st.ldctx16b + c] = (__bf16)sum;
        }
    }
    T2_ASTAMP(0, 5);
}


// ---------------------------------------------------------------------------------------------
// Backward of one StepwiseMonotonicAttention step (reverse time), one workgroup per (b, stream).
//   dctx   = direct sources + recurrent partials                      (saved: feeds d(memory) GEMM)
//   g_j    = dctx . memory_j + dalign_j + carry_j                     total gradient on a_t[j]
//   dp_j   = a_{t-1}[j] (g_j - g_{j+1}) ;  de_j = dp_j p_j (1 - p_j)
//   u_jk   = tanh(q_k + pm_jk) ;  dpre_jk = de_j v_k (1 - u_jk^2)
//   dq_k   = sum_j dpre_jk ; dv_k += sum_j de_j u_jk ; dpm_jk += dpre_jk
//   carry_out_j = g_j p_j + g_{j+1} (1 - p_j)                         gradient on a_{t-1}[j]
// ---------------------------------------------------------------------------------------------
constexpr int NTB = 512;     // 8 waves (1024 threads spill at 128 VGPRs); MAXI (= ceil(A/64)) sizes the per-lane dq/dv accumulators

// The memory positions of an item can be split over d.nsplit workgroups (blockIdx.z): each takes a contiguous range
// [jb, je), recomputes the (cheap) total ctx gradient, needs one extra g value at je for the recurrence, and emits its
// own partial of dq / dv (consumers add the partials).  With B*2 = 128 workgroups the kernel was bound by what one CU
// can pull from L2/MALL (~0.36 MB per item); two workgroups per item use all 256 CUs.
template <int MAXI>
__global__ __launch_bounds__(NTB) void attention_step_bwd_kernel(AttnBwdDesc d) {
    const AttnBwdStream& st = d.st[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, split = blockIdx.z;
    const int Tin = st.Tin, A = d.A, E = d.E;
    const int chunk = (((Tin + d.nsplit - 1) / d.nsplit) + 3) & ~3;
    const int jb = min(split * chunk, Tin), je = min(jb + chunk, Tin), len = je - jb;
    const int ng = je < Tin ? len + 1 : len;          // g values computed here: positions [jb, jb + ng)
    const int Tp = chunk + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dctx = smem;              // [E]
    float* q = dctx + E;             // [A]
    float* vs = q + A;               // [A]
    float* g = vs + A;               // [Tp]  g[jl] = g_{jb+jl}; g[len] = g_{je} (0 past the end)
    float* de = g + Tp;              // [Tp]
    float* ps = de + Tp;             // [Tp]
    float* red = ps + Tp;            // [NT/16][A] x 2 (dq, dv partials of the NT/16 position groups)

    T2_ASTAMP(8, 0);
    // the first batch of memory rows of the g pass (4 positions x 2 x 256 channels per wave) is requested before the
    // ctx gradient is assembled: the rows do not depend on it
    constexpr int GU = 4, GC = 2;
    f32x4 mpre[GC][GU];
    {
        const int wave = tid >> 6, lane = tid & 63;
        const int ngp = max((je < Tin ? je - jb + 1 : je - jb), 1);
#pragma unroll
        for (int cc = 0; cc < GC; ++cc)
#pragma unroll
            for (int u = 0; u < GU; ++u)
                mpre[cc][u] = *reinterpret_cast<const f32x4*>(st.memory + ((long)b * Tin + min(jb + min(wave + u * (NTB / 64), ngp - 1), Tin - 1)) * E + min(lane * 4 + 256 * cc, E - 4));
    }

    for (int c = tid; c < E; c += NTB) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) if (st.dctx[i]) v += st.dctx[i][(long)b * st.lddctx[i] + c];
        if (st.part && !d.first) {
            const float* p = st.part + (long)b * st.ldpart + st.part_col + c;
            float pv[8];                              // all K-split partials requested at once (nparts <= 8)
#pragma unroll
            for (int z = 0; z < 8; ++z) pv[z] = z < st.nparts ? p[(long)z * st.part_stride] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int z = 0; z < 8; ++z) acc += pv[z];
            v += acc;
        }
        dctx[c] = v;
        if (split == 0) st.dctx_out[(long)b * st.lddctx_out + c] = v;
    }
    for (int a = tid; a < A; a += NTB) { q[a] = st.q[(long)b * st.ldq + a]; vs[a] = st.v[a]; }
    for (int jl = tid; jl < len; jl += NTB) ps[jl] = st.p[(long)b * st.ldp + jb + jl];
    if (tid == 0) g[len] = 0.f;
    __syncthreads();
    if (len == 0) {                                   // (only when Tin is tiny) nothing to do but the partial outputs
        for (int a = tid; a < A; a += NTB) {
            st.dq_out[(long)b * st.lddq_out + split * A + a] = 0.f;
            float* dvp = st.dv_acc + ((long)split * d.B + b) * A + a;
            if (d.first) *dvp = 0.f;
        }
        return;
    }

    T2_ASTAMP(8, 1);
    // g_j: one wave per position, lanes stride the E channels 16 B at a time; 4 positions are in
    // flight per wave so that the row loads overlap instead of serialising on L2 latency
    {
        const int wave = tid >> 6, lane = tid & 63;
        constexpr int NWV = NTB / 64, U = 4;
        static_assert(U == GU, "prefetch shape");
        for (int j0 = wave; j0 < ng; j0 += NWV * U) {
            float sum[U] = {0.f, 0.f, 0.f, 0.f};
            int cc = 0;
            for (int c = lane * 4; c < E; c += 256, ++cc) {
                const f32x4 dc = *reinterpret_cast<const f32x4*>(dctx + c);
                f32x4 mv[U];
                const bool pre = j0 == wave && cc < GC;      // wave-uniform
#pragma unroll
                for (int u = 0; u < U; ++u) {           // clamp instead of branching: the U loads issue back to back
                    const int j = jb + min(j0 + u * NWV, ng - 1);
                    if (pre) mv[u] = cc == 0 ? mpre[0][u] : mpre[1][u];
                    else mv[u] = *reinterpret_cast<const f32x4*>(st.memory + ((long)b * Tin + j) * E + c);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) sum[u] += mv[u][0] * dc[0] + mv[u][1] * dc[1] + mv[u][2] * dc[2] + mv[u][3] * dc[3];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jl = j0 + u * NWV;
                const float tot = wave_sum(sum[u]);
                if (lane == 0 && jl < ng) {
                    const int j = jb + jl;
                    float gsum = tot;
                    if (st.dalign) gsum += st.dalign[(long)b * st.lddalign + j];
                    if (!d.first) gsum += st.carry[(long)b * Tin + j];
                    g[jl] = gsum;
                }
            }
        }
    }
    __syncthreads();
    T2_ASTAMP(8, 2);
    // carry is read at position je too, which the next split's workgroup updates: the new carry therefore goes to a
    // second buffer (carry / carry_out swap roles every step)
    for (int jl = tid; jl < len; jl += NTB) {
        const int j = jb + jl;
        const float ap = st.a_prev ? st.a_prev[(long)b * st.lda_prev + j] : (j == 0 ? 1.f : 0.f);
        const float p = ps[jl];
        const float gj = g[jl], gn = g[jl + 1];
        de[jl] = ap * (gj - gn) * p * (1.0f - p);
        st.carry_out[(long)b * Tin + j] = gj * p + gn * (1.0f - p);
    }
    __syncthreads();

    T2_ASTAMP(8, 3);
    // energies backward: 16 lanes per position, each lane owns channels sub*4 + 64*i
    {
        const int gid = tid >> 4, sub = tid & 15;
        float dq[MAXI][4], dv[MAXI][4];
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) { dq[i][c] = 0.f; dv[i][c] = 0.f; }
        for (int jl = gid; jl < len; jl += NTB / 16) {
            const float dej = de[jl];
            const float* pmr = st.pm + ((long)b * Tin + jb + jl) * A;
            float* dpr = st.dpm_acc + ((long)b * Tin + jb + jl) * A;
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int a = sub * 4 + 64 * i;
                if (a < A) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(pmr + a);
                    f32x4 acc = d.first ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(dpr + a);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float u = tanhf(q[a + c] + pv[c]);
                        const float dpre = dej * vs[a + c] * (1.0f - u * u);
                        dq[i][c] += dpre;
                        dv[i][c] += dej * u;
                        acc[c] += dpre;
                    }
                    *reinterpret_cast<f32x4*>(dpr + a) = acc;
                }
            }
        }
        T2_ASTAMP(8, 4);
        float* rq = red;
        float* rv = red + (NTB / 16) * A;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int a = sub * 4 + 64 * i;
            if (a < A) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { rq[gid * A + a + c] = dq[i][c]; rv[gid * A + a + c] = dv[i][c]; }
            }
        }
        __syncthreads();
        for (int a = tid; a < A; a += NTB) {
            float sq = 0.f, sv = 0.f;
            for (int k = 0; k < NTB / 16; ++k) { sq += rq[k * A + a]; sv += rv[k * A + a]; }
            st.dq_out[(long)b * st.lddq_out + split * A + a] = sq;
            float* dvp = st.dv_acc + ((long)split * d.B + b) * A + a;
            *dvp = (d.first ? 0.f : *dvp) + sv;
        }
    }
    T2_ASTAMP(8, 5);
}

// ---------------------------------------------------------------------------------------------
// Backward of one LocationSensitiveAttention step (reverse time), one workgroup per (b, stream).
//   forward:  loc_jf = sum_{c,k} Wc[f][c][k] wcat[c][j+k-pad]      wcat = [w_{t-1} ; cum_{t-1}]
//             u_ja = tanh(q_a + sum_f Wd[a][f] loc_jf + pm_ja) ; e_j = v . u_j ; w = softmax(e)
//             cum_t = cum_{t-1} + w ; ctx = w . memory
//   backward: g_j  = dctx . memory_j + dalign_j + carry_w_j + carry_cum_j       total gradient on w_t[j]
//             de_j = w_j (g_j - sum_k w_k g_k)                                   softmax
//             dpre_ja = de_j v_a (1 - u_ja^2) ; dq_a = sum_j dpre_ja ; dv_a += sum_j de_j u_ja ; dpm_ja += dpre_ja
//             dWd[a][f] += sum_j dpre_ja loc_jf ; dloc_jf = sum_a dpre_ja Wd[a][f]
//             dWc[f][c][k] += sum_j dloc_jf wcat[c][j+k-pad]
//             dwcat[c][i] = sum_{f,k} Wc[f][c][k] dloc_{i-k+pad, f}
//             carry_w <- dwcat[0] (gradient on w_{t-1}) ; carry_cum <- carry_cum + dwcat[1] (gradient on cum_{t-1})
// Positions are processed 32 at a time (16 lanes each); the per-chunk dpre tile goes through LDS
// for the two location-layer contractions.  Fixed summation orders everywhere (no atomics).
// ---------------------------------------------------------------------------------------------
constexpr int NPG = NTB / 16;     // positions per chunk
constexpr int NQ = 4;             // f-groups of the dwcat contraction

struct LsaBwdSmem { int dctx, q, vs, g, w, de, wpad, convw, dense, dlocP, X, dpreS, loc, red2, tmp, total; };
__host__ __device__ inline LsaBwdSmem lsa_bwd_smem(int Tin, int A, int E, int F, int Kc) {
    LsaBwdSmem m; int o = 0;
    const int Tp = (Tin + 3) & ~3, Tw = Tin + Kc - 1;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.dctx = take(E); m.q = take(A); m.vs = take(A); m.g = take(Tp); m.w = take(Tp); m.de = take(Tp);
    m.wpad = take(2 * Tw); m.convw = take(F * 2 * Kc); m.dense = take(A * (F + 1));
    m.dlocP = take(Tw * (F + 1));
    const int xs = NPG * (A + 4) + Tin * (F + 1), rs = 2 * NPG * A;     // dpre tile + loc, later reused for the dq/dv partials
    m.X = take(xs > rs ? xs : rs); m.dpreS = m.X; m.loc = m.X + NPG * (A + 4);
    m.red2 = take(NTB / 64 + 4); m.tmp = take(NQ * 2 * Tin);
    m.total = o;
    return m;
}

__device__ __forceinline__ float block_sum_b(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NTB / 64; ++i) r += red[i];
    return r;
}

template <int MAXI>
__global__ __launch_bounds__(NTB) void attention_lsa_step_bwd_kernel(AttnBwdDesc d) {
    const AttnBwdStream& st = d.st[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Tin = st.Tin, A = d.A, E = d.E, F = d.F, Kc = d.Kc;
    const int pad = (Kc - 1) / 2, Tw = Tin + Kc - 1, F1 = F + 1, AS = A + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LsaBwdSmem m = lsa_bwd_smem(Tin, A, E, F, Kc);
    float* dctx = smem + m.dctx; float* q = smem + m.q; float* vs = smem + m.vs;
    float* g = smem + m.g; float* wS = smem + m.w; float* de = smem + m.de;
    float* wpad = smem + m.wpad; float* convw = smem + m.convw; float* dense = smem + m.dense;
    float* dlocP = smem + m.dlocP; float* dpreS = smem + m.dpreS; float* loc = smem + m.loc;
    float* red2 = smem + m.red2; float* tmp = smem + m.tmp;

    // ---- stage: total ctx gradient, query, v, weights of this step, location-layer weights, padded conv input
    for (int c = tid; c < E; c += NTB) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) if (st.dctx[i]) v += st.dctx[i][(long)b * st.lddctx[i] + c];
        if (st.part && !d.first) {
            const float* p = st.part + (long)b * st.ldpart + st.part_col + c;
            float pv[8];
#pragma unroll
            for (int z = 0; z < 8; ++z) pv[z] = z < st.nparts ? p[(long)z * st.part_stride] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int z = 0; z < 8; ++z) acc += pv[z];
            v += acc;
        }
        dctx[c] = v;
        st.dctx_out[(long)b * st.lddctx_out + c] = v;
    }
    for (int a = tid; a < A; a += NTB) { q[a] = st.q[(long)b * st.ldq + a]; vs[a] = st.v[a]; }
    for (int j = tid; j < Tin; j += NTB) wS[j] = st.w[(long)b * st.ldw + j];
    for (int i = tid; i < F * 2 * Kc; i += NTB) convw[i] = st.loc_conv[i];
    for (int i = tid; i < A * F; i += NTB) dense[(i / F) * F1 + (i % F)] = st.loc_dense[i];
    for (int i = tid; i < 2 * Tw; i += NTB) {
        const int c = i / Tw, j = i % Tw - pad;
        float v = 0.f;
        if (j >= 0 && j < Tin) v = c == 0 ? (st.a_prev ? st.a_prev[(long)b * st.lda_prev + j] : 0.f)
                                          : (st.wcum_prev ? st.wcum_prev[(long)b * st.ldwcum_prev + j] : 0.f);
        wpad[i] = v;
    }
    for (int i = tid; i < Tw * F1; i += NTB) dlocP[i] = 0.f;          // rows [pad, pad+Tin) are overwritten below
    __syncthreads();

    // ---- g_j = dctx . memory_j (+ external and carried gradients): one wave per position, 4 in flight
    {
        const int wave = tid >> 6, lane = tid & 63;
        constexpr int NWV = NTB / 64, U = 4;
        for (int j0 = wave; j0 < Tin; j0 += NWV * U) {
            float sum[U] = {0.f, 0.f, 0.f, 0.f};
            for (int c = lane * 4; c < E; c += 256) {
                const f32x4 dc = *reinterpret_cast<const f32x4*>(dctx + c);
                f32x4 mv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = min(j0 + u * NWV, Tin - 1);
                    mv[u] = *reinterpret_cast<const f32x4*>(st.memory + ((long)b * Tin + j) * E + c);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) sum[u] += mv[u][0] * dc[0] + mv[u][1] * dc[1] + mv[u][2] * dc[2] + mv[u][3] * dc[3];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * NWV;
                const float tot = wave_sum(sum[u]);
                if (lane == 0 && j < Tin) {
                    float gsum = tot;
                    if (st.dalign) gsum += st.dalign[(long)b * st.lddalign + j];
                    if (!d.first) gsum += st.carry[(long)b * Tin + j] + st.carry_cum[(long)b * Tin + j];
                    g[j] = gsum;
                }
            }
        }
    }
    // ---- location conv recomputed (same loop order as the forward kernel)
    for (int i = tid; i < Tin * F; i += NTB) {
        const int j = i / F, f = i % F;
        float sum = 0.f;
        for (int c = 0; c < 2; ++c) {
            const float* w = convw + (f * 2 + c) * Kc;
            const float* x = wpad + c * Tw + j;
            for (int k = 0; k < Kc; ++k) sum += w[k] * x[k];
        }
        loc[j * F1 + f] = sum;
    }
    __syncthreads();
    // ---- softmax backward
    {
        float part = 0.f;
        for (int j = tid; j < Tin; j += NTB) part += wS[j] * g[j];
        const float sdot = block_sum_b(part, red2);
        for (int j = tid; j < Tin; j += NTB) de[j] = wS[j] * (g[j] - sdot);
    }
    __syncthreads();

    // ---- energies backward + location-layer contractions, NPG positions per chunk
    const int gid = tid >> 4, sub = tid & 15;
    float dq[MAXI][4], dv[MAXI][4], dd[4 * MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { dq[i][c] = 0.f; dv[i][c] = 0.f; }
#pragma unroll
    for (int k = 0; k < 4 * MAXI; ++k) dd[k] = 0.f;
    for (int j0 = 0; j0 < Tin; j0 += NPG) {
        const int j = j0 + gid;
        const bool valid = j < Tin;
        const int jc = valid ? j : Tin - 1;
        const float dej = valid ? de[jc] : 0.f;
        const float* pmr = st.pm + ((long)b * Tin + jc) * A;
        float* dpr = st.dpm_acc + ((long)b * Tin + jc) * A;
        const float* lr = loc + jc * F1;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int a = sub * 4 + 64 * i;
            if (a < A) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(pmr + a);
                f32x4 acc = d.first ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(dpr + a);
                float pa[4] = {0.f, 0.f, 0.f, 0.f};
                for (int f = 0; f < F; ++f) {
                    const float lv = lr[f];
#pragma unroll
                    for (int c = 0; c < 4; ++c) pa[c] += dense[(a + c) * F1 + f] * lv;
                }
                f32x4 dp;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float u = tanhf(q[a + c] + pa[c] + pv[c]);
                    const float dpre = dej * vs[a + c] * (1.0f - u * u);
                    dq[i][c] += dpre;
                    dv[i][c] += dej * u;
                    acc[c] += dpre;
                    dp[c] = dpre;
                }
                if (valid) *reinterpret_cast<f32x4*>(dpr + a) = acc;
                *reinterpret_cast<f32x4*>(dpreS + gid * AS + a) = dp;
            }
        }
        __syncthreads();
        const int nj = min(NPG, Tin - j0);
        // dloc rows of this chunk
        for (int it = tid; it < nj * F; it += NTB) {
            const int jj = it / F, f = it % F;
            const float* dr = dpreS + jj * AS;
            float sum = 0.f;
            for (int a = 0; a < A; ++a) sum += dr[a] * dense[a * F1 + f];
            dlocP[(j0 + jj + pad) * F1 + f] = sum;
        }
        // d(location_dense): thread-owned outputs o = tid + k*NTB  ->  (a, f) = (o / F, o % F)
#pragma unroll
        for (int k = 0; k < 4 * MAXI; ++k) {
            const int o = tid + k * NTB;
            if (o < A * F) {
                const int a = o / F, f = o % F;
                float sum = dd[k];
                for (int jj = 0; jj < nj; ++jj) sum += dpreS[jj * AS + a] * loc[(j0 + jj) * F1 + f];
                dd[k] = sum;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4 * MAXI; ++k) {
        const int o = tid + k * NTB;
        if (o < A * F) {
            float* p = st.ddense_acc + (long)b * A * F + o;
            *p = (d.first ? 0.f : *p) + dd[k];
        }
    }
    // dq / dv: reduce the NPG position groups (the dpre tile + loc region is free now)
    {
        float* rq = smem + m.X;
        float* rv = rq + NPG * A;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int a = sub * 4 + 64 * i;
            if (a < A) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { rq[gid * A + a + c] = dq[i][c]; rv[gid * A + a + c] = dv[i][c]; }
            }
        }
        __syncthreads();
        for (int a = tid; a < A; a += NTB) {
            float sq = 0.f, sv = 0.f;
            for (int k = 0; k < NPG; ++k) { sq += rq[k * A + a]; sv += rv[k * A + a]; }
            st.dq_out[(long)b * st.lddq_out + a] = sq;
            float* dvp = st.dv_acc + (long)b * A + a;
            *dvp = (d.first ? 0.f : *dvp) + sv;
        }
    }
    // d(location_conv)[f][c][k] += sum_j dloc_jf wcat[c][j+k-pad]
    for (int it = tid; it < F * 2 * Kc; it += NTB) {
        const int f = it / (2 * Kc), c = (it / Kc) % 2, k = it % Kc;
        const float* x = wpad + c * Tw + k;
        float sum = 0.f;
        for (int j = 0; j < Tin; ++j) sum += dlocP[(j + pad) * F1 + f] * x[j];
        float* p = st.dconv_acc + (long)b * F * 2 * Kc + it;
        *p = (d.first ? 0.f : *p) + sum;
    }
    // gradient on the conv input: dwcat[c][i] = sum_{f,k} Wc[f][c][k] dloc[i-k+pad][f]   (zero-padded rows)
    if (st.a_prev) {
        for (int it = tid; it < NQ * 2 * Tin; it += NTB) {
            const int fq = it / (2 * Tin), c = (it / Tin) % 2, i = it % Tin;
            float sum = 0.f;
            for (int f = fq; f < F; f += NQ) {
                const float* w = convw + (f * 2 + c) * Kc;
                const float* dl = dlocP + (i + 2 * pad) * F1 + f;
                for (int k = 0; k < Kc; ++k) sum += w[k] * dl[-k * F1];
            }
            tmp[it] = sum;
        }
        __syncthreads();
        for (int it = tid; it < 2 * Tin; it += NTB) {
            float sum = 0.f;
#pragma unroll
            for (int fq = 0; fq < NQ; ++fq) sum += tmp[fq * 2 * Tin + it];
            const int c = it / Tin, i = it % Tin;
            if (c == 0) st.carry[(long)b * Tin + i] = sum;
            else {
                float* p = st.carry_cum + (long)b * Tin + i;
                *p = (d.first ? 0.f : *p) + sum;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// LSA backward, matrix-core variant (used when the [T_in][A] tile fits in LDS; same math and outputs as
// attention_lsa_step_bwd_kernel above).  The four contractions of the location layer run as fp32 MFMAs
// (v_mfma_f32_32x32x2_f32, exact fma chains) on one [T_in][A] LDS tile that first holds
// pa = loc . Wd^T and is then overwritten in place by dpre:
//     pa     [Tin x A]  = loc  [Tin x F] . Wd^T           dloc [Tin x F] = dpre [Tin x A] . Wd
//     dWd    [A x F]   += dpre^T . loc                     dWc  [F x 2Kc] += dloc^T . toeplitz(wcat)
// 1024 threads: 16 waves; the scalar version spent ~110 us per step on LDS operand reads.
// ---------------------------------------------------------------------------------------------
constexpr int NTL = 1024;
constexpr int NQ2 = 8;            // f-groups of the dwcat contraction

struct LsaMfmaSmem { int dctx, q, vs, g, w, de, wpad, convw, dense, loc, dlocP, PD, red, red2, tmp, total; int TwP, TinE, PA, F1; };
__host__ __device__ inline LsaMfmaSmem lsa_mfma_smem(int Tin, int A, int E, int F, int Kc) {
    LsaMfmaSmem m; int o = 0;
    const int Tp = (Tin + 3) & ~3;
    m.TwP = (Tin + Kc - 1 + 4 + 3) & ~3; m.TinE = (Tin + 1) & ~1; m.PA = A + 8; m.F1 = F + 1;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    m.dctx = take(E); m.q = take(A); m.vs = take(A); m.g = take(Tp); m.w = take(Tp); m.de = take(Tp);
    m.wpad = take(2 * m.TwP); m.convw = take(F * 2 * Kc); m.dense = take(A * m.F1);
    m.loc = take(m.TinE * m.F1); m.dlocP = take((Tin + Kc - 1 + 4) * m.F1); m.PD = take(m.TinE * m.PA);
    const int nred = (NTL / 16) * A, ntmp = NQ2 * 2 * Tp;            // tmp is written after red has been consumed
    m.red = take(nred > ntmp ? nred : ntmp); m.tmp = m.red; m.red2 = take(NTL / 64 + 4);
    m.total = o;
    return m;
}

__device__ __forceinline__ float block_sum_l(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NTL / 64; ++i) r += red[i];
    return r;
}

template <int MAXI>
__global__ __launch_bounds__(NTL) void attention_lsa_step_bwd_mfma_kernel(AttnBwdDesc d) {
    const AttnBwdStream& st = d.st[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int Tin = st.Tin, A = d.A, E = d.E, F = d.F, Kc = d.Kc;
    const int pad = (Kc - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LsaMfmaSmem m = lsa_mfma_smem(Tin, A, E, F, Kc);
    const int TwP = m.TwP, TinE = m.TinE, PA = m.PA, F1 = m.F1;
    float* dctx = smem + m.dctx; float* q = smem + m.q; float* vs = smem + m.vs;
    float* g = smem + m.g; float* wS = smem + m.w; float* de = smem + m.de;
    float* wpad = smem + m.wpad; float* convw = smem + m.convw; float* dense = smem + m.dense;
    float* loc = smem + m.loc; float* dlocP = smem + m.dlocP; float* PD = smem + m.PD;
    float* red = smem + m.red; float* red2 = smem + m.red2; float* tmp = smem + m.tmp;

    // ---- stage
    for (int c = tid; c < E; c += NTL) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) if (st.dctx[i]) v += st.dctx[i][(long)b * st.lddctx[i] + c];
        if (st.part && !d.first) {
            const float* p = st.part + (long)b * st.ldpart + st.part_col + c;
            float pv[8];
#pragma unroll
            for (int z = 0; z < 8; ++z) pv[z] = z < st.nparts ? p[(long)z * st.part_stride] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int z = 0; z < 8; ++z) acc += pv[z];
            v += acc;
        }
        dctx[c] = v;
        st.dctx_out[(long)b * st.lddctx_out + c] = v;
    }
    for (int a = tid; a < A; a += NTL) { q[a] = st.q[(long)b * st.ldq + a]; vs[a] = st.v[a]; }
    for (int j = tid; j < Tin; j += NTL) wS[j] = st.w[(long)b * st.ldw + j];
    for (int i = tid; i < F * 2 * Kc; i += NTL) convw[i] = st.loc_conv[i];
    for (int i = tid; i < A * F1; i += NTL) dense[i] = (i % F1) < F ? st.loc_dense[(i / F1) * F + (i % F1)] : 0.f;
    for (int i = tid; i < 2 * TwP; i += NTL) {
        const int c = i / TwP, j = i % TwP - pad;
        float v = 0.f;
        if (j >= 0 && j < Tin) v = c == 0 ? (st.a_prev ? st.a_prev[(long)b * st.lda_prev + j] : 0.f)
                                          : (st.wcum_prev ? st.wcum_prev[(long)b * st.ldwcum_prev + j] : 0.f);
        wpad[i] = v;
    }
    for (int i = tid; i < (Tin + Kc - 1 + 4) * F1; i += NTL) dlocP[i] = 0.f;     // rows [pad, pad+Tin) are overwritten below
    for (int i = tid; i < (TinE - Tin) * PA; i += NTL) PD[Tin * PA + i] = 0.f;  // K of the dWd product is rounded up to even
    for (int i = tid; i < (TinE - Tin) * F1; i += NTL) loc[Tin * F1 + i] = 0.f;
    __syncthreads();

    // ---- g_j = dctx . memory_j (+ external and carried gradients): one wave per position, 4 in flight
    {
        constexpr int NWV = NTL / 64, U = 4;
        for (int j0 = wave; j0 < Tin; j0 += NWV * U) {
            float sum[U] = {0.f, 0.f, 0.f, 0.f};
            for (int c = lane * 4; c < E; c += 256) {
                const f32x4 dc = *reinterpret_cast<const f32x4*>(dctx + c);
                f32x4 mv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = min(j0 + u * NWV, Tin - 1);
                    mv[u] = *reinterpret_cast<const f32x4*>(st.memory + ((long)b * Tin + j) * E + c);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) sum[u] += mv[u][0] * dc[0] + mv[u][1] * dc[1] + mv[u][2] * dc[2] + mv[u][3] * dc[3];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * NWV;
                const float tot = wave_sum(sum[u]);
                if (lane == 0 && j < Tin) {
                    float gsum = tot;
                    if (st.dalign) gsum += st.dalign[(long)b * st.lddalign + j];
                    if (!d.first) gsum += st.carry[(long)b * Tin + j] + st.carry_cum[(long)b * Tin + j];
                    g[j] = gsum;
                }
            }
        }
    }
    // ---- location conv recomputed: one thread per (f, 4 positions), sliding window in registers
    {
        const int nj4 = (Tin + 3) / 4;
        for (int i = tid; i < nj4 * F; i += NTL) {
            const int f = i % F, j0 = (i / F) * 4;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int c = 0; c < 2; ++c) {
                const float* w = convw + (f * 2 + c) * Kc;
                const float* x = wpad + c * TwP + j0;
                float x0 = x[0], x1 = x[1], x2 = x[2];
                for (int k = 0; k < Kc; ++k) {
                    const float x3 = x[k + 3], wk = w[k];
                    a0 += wk * x0; a1 += wk * x1; a2 += wk * x2; a3 += wk * x3;
                    x0 = x1; x1 = x2; x2 = x3;
                }
            }
            loc[j0 * F1 + f] = a0;
            if (j0 + 1 < Tin) loc[(j0 + 1) * F1 + f] = a1;
            if (j0 + 2 < Tin) loc[(j0 + 2) * F1 + f] = a2;
            if (j0 + 3 < Tin) loc[(j0 + 3) * F1 + f] = a3;
        }
        for (int j = tid; j < Tin; j += NTL) loc[j * F1 + F] = 0.f;
    }
    __syncthreads();
    const int njt = (Tin + 31) / 32, nat = A / 32;
    // ---- pa = loc . Wd^T  -> PD
    {
        const int Ke = (F + 1) & ~1;
        for (int tile = wave; tile < njt * nat; tile += NTL / 64) {
            const int jt = tile / nat, at = tile % nat;
            const float* lr = loc + min(jt * 32 + r, Tin - 1) * F1 + h;
            const float* dr = dense + (at * 32 + r) * F1 + h;
            f32x16 acc;
#pragma unroll
            for (int e2 = 0; e2 < 16; ++e2) acc[e2] = 0.f;
            for (int kk = 0; kk < Ke; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lr[kk], dr[kk], acc, 0, 0, 0);
#pragma unroll
            for (int e2 = 0; e2 < 16; ++e2) {
                const int row = jt * 32 + (e2 & 3) + 8 * (e2 >> 2) + 4 * h;
                if (row < Tin) PD[row * PA + at * 32 + r] = acc[e2];
            }
        }
    }
    // ---- softmax backward
    {
        float part = 0.f;
        for (int j = tid; j < Tin; j += NTL) part += wS[j] * g[j];
        const float sdot = block_sum_l(part, red2);          // (its barriers also publish PD)
        for (int j = tid; j < Tin; j += NTL) de[j] = wS[j] * (g[j] - sdot);
    }
    __syncthreads();

    // ---- energies backward: 16 lanes per position; dpre overwrites pa in place
    {
        const int gid = tid >> 4, sub = tid & 15;
        float dv[MAXI][4];
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) dv[i][c] = 0.f;
        for (int j = gid; j < Tin; j += NTL / 16) {
            const float dej = de[j];
            const float* pmr = st.pm + ((long)b * Tin + j) * A;
            float* dpr = st.dpm_acc + ((long)b * Tin + j) * A;
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int a = sub * 4 + 64 * i;
                if (a < A) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(pmr + a);
                    f32x4 acc = d.first ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(dpr + a);
                    const f32x4 pa = *reinterpret_cast<const f32x4*>(PD + j * PA + a);
                    f32x4 dp;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float u = tanhf(q[a + c] + pa[c] + pv[c]);
                        const float dpre = dej * vs[a + c] * (1.0f - u * u);
                        dv[i][c] += dej * u;
                        acc[c] += dpre;
                        dp[c] = dpre;
                    }
                    *reinterpret_cast<f32x4*>(dpr + a) = acc;
                    *reinterpret_cast<f32x4*>(PD + j * PA + a) = dp;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int a = sub * 4 + 64 * i;
            if (a < A) {
#pragma unroll
                for (int c = 0; c < 4; ++c) red[gid * A + a + c] = dv[i][c];
            }
        }
    }
    __syncthreads();
    // ---- four independent consumers of the dpre tile, spread over the waves:
    //   waves [0, njt):            dloc[j][f] = sum_a dpre[j][a] Wd[a][f]                (MFMA, K = A)
    //   waves [njt, njt + nat):    dWd[a][f] += sum_j dpre[j][a] loc[j][f]               (MFMA, K = Tin)
    //   remaining waves:           dq[a] = sum_j dpre[j][a] ; dv[a] += sum of the group partials
    if (wave < njt) {
        const int jt = wave;
        const float* ar = PD + min(jt * 32 + r, TinE - 1) * PA + h;       // A[m = j][k = a]
        const float* br = dense + h * F1 + min(r, F);                      // B[k = a][n = f]   (column F is the zero pad)
        f32x16 acc;
#pragma unroll
        for (int e2 = 0; e2 < 16; ++e2) acc[e2] = 0.f;
        for (int kk = 0; kk < A; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kk], br[kk * F1], acc, 0, 0, 0);
        if (r < F) {
#pragma unroll
            for (int e2 = 0; e2 < 16; ++e2) {
                const int row = jt * 32 + (e2 & 3) + 8 * (e2 >> 2) + 4 * h;
                if (row < Tin) dlocP[(row + pad) * F1 + r] = acc[e2];
            }
        }
    } else if (wave < njt + nat) {
        const int at = wave - njt;
        const float* ar = PD + h * PA + at * 32 + r;                       // A[m = a][k = j]
        const float* br = loc + h * F1 + min(r, F);                        // B[k = j][n = f]
        f32x16 acc;
#pragma unroll
        for (int e2 = 0; e2 < 16; ++e2) acc[e2] = 0.f;
        for (int kk = 0; kk < TinE; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kk * PA], br[kk * F1], acc, 0, 0, 0);
        if (r < F) {
#pragma unroll
            for (int e2 = 0; e2 < 16; ++e2) {
                const int a = at * 32 + (e2 & 3) + 8 * (e2 >> 2) + 4 * h;
                float* p = st.ddense_acc + (long)b * A * F + a * F + r;
                *p = (d.first ? 0.f : *p) + acc[e2];
            }
        }
    } else {
        const int t0 = tid - (njt + nat) * 64, nth = NTL - (njt + nat) * 64;
        for (int a = t0; a < A; a += nth) {
            float sq = 0.f, sv = 0.f;
            for (int j = 0; j < Tin; ++j) sq += PD[j * PA + a];
            for (int k = 0; k < NTL / 16; ++k) sv += red[k * A + a];
            st.dq_out[(long)b * st.lddq_out + a] = sq;
            float* dvp = st.dv_acc + (long)b * A + a;
            *dvp = (d.first ? 0.f : *dvp) + sv;
        }
    }
    __syncthreads();
    // ---- dWc[f][(c,k)] += sum_j dloc[j][f] wcat[c][j+k-pad]   (MFMA, K = Tin; waves 0..1)  |  dwcat on the other waves
    const int ncol = 2 * Kc, nct = (ncol + 31) / 32;
    if (wave < nct) {
        const int n = min(wave * 32 + r, ncol - 1), c = n / Kc, k = n % Kc;
        const float* ar = dlocP + (pad + h) * F1 + min(r, F);              // A[m = f][k = j]   (row F... column F is never written: 0)
        const float* br = wpad + c * TwP + k + h;                          // B[k = j][n = (c,k)] = wcat[c][j + k - pad]
        f32x16 acc;
#pragma unroll
        for (int e2 = 0; e2 < 16; ++e2) acc[e2] = 0.f;
        for (int kk = 0; kk < TinE; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kk * F1], br[kk], acc, 0, 0, 0);
        if (wave * 32 + r < ncol) {
#pragma unroll
            for (int e2 = 0; e2 < 16; ++e2) {
                const int f = (e2 & 3) + 8 * (e2 >> 2) + 4 * h;
                if (f < F) {
                    float* p = st.dconv_acc + (long)b * F * ncol + f * ncol + n;
                    *p = (d.first ? 0.f : *p) + acc[e2];
                }
            }
        }
    } else if (st.a_prev) {
        // dwcat[c][i] = sum_{f,k} Wc[f][c][k] dloc[i-k+pad][f]: work item (fq, c, 4 positions), window slides downwards
        const int t0 = tid - nct * 64, nth = NTL - nct * 64, ni4 = (Tin + 3) / 4, Tp = (Tin + 3) & ~3;
        for (int it = t0; it < NQ2 * 2 * ni4; it += nth) {
            const int fq = it / (2 * ni4), c = (it / ni4) % 2, i0 = (it % ni4) * 4;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int f = fq; f < F; f += NQ2) {
                const float* w = convw + (f * 2 + c) * Kc;
                const float* dl = dlocP + (i0 + 2 * pad) * F1 + f;         // row i0 - k + 2*pad at k = 0
                float d3 = dl[3 * F1], d2 = dl[2 * F1], d1 = dl[F1];       // rows i0+3, i0+2, i0+1 (zero rows past the end)
                for (int k = 0; k < Kc; ++k) {
                    const float d0 = dl[-k * F1], wk = w[k];
                    s0 += wk * d0; s1 += wk * d1; s2 += wk * d2; s3 += wk * d3;
                    d3 = d2; d2 = d1; d1 = d0;
                }
            }
            float* tp = tmp + (fq * 2 + c) * Tp + i0;
            tp[0] = s0; tp[1] = s1; tp[2] = s2; tp[3] = s3;
        }
    }
    __syncthreads();
    if (st.a_prev) {
        const int Tp = (Tin + 3) & ~3;
        for (int it = tid; it < 2 * Tin; it += NTL) {
            const int c = it / Tin, i = it % Tin;
            float sum = 0.f;
#pragma unroll
            for (int fq = 0; fq < NQ2; ++fq) sum += tmp[(fq * 2 + c) * Tp + i];
            if (c == 0) st.carry[(long)b * Tin + i] = sum;
            else {
                float* p = st.carry_cum + (long)b * Tin + i;
                *p = (d.first ? 0.f : *p) + sum;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// GMMAttention, version '2' (attention.py:401-506): purely location-based.
//   hid = tanh(W1 h + b1) ; [omega^, delta^, sigma^] = W2 hid + b2            (3 x K, K = 5)
//   sigma = softplus(sigma^) + 1e-5 ; delta = softplus(delta^) ; omega = softmax(omega^) ; Z = sqrt(2 pi sigma^2)
//   mu_t = mu_{t-1} + delta ; phi_j = sum_k omega_k / Z_k exp(-(j - mu_k)^2 / sigma_k^2 / 2)
//   w = softmax(mask(phi)) ; ctx = w . memory
// W1 h arrives as the ordered partials of the LSTM step kernel (the same path as the query projection of the other
// attention kinds), so only the tiny second layer runs here.  One workgroup per (b, stream).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // torch's threshold

struct GmmPar { float omega[kGmmK], sigma[kGmmK], delta[kGmmK], Z[kGmmK]; };
// ip: [3K] = omega^ | delta^ | sigma^ (interm_params.view(B, 3, K), attention.py:437-443)
__device__ __forceinline__ GmmPar gmm_params(const float* ip) {
    GmmPar p;
    float mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int k = 0; k < kGmmK; ++k) mx = fmaxf(mx, ip[k]);
#pragma unroll
    for (int k = 0; k < kGmmK; ++k) { p.omega[k] = expf(ip[k] - mx); sum += p.omega[k]; }
#pragma unroll
    for (int k = 0; k < kGmmK; ++k) {
        p.omega[k] /= sum;
        p.delta[k] = softplusf_(ip[kGmmK + k]);
        p.sigma[k] = softplusf_(ip[2 * kGmmK + k]) + 1e-5f;
        p.Z[k] = sqrtf(2.0f * 3.14159265358979323846f * p.sigma[k] * p.sigma[k]);
    }
    return p;
}

__global__ __launch_bounds__(NT) void attention_gmm_step_fwd_kernel(AttnStepDesc d) {
    const AttnStream& st = d.st[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Tin = st.Tin, A = d.A, E = d.E;
    const int Tp = (Tin + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hid = smem;               // [A]
    float* e = hid + A;              // [Tp]
    float* an = e + Tp;              // [Tp]
    float* red = an + Tp;            // [4*NT]
    const int nd = E / 4, nh = NT / nd;
    float* cred = red + 4 * NT;      // [nh*E]
    float* ip = cred + nh * E;       // [16]
    float* par = ip + 16;            // [4][8]: c = omega/Z, mu, sigma^2

    // ---- first MLP layer: ordered sum of the partials + bias, tanh
    {
        const int a4n = A / 4, ng = NT / a4n;
        const int pg = tid / a4n, a4 = (tid % a4n) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* p = st.qpart + (long)b * A + a4;
        const long ps = (long)d.B * A;
#pragma unroll 4
        for (int i = pg; i < st.nparts; i += ng) acc += *reinterpret_cast<const f32x4*>(p + (long)i * ps);
        *reinterpret_cast<f32x4*>(red + pg * A + a4) = acc;
        __syncthreads();
        if (tid < A) {
            float sum = 0.f;
            const int used = st.nparts < ng ? st.nparts : ng;
            for (int h2 = 0; h2 < used; ++h2) sum += red[h2 * A + tid];
            sum += st.gmm_b1[tid];
            if (st.q_out) st.q_out[(long)b * st.ldq_out + tid] = sum;      // pre-activation, saved for backward
            hid[tid] = tanhf(sum);
        }
    }
    __syncthreads();
    // ---- second layer: 3K rows, one wave each
    for (int i = wave; i < 3 * kGmmK; i += NT / 64) {
        float sum = 0.f;
        for (int a = lane; a < A; a += 64) sum += st.gmm_w2[(long)i * A + a] * hid[a];
        sum = wave_sum(sum);
        if (lane == 0) ip[i] = sum + st.gmm_b2[i];
    }
    __syncthreads();
    if (tid == 0) {
        const GmmPar p = gmm_params(ip);
#pragma unroll
        for (int k = 0; k < kGmmK; ++k) {
            const float mu = (st.mu_prev ? st.mu_prev[(long)b * kGmmPad + k] : 0.f) + p.delta[k];
            st.mu_out[(long)b * kGmmPad + k] = mu;
            par[k] = p.omega[k] / p.Z[k]; par[8 + k] = mu; par[16 + k] = p.sigma[k] * p.sigma[k];
        }
    }
    __syncthreads();
    // ---- mixture, mask, softmax
    const int len = st.lengths ? st.lengths[b] : Tin;
    float mx = -INFINITY;
    for (int j = tid; j < Tin; j += NT) {
        float phi = 0.f;
#pragma unroll
        for (int k = 0; k < kGmmK; ++k) {
            const float dj = (float)j - par[8 + k];
            phi += par[k] * expf(-(dj * dj) / par[16 + k] / 2.0f);
        }
        if (j >= len) phi = st.mask_value;
        e[j] = phi;
        mx = fmaxf(mx, phi);
    }
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
    for (int j = tid; j < Tin; j += NT) { const float x = expf(e[j] - mx); e[j] = x; sum += x; }
    sum = block_reduce(sum, red, false);
    const float inv = 1.0f / sum;
    for (int j = tid; j < Tin; j += NT) {
        const float w = e[j] * inv;
        an[j] = w;
        st.a_out[(long)b * st.lda_out + j] = w;
    }
    __syncthreads();
    // ---- context
    {
        const int h = tid / nd, dd = (tid % nd) * 4;
        if (h < nh) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* mp = st.memory + (long)b * Tin * E + dd;
#pragma unroll 8
            for (int j = h; j < Tin; j += nh) acc += an[j] * *reinterpret_cast<const f32x4*>(mp + (long)j * E);
            *reinterpret_cast<f32x4*>(cred + h * E + dd) = acc;
        }
        __syncthreads();
        for (int c = tid; c < E; c += NT) {
            float s2 = 0.f;
            for (int h2 = 0; h2 < nh; ++h2) s2 += cred[h2 * E + c];
            st.ctx1[(long)b * st.ldctx1 + c] = s2;
            if (st.ctx2) st.ctx2[(long)b * st.ldctx2 + c] = s2;
            if (st.ctx16) st.ctx16[(long)b * st.ldctx16 + c] = (__bf16)s2;
            if (st.ctx16b) st.ctx16b[(long)b * st.ldctx16b + c] = (__bf16)s2;
        }
    }
}

// Backward of one GMM attention step (reverse time), one workgroup per (b, stream).
//   g_j = dctx . memory_j + dalign_j ; dphi_j = w_j (g_j - sum w g)                            softmax
//   E_jk = exp(-(j-mu_k)^2 / (2 s_k^2)) ; t_jk = omega_k / Z_k E_jk
//   domega_k = sum_j dphi_j E_jk / Z_k ; dmu_k = sum_j dphi_j t_jk (j-mu_k)/s_k^2 (+ carry: mu_{t+1} = mu_t + delta_{t+1})
//   dsigma_k = sum_j dphi_j t_jk ((j-mu_k)^2 / s_k^3 - 1/s_k) ; ddelta_k = dmu_k (total)
//   through softplus / softmax to the 3K pre-activations, then the second MLP layer (per-item accumulators of dW2, db2)
//   and tanh: dq_a = (W2^T dip)_a (1 - hid_a^2)  -> the existing dq path (d h via W1, dW1 by the big GEMM, db1 = colsum)
__global__ __launch_bounds__(NTB) void attention_gmm_step_bwd_kernel(AttnBwdDesc d) {
    const AttnBwdStream& st = d.st[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Tin = st.Tin, A = d.A, E = d.E;
    const int Tp = (Tin + 3) & ~3;
    constexpr int NWV = NTB / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dctx = smem;              // [E]
    float* hid = dctx + E;           // [A]
    float* g = hid + A;              // [Tp]
    float* wS = g + Tp;              // [Tp]
    float* ip = wS + Tp;             // [16]
    float* dip = ip + 16;            // [16]
    float* red2 = dip + 16;          // [NWV + 8]
    float* psum = red2 + NWV + 8;    // [NWV][16]

    for (int c = tid; c < E; c += NTB) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) if (st.dctx[i]) v += st.dctx[i][(long)b * st.lddctx[i] + c];
        if (st.part && !d.first) {
            const float* p = st.part + (long)b * st.ldpart + st.part_col + c;
            float pv[8];
#pragma unroll
            for (int z = 0; z < 8; ++z) pv[z] = z < st.nparts ? p[(long)z * st.part_stride] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int z = 0; z < 8; ++z) acc += pv[z];
            v += acc;
        }
        dctx[c] = v;
        st.dctx_out[(long)b * st.lddctx_out + c] = v;
    }
    for (int a = tid; a < A; a += NTB) hid[a] = tanhf(st.q[(long)b * st.ldq + a]);
    for (int j = tid; j < Tin; j += NTB) wS[j] = st.w[(long)b * st.ldw + j];
    __syncthreads();
    for (int i = wave; i < 3 * kGmmK; i += NWV) {           // recompute the 3K pre-activations
        float sum = 0.f;
        for (int a = lane; a < A; a += 64) sum += st.gmm_w2[(long)i * A + a] * hid[a];
        sum = wave_sum(sum);
        if (lane == 0) ip[i] = sum + st.gmm_b2[i];
    }
    // g_j: one wave per position, 4 in flight
    {
        constexpr int U = 4;
        for (int j0 = wave; j0 < Tin; j0 += NWV * U) {
            float sum[U] = {0.f, 0.f, 0.f, 0.f};
            for (int c = lane * 4; c < E; c += 256) {
                const f32x4 dc = *reinterpret_cast<const f32x4*>(dctx + c);
                f32x4 mv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) mv[u] = *reinterpret_cast<const f32x4*>(st.memory + ((long)b * Tin + min(j0 + u * NWV, Tin - 1)) * E + c);
#pragma unroll
                for (int u = 0; u < U; ++u) sum[u] += mv[u][0] * dc[0] + mv[u][1] * dc[1] + mv[u][2] * dc[2] + mv[u][3] * dc[3];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * NWV;
                const float tot = wave_sum(sum[u]);
                if (lane == 0 && j < Tin) g[j] = tot + (st.dalign ? st.dalign[(long)b * st.lddalign + j] : 0.f);
            }
        }
    }
    __syncthreads();
    // softmax backward: sdot = sum_j w_j g_j
    float part = 0.f;
    for (int j = tid; j < Tin; j += NTB) part += wS[j] * g[j];
    part = wave_sum(part);
    if (lane == 0) red2[wave] = part;
    __syncthreads();
    float sdot = 0.f;
#pragma unroll
    for (int i = 0; i < NWV; ++i) sdot += red2[i];
    const GmmPar p = gmm_params(ip);
    float mu[kGmmK];
#pragma unroll
    for (int k = 0; k < kGmmK; ++k) mu[k] = st.mu[(long)b * st.ldmu + k];
    float s1[kGmmK], s2[kGmmK], s3[kGmmK];
#pragma unroll
    for (int k = 0; k < kGmmK; ++k) { s1[k] = 0.f; s2[k] = 0.f; s3[k] = 0.f; }
    for (int j = tid; j < Tin; j += NTB) {
        const float dphi = wS[j] * (g[j] - sdot);
#pragma unroll
        for (int k = 0; k < kGmmK; ++k) {
            const float dj = (float)j - mu[k], sg = p.sigma[k], sg2 = sg * sg;
            const float Ejk = expf(-(dj * dj) / sg2 / 2.0f);
            const float t = dphi * p.omega[k] / p.Z[k] * Ejk;
            s1[k] += dphi * Ejk;
            s2[k] += t * dj / sg2;
            s3[k] += t * (dj * dj / (sg2 * sg) - 1.0f / sg);
        }
    }
#pragma unroll
    for (int k = 0; k < kGmmK; ++k) {
        const float a1 = wave_sum(s1[k]), a2 = wave_sum(s2[k]), a3 = wave_sum(s3[k]);
        if (lane == 0) { psum[wave * 16 + k] = a1; psum[wave * 16 + 5 + k] = a2; psum[wave * 16 + 10 + k] = a3; }
    }
    __syncthreads();
    if (tid == 0) {
        float S[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) { float v = 0.f; for (int w2 = 0; w2 < NWV; ++w2) v += psum[w2 * 16 + i]; S[i] = v; }
        float dom[kGmmK], wsum = 0.f;
#pragma unroll
        for (int k = 0; k < kGmmK; ++k) { dom[k] = S[k] / p.Z[k]; wsum += p.omega[k] * dom[k]; }
#pragma unroll
        for (int k = 0; k < kGmmK; ++k) {
            float dmu = S[5 + k];
            float* cp = st.mu_carry + (long)b * kGmmPad + k;
            if (!d.first) dmu += *cp;
            *cp = dmu;                                                   // gradient on mu_{t-1}
            const float sig_d = 1.0f / (1.0f + expf(-ip[kGmmK + k])), sig_s = 1.0f / (1.0f + expf(-ip[2 * kGmmK + k]));
            dip[k] = p.omega[k] * (dom[k] - wsum);                       // softmax
            dip[kGmmK + k] = dmu * (ip[kGmmK + k] > 20.f ? 1.0f : sig_d);            // softplus'
            dip[2 * kGmmK + k] = S[10 + k] * (ip[2 * kGmmK + k] > 20.f ? 1.0f : sig_s);
        }
        float* db = st.db2_acc + (long)b * 16;
#pragma unroll
        for (int i = 0; i < 15; ++i) db[i] = (d.first ? 0.f : db[i]) + dip[i];
    }
    __syncthreads();
    for (int i = tid; i < 3 * kGmmK * A; i += NTB) {                     // dW2[i][a] += dip_i hid_a
        float* pw = st.dw2_acc + (long)b * 3 * kGmmK * A + i;
        *pw = (d.first ? 0.f : *pw) + dip[i / A] * hid[i % A];
    }
    for (int a = tid; a < A; a += NTB) {
        float dh = 0.f;
#pragma unroll
        for (int i = 0; i < 3 * kGmmK; ++i) dh += st.gmm_w2[(long)i * A + a] * dip[i];
        st.dq_out[(long)b * st.lddq_out + a] = dh * (1.0f - hid[a] * hid[a]);
    }
}


// ---------------------------------------------------------------------------------------------
// DynamicConvolutionAttention (attention.py:195-289).  Per step, with a = previous alignment (one-hot at 0 first):
//   hq = tanh(W h + bW) ; G = V hq -> 8 dynamic filters of 21 taps            (W h arrives as the LSTM kernel's partials)
//   f_jc = sum_k F[c][k] a[j+k-10] ; g_jc = sum_k G[c][k] a[j+k-10]
//   prior_j = sum_m P[m] a[j+m-10] ; p_j = log(max(prior_j, 1e-6))
//   e_j = v . tanh(U f_j + T g_j + bT) + p_j ; w = softmax(mask(e)) ; ctx = w . memory
// One workgroup per (b, stream).  LDS: padded alignment, filters, the [T_in][16] feature tile, U|T rows at pitch 17.
// ---------------------------------------------------------------------------------------------
constexpr int kDcaCK = kDcaC * kDcaK, kDcaUT = 2 * kDcaC + 1;          // 168 filter taps; pitch of a [U row | T row]

__global__ __launch_bounds__(NT) void attention_dca_step_fwd_kernel(AttnStepDesc d) {
    const AttnStream& st = d.st[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Tin = st.Tin, A = d.A, E = d.E;
    const int Tp = (Tin + 3) & ~3, TwP = (Tin + 2 * kDcaPad + 3) & ~3;
    const int nd = E / 4, nh = NT / nd;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hq = smem;                       // [A]
    float* e = hq + A;                      // [Tp]
    float* an = e + Tp;                     // [Tp]
    float* red = an + Tp;                   // [4*NT]
    float* cred = red + 4 * NT;             // [nh*E]
    float* apad = cred + nh * E;            // [TwP]  apad[i] = a_prev[i - 10]
    float* G = apad + TwP;                  // [168]
    float* Fw = G + kDcaCK;                 // [168]
    float* fg = Fw + kDcaCK;                // [Tin][16]  f (8) | g (8)
    float* UT = fg + Tp * 16;               // [A][17]    U row (8) | T row (8)
    float* bT = UT + A * kDcaUT;            // [A]
    float* vS = bT + A;                     // [A]
    float* Pf = vS + A;                     // [12]

    // ---- W h + bW from the partials, tanh
    {
        const int a4n = A / 4, ng = NT / a4n;
        const int pg = tid / a4n, a4 = (tid % a4n) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* p = st.qpart + (long)b * A + a4;
        const long ps = (long)d.B * A;
#pragma unroll 4
        for (int i = pg; i < st.nparts; i += ng) acc += *reinterpret_cast<const f32x4*>(p + (long)i * ps);
        *reinterpret_cast<f32x4*>(red + pg * A + a4) = acc;
        for (int i = tid; i < A * 2 * kDcaC; i += NT) {
            const int a = i / (2 * kDcaC), c = i % (2 * kDcaC);
            UT[a * kDcaUT + c] = c < kDcaC ? st.dca.U[a * kDcaC + c] : st.dca.T[a * kDcaC + c - kDcaC];
        }
        for (int a = tid; a < A; a += NT) { bT[a] = st.dca.bT[a]; vS[a] = st.dca.v[a]; }
        for (int i = tid; i < kDcaCK; i += NT) Fw[i] = st.dca.F[i];
        if (tid < kDcaP) Pf[tid] = st.dca.P[tid];
        for (int i = tid; i < TwP; i += NT) {
            const int j = i - kDcaPad;
            apad[i] = (j >= 0 && j < Tin) ? (st.a_prev ? st.a_prev[(long)b * st.lda_prev + j] : (j == 0 ? 1.f : 0.f)) : 0.f;
        }
        __syncthreads();
        if (tid < A) {
            float sum = 0.f;
            const int used = st.nparts < ng ? st.nparts : ng;
            for (int h2 = 0; h2 < used; ++h2) sum += red[h2 * A + tid];
            sum += st.dca.bW[tid];
            if (st.q_out) st.q_out[(long)b * st.ldq_out + tid] = sum;
            hq[tid] = tanhf(sum);
        }
    }
    __syncthreads();
    for (int i = wave; i < kDcaCK; i += NT / 64) {                       // dynamic filters G = V hq
        float sum = 0.f;
        for (int a = lane; a < A; a += 64) sum += st.dca.V[(long)i * A + a] * hq[a];
        sum = wave_sum(sum);
        if (lane == 0) G[i] = sum;
    }
    __syncthreads();
    for (int i = tid; i < Tin * kDcaC; i += NT) {                        // static and dynamic features
        const int j = i / kDcaC, c = i % kDcaC;
        float f = 0.f, g = 0.f;
        for (int k = 0; k < kDcaK; ++k) { const float x = apad[j + k]; f += Fw[c * kDcaK + k] * x; g += G[c * kDcaK + k] * x; }
        fg[j * 16 + c] = f; fg[j * 16 + kDcaC + c] = g;
    }
    __syncthreads();
    // ---- energies: 16 lanes per position
    {
        const int gid = tid >> 4, sub = tid & 15;
        for (int j = gid; j < Tp; j += NT / 16) {
            float sum = 0.f;
            if (j < Tin) {
                float x[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = fg[j * 16 + c];
                for (int a0 = sub * 4; a0 < A; a0 += 64) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const int a = a0 + c4;
                        float u = bT[a];
#pragma unroll
                        for (int c = 0; c < 16; ++c) u += UT[a * kDcaUT + c] * x[c];
                        sum += vS[a] * tanhf(u);
                    }
                }
            }
            sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 1, 64);
            if (sub == 0 && j < Tin) {
                float pr = 0.f;
#pragma unroll
                for (int m = 0; m < kDcaP; ++m) pr += Pf[m] * apad[j + m];
                e[j] = sum + logf(fmaxf(pr, 1e-6f));
            }
        }
    }
    __syncthreads();
    const int len = st.lengths ? st.lengths[b] : Tin;
    float mx = -INFINITY;
    for (int j = tid; j < Tin; j += NT) {
        float ev = e[j];
        if (j >= len) ev = st.mask_value;
        e[j] = ev;
        mx = fmaxf(mx, ev);
    }
    mx = block_reduce(mx, red, true);
    float ssum = 0.f;
    for (int j = tid; j < Tin; j += NT) { const float x = expf(e[j] - mx); e[j] = x; ssum += x; }
    ssum = block_reduce(ssum, red, false);
    const float inv = 1.0f / ssum;
    for (int j = tid; j < Tin; j += NT) {
        const float w = e[j] * inv;
        an[j] = w;
        st.a_out[(long)b * st.lda_out + j] = w;
    }
    __syncthreads();
    {
        const int h = tid / nd, dd = (tid % nd) * 4;
        if (h < nh) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* mp = st.memory + (long)b * Tin * E + dd;
#pragma unroll 8
            for (int j = h; j < Tin; j += nh) acc += an[j] * *reinterpret_cast<const f32x4*>(mp + (long)j * E);
            *reinterpret_cast<f32x4*>(cred + h * E + dd) = acc;
        }
        __syncthreads();
        for (int c = tid; c < E; c += NT) {
            float s2 = 0.f;
            for (int h2 = 0; h2 < nh; ++h2) s2 += cred[h2 * E + c];
            st.ctx1[(long)b * st.ldctx1 + c] = s2;
            if (st.ctx2) st.ctx2[(long)b * st.ldctx2 + c] = s2;
            if (st.ctx16) st.ctx16[(long)b * st.ldctx16 + c] = (__bf16)s2;
            if (st.ctx16b) st.ctx16b[(long)b * st.ldctx16b + c] = (__bf16)s2;
        }
    }
}

// Backward of one DCA step (reverse time), one workgroup per (b, stream).  Per-item accumulators (dca_acc):
// dv [A] | dbT [A] | dU [A][8] | dT [A][8] | dF [168] | dV [168][A]; dW / dbW go through the dq path.
__global__ __launch_bounds__(NTB) void attention_dca_step_bwd_kernel(AttnBwdDesc d) {
    const AttnBwdStream& st = d.st[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Tin = st.Tin, A = d.A, E = d.E;
    const int Tp = (Tin + 3) & ~3, TwP = (Tin + 2 * kDcaPad + 3) & ~3, AS = A + 4;
    constexpr int NWV = NTB / 64, NPG2 = NTB / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dctx = smem;                     // [E]
    float* hq = dctx + E;                   // [A]
    float* g = hq + A;                      // [Tp]
    float* wS = g + Tp;                     // [Tp]
    float* de = wS + Tp;                    // [Tp]
    float* apad = de + Tp;                  // [TwP]
    float* dprP = apad + TwP;               // [TwP]      dprP[i] = d(prior)_{i-10}, zero padded
    float* G = dprP + TwP;                  // [168]
    float* Fw = G + kDcaCK;                 // [168]
    float* dGs = Fw + kDcaCK;               // [168]
    float* fg = dGs + kDcaCK;               // [Tp][16]
    float* dfgP = fg + Tp * 16;             // [TwP][16]  d(f|g)_{i-10}, zero padded rows
    float* UT = dfgP + TwP * 16;            // [A][17]
    float* bT = UT + A * kDcaUT;            // [A]
    float* vS = bT + A;                     // [A]
    float* Pf = vS + A;                     // [12]
    float* red2 = Pf + 12;                  // [16]
    float* S = red2 + 16;                   // [NPG2][AS]  s tile of a chunk; later the dv / dbT group partials and dwcat partials
    float* acc0 = st.dca_acc + (long)b * dca_acc_floats(A);
    float* acc_dv = acc0; float* acc_dbT = acc0 + A; float* acc_dU = acc0 + 2 * A; float* acc_dT = acc_dU + A * kDcaC;
    float* acc_dF = acc_dT + A * kDcaC; float* acc_dV = acc_dF + kDcaCK;

    for (int c = tid; c < E; c += NTB) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) if (st.dctx[i]) v += st.dctx[i][(long)b * st.lddctx[i] + c];
        if (st.part && !d.first) {
            const float* p = st.part + (long)b * st.ldpart + st.part_col + c;
            float pv[8];
#pragma unroll
            for (int z = 0; z < 8; ++z) pv[z] = z < st.nparts ? p[(long)z * st.part_stride] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int z = 0; z < 8; ++z) acc += pv[z];
            v += acc;
        }
        dctx[c] = v;
        st.dctx_out[(long)b * st.lddctx_out + c] = v;
    }
    for (int a = tid; a < A; a += NTB) { hq[a] = tanhf(st.q[(long)b * st.ldq + a]); bT[a] = st.dca.bT[a]; vS[a] = st.dca.v[a]; }
    for (int j = tid; j < Tin; j += NTB) wS[j] = st.w[(long)b * st.ldw + j];
    for (int i = tid; i < A * 2 * kDcaC; i += NTB) {
        const int a = i / (2 * kDcaC), c = i % (2 * kDcaC);
        UT[a * kDcaUT + c] = c < kDcaC ? st.dca.U[a * kDcaC + c] : st.dca.T[a * kDcaC + c - kDcaC];
    }
    for (int i = tid; i < kDcaCK; i += NTB) Fw[i] = st.dca.F[i];
    if (tid < kDcaP) Pf[tid] = st.dca.P[tid];
    for (int i = tid; i < TwP; i += NTB) {
        const int j = i - kDcaPad;
        apad[i] = (j >= 0 && j < Tin) ? (st.a_prev ? st.a_prev[(long)b * st.lda_prev + j] : (j == 0 ? 1.f : 0.f)) : 0.f;
        dprP[i] = 0.f;
    }
    for (int i = tid; i < TwP * 16; i += NTB) dfgP[i] = 0.f;
    __syncthreads();
    for (int i = wave; i < kDcaCK; i += NWV) {                           // dynamic filters recomputed
        float sum = 0.f;
        for (int a = lane; a < A; a += 64) sum += st.dca.V[(long)i * A + a] * hq[a];
        sum = wave_sum(sum);
        if (lane == 0) G[i] = sum;
    }
    {   // g_j = dctx . memory_j + dalign_j + carry_j
        constexpr int U = 4;
        for (int j0 = wave; j0 < Tin; j0 += NWV * U) {
            float sum[U] = {0.f, 0.f, 0.f, 0.f};
            for (int c = lane * 4; c < E; c += 256) {
                const f32x4 dc = *reinterpret_cast<const f32x4*>(dctx + c);
                f32x4 mv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) mv[u] = *reinterpret_cast<const f32x4*>(st.memory + ((long)b * Tin + min(j0 + u * NWV, Tin - 1)) * E + c);
#pragma unroll
                for (int u = 0; u < U; ++u) sum[u] += mv[u][0] * dc[0] + mv[u][1] * dc[1] + mv[u][2] * dc[2] + mv[u][3] * dc[3];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * NWV;
                const float tot = wave_sum(sum[u]);
                if (lane == 0 && j < Tin) {
                    float gs = tot;
                    if (st.dalign) gs += st.dalign[(long)b * st.lddalign + j];
                    if (!d.first) gs += st.carry[(long)b * Tin + j];
                    g[j] = gs;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < Tin * kDcaC; i += NTB) {                       // features recomputed
        const int j = i / kDcaC, c = i % kDcaC;
        float f = 0.f, gg = 0.f;
        for (int k = 0; k < kDcaK; ++k) { const float x = apad[j + k]; f += Fw[c * kDcaK + k] * x; gg += G[c * kDcaK + k] * x; }
        fg[j * 16 + c] = f; fg[j * 16 + kDcaC + c] = gg;
    }
    {   // softmax backward, prior backward
        float part = 0.f;
        for (int j = tid; j < Tin; j += NTB) part += wS[j] * g[j];
        part = wave_sum(part);
        if (lane == 0) red2[wave] = part;
        __syncthreads();
        float sdot = 0.f;
#pragma unroll
        for (int i = 0; i < NWV; ++i) sdot += red2[i];
        for (int j = tid; j < Tin; j += NTB) {
            const float dej = wS[j] * (g[j] - sdot);
            de[j] = dej;
            float pr = 0.f;
#pragma unroll
            for (int m = 0; m < kDcaP; ++m) pr += Pf[m] * apad[j + m];
            dprP[j + kDcaPad] = pr >= 1e-6f ? dej / pr : 0.f;             // log(clamp_min(prior, 1e-6))
        }
    }
    __syncthreads();

    // ---- energies backward in chunks of NPG2 positions: s_ja = de_j v_a (1 - tanh(u_ja)^2) through an LDS tile
    const int gid = tid >> 4, sub = tid & 15;
    float dv[4][4], dbt[4][4], dUT[2 * kDcaC * 256 / NTB];               // A <= 256: A*16 outputs over NTB threads
    constexpr int NUT = 2 * kDcaC * 256 / NTB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { dv[i][c] = 0.f; dbt[i][c] = 0.f; }
#pragma unroll
    for (int k = 0; k < NUT; ++k) dUT[k] = 0.f;
    for (int j0 = 0; j0 < Tin; j0 += NPG2) {
        const int j = j0 + gid;
        const bool valid = j < Tin;
        const int jc = valid ? j : Tin - 1;
        const float dej = valid ? de[jc] : 0.f;
        float x[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = fg[jc * 16 + c];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a0 = sub * 4 + 64 * i;
            if (a0 < A) {
                f32x4 sv;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int a = a0 + c4;
                    float u = bT[a];
#pragma unroll
                    for (int c = 0; c < 16; ++c) u += UT[a * kDcaUT + c] * x[c];
                    const float th = tanhf(u);
                    const float sa = dej * vS[a] * (1.0f - th * th);
                    dv[i][c4] += dej * th;
                    dbt[i][c4] += sa;
                    sv[c4] = sa;
                }
                *reinterpret_cast<f32x4*>(S + gid * AS + a0) = sv;
            }
        }
        __syncthreads();
        const int nj = min(NPG2, Tin - j0);
#pragma unroll
        for (int k = 0; k < NUT; ++k) {                                  // dU | dT: thread-owned outputs (a, c)
            const int o = tid + k * NTB;
            if (o < A * 16) {
                const int a = o >> 4, c = o & 15;
                float sum = dUT[k];
                for (int jj = 0; jj < nj; ++jj) sum += S[jj * AS + a] * fg[(j0 + jj) * 16 + c];
                dUT[k] = sum;
            }
        }
        {   // d(f|g) of this chunk: one output (jj, c) per thread
            const int jj = tid >> 4, c = tid & 15;
            if (jj < nj) {
                float sum = 0.f;
                for (int a = 0; a < A; ++a) sum += UT[a * kDcaUT + c] * S[jj * AS + a];
                dfgP[(j0 + jj + kDcaPad) * 16 + c] = sum;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NUT; ++k) {
        const int o = tid + k * NTB;
        if (o < A * 16) {
            const int a = o >> 4, c = o & 15;
            float* p = c < kDcaC ? acc_dU + a * kDcaC + c : acc_dT + a * kDcaC + c - kDcaC;
            *p = (d.first ? 0.f : *p) + dUT[k];
        }
    }
    // dv, dbT: reduce the position groups through the (now free) tile
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a0 = sub * 4 + 64 * i;
            if (a0 < A) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) S[gid * AS + a0 + c4] = pass == 0 ? dv[i][c4] : dbt[i][c4];
            }
        }
        __syncthreads();
        for (int a = tid; a < A; a += NTB) {
            float sum = 0.f;
            for (int k = 0; k < NPG2; ++k) sum += S[k * AS + a];
            float* p = (pass == 0 ? acc_dv : acc_dbT) + a;
            *p = (d.first ? 0.f : *p) + sum;
        }
        __syncthreads();
    }
    // dF (accumulated) and dG (this step's dynamic-filter gradient)
    for (int it = tid; it < 2 * kDcaCK; it += NTB) {
        const int which = it / kDcaCK, ck = it % kDcaCK, c = ck / kDcaK, k = ck % kDcaK;
        float sum = 0.f;
        for (int j = 0; j < Tin; ++j) sum += dfgP[(j + kDcaPad) * 16 + which * kDcaC + c] * apad[j + k];
        if (which == 0) { float* p = acc_dF + ck; *p = (d.first ? 0.f : *p) + sum; }
        else dGs[ck] = sum;
    }
    // gradient on the previous alignment: prior conv + static + dynamic convs, 8 partials per position
    float* tmp = S;                                                       // [8][Tp]
    if (st.a_prev) {
        for (int it = tid; it < kDcaC * Tin; it += NTB) {
            const int c = it / Tin, i = it % Tin;
            float sum = 0.f;
            if (c == 0) {
#pragma unroll
                for (int m = 0; m < kDcaP; ++m) sum += Pf[m] * dprP[i - m + 2 * kDcaPad];
            }
            for (int k = 0; k < kDcaK; ++k) {
                const float* row = dfgP + (i - k + 2 * kDcaPad) * 16;
                sum += Fw[c * kDcaK + k] * row[c] + G[c * kDcaK + k] * row[kDcaC + c];
            }
            tmp[c * Tp + i] = sum;
        }
    }
    __syncthreads();
    if (st.a_prev) {
        for (int i = tid; i < Tin; i += NTB) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < kDcaC; ++c) sum += tmp[c * Tp + i];
            st.carry[(long)b * Tin + i] = sum;
        }
    }
    // through V and tanh to dq; dV accumulated
    for (int a = tid; a < A; a += NTB) {
        float dh = 0.f;
        for (int i = 0; i < kDcaCK; ++i) dh += st.dca.V[(long)i * A + a] * dGs[i];
        st.dq_out[(long)b * st.lddq_out + a] = dh * (1.0f - hq[a] * hq[a]);
    }
    for (int i = tid; i < kDcaCK * A; i += NTB) {
        float* p = acc_dV + i;
        *p = (d.first ? 0.f : *p) + dGs[i / A] * hq[i % A];
    }
}

}  // namespace

size_t attention_fwd_smem(const AttnStepDesc& d, bool with_pa) {
    int Tmax = 0;
    for (int i = 0; i < d.nstreams; ++i) Tmax = d.st[i].Tin > Tmax ? d.st[i].Tin : Tmax;
    const int Tp = (Tmax + 3) & ~3;
    const int nh = NT / (d.E / 4);
    size_t n = 2 * d.A + 3 * Tp + 4 * NT + (size_t)nh * d.E;
    if (d.kind == 1) {
        const int TwP = (Tmax + d.Kc - 1 + 4 + 3) & ~3;
        n += (size_t)d.F * 2 * d.Kc + (size_t)d.A * (d.F + 1) + (((size_t)Tmax * (d.F + 1) + 3) & ~(size_t)3) + 2 * (size_t)TwP;
        if (with_pa) n += (size_t)Tmax * (d.A + 8);
    }
    return n * sizeof(float);
}

int attention_step_fwd(const AttnStepDesc& din, hipStream_t s) {
    AttnStepDesc d = din;
    if (d.kind == 3) {                                               // DCA
        T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= 2 && d.A % 4 == 0 && d.A <= 256 && NT % (d.A / 4) == 0 && d.E % 4 == 0 && NT % (d.E / 4) == 0,
                   "attention_step (DCA): A=%d E=%d unsupported", d.A, d.E);
        int Tmax = 0;
        for (int i = 0; i < d.nstreams; ++i) {
            Tmax = d.st[i].Tin > Tmax ? d.st[i].Tin : Tmax;
            const DcaWeights& w = d.st[i].dca;
            T2_REQUIRE(d.st[i].qpart && w.bW && w.V && w.F && w.U && w.T && w.bT && w.v && w.P, "attention_step (DCA): missing weights");
        }
        const int Tp = (Tmax + 3) & ~3, TwP = (Tmax + 2 * kDcaPad + 3) & ~3;
        const size_t smem = ((size_t)d.A + 2 * Tp + 4 * NT + (size_t)(NT / (d.E / 4)) * d.E + TwP + 2 * kDcaCK + (size_t)Tp * 16 +
                             (size_t)d.A * kDcaUT + 2 * d.A + 12) * sizeof(float);
        T2_REQUIRE(smem <= 160 * 1024, "attention_step (DCA): T_in too long for LDS (%zu bytes)", smem);
        if (smem > 64 * 1024)
            T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_dca_step_fwd_kernel), smem));
        hipLaunchKernelGGL(attention_dca_step_fwd_kernel, dim3(d.B, d.nstreams), dim3(NT), smem, s, d);
        T2_LAUNCH_CHECK();
        return 0;
    }
    if (d.kind == 2) {                                               // GMM
        T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= 2 && d.A % 4 == 0 && d.A <= 256 && NT % (d.A / 4) == 0 && d.E % 4 == 0 && NT % (d.E / 4) == 0,
                   "attention_step (GMM): A=%d E=%d unsupported", d.A, d.E);
        int Tmax = 0;
        for (int i = 0; i < d.nstreams; ++i) {
            Tmax = d.st[i].Tin > Tmax ? d.st[i].Tin : Tmax;
            T2_REQUIRE(d.st[i].qpart && d.st[i].gmm_b1 && d.st[i].gmm_w2 && d.st[i].gmm_b2 && d.st[i].mu_out, "attention_step (GMM): missing buffers");
        }
        const int Tp = (Tmax + 3) & ~3;
        const size_t smem = ((size_t)d.A + 2 * Tp + 4 * NT + (size_t)(NT / (d.E / 4)) * d.E + 16 + 32) * sizeof(float);
        T2_REQUIRE(smem <= 160 * 1024, "attention_step (GMM): T_in too long for LDS (%zu bytes)", smem);
        if (smem > 64 * 1024)
            T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_gmm_step_fwd_kernel), smem));
        hipLaunchKernelGGL(attention_gmm_step_fwd_kernel, dim3(d.B, d.nstreams), dim3(NT), smem, s, d);
        T2_LAUNCH_CHECK();
        return 0;
    }
    T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= 2, "attention_step: nstreams=%d", d.nstreams);
    T2_REQUIRE(d.A % 4 == 0 && d.A <= 256 && NT % (d.A / 4) == 0, "attention_step: attention_dim %d must be a multiple of 4 dividing %d, <= 256", d.A, 4 * NT);
    T2_REQUIRE(d.E % 4 == 0 && d.E / 4 <= NT && NT % (d.E / 4) == 0, "attention_step: encoder dim %d unsupported", d.E);
    T2_REQUIRE(d.kind == 0 || (d.Kc % 2 == 1 && d.F >= 1), "attention_step: bad location layer F=%d Kc=%d", d.F, d.Kc);
    for (int i = 0; i < d.nstreams; ++i)
        T2_REQUIRE(((uintptr_t)d.st[i].pm & 15) == 0 && ((uintptr_t)d.st[i].memory & 15) == 0, "attention_step: pm/memory must be 16-byte aligned");
    // LSA: the dense location projection runs on the matrix cores when its [T_in][A] tile fits in LDS beside the rest
    d.lsa_pa = d.kind == 1 && d.A % 32 == 0 && attention_fwd_smem(d, true) <= 160 * 1024;
    const size_t smem = attention_fwd_smem(d, d.lsa_pa);
    T2_REQUIRE(smem <= 160 * 1024, "attention_step: T_in too long for LDS (%zu bytes)", smem);
    if (smem > 64 * 1024) {
        T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_step_fwd_kernel), smem));
    }
    hipLaunchKernelGGL(attention_step_fwd_kernel, dim3(d.B, d.nstreams), dim3(NT), smem, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}


template <int MAXI>
static int launch_lsa_bwd(const AttnBwdDesc& d, size_t smem, hipStream_t s) {
    if (smem > 64 * 1024)
        T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_lsa_step_bwd_kernel<MAXI>), smem));
    hipLaunchKernelGGL(attention_lsa_step_bwd_kernel<MAXI>, dim3(d.B, d.nstreams), dim3(NTB), smem, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

int attention_step_bwd(const AttnBwdDesc& d, hipStream_t s) {
    T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= 2, "attention_bwd: nstreams=%d", d.nstreams);
    T2_REQUIRE(d.A % 4 == 0 && d.A <= 256 && d.E % 4 == 0, "attention_bwd: A=%d E=%d unsupported", d.A, d.E);
    int Tmax = 0;
    for (int i = 0; i < d.nstreams; ++i) Tmax = d.st[i].Tin > Tmax ? d.st[i].Tin : Tmax;
    if (d.kind == 3) {                                               // DCA
        for (int i = 0; i < d.nstreams; ++i) {
            const DcaWeights& w = d.st[i].dca;
            T2_REQUIRE(d.st[i].w && d.st[i].dca_acc && w.V && w.F && w.U && w.T && w.bT && w.v && w.P, "attention_bwd (DCA): missing buffers");
        }
        const int Tp = (Tmax + 3) & ~3, TwP = (Tmax + 2 * kDcaPad + 3) & ~3;
        const size_t tile = std::max((size_t)(NTB / 16) * (d.A + 4), (size_t)kDcaC * Tp);
        const size_t smem = ((size_t)d.E + d.A + 3 * Tp + 2 * TwP + 3 * kDcaCK + (size_t)Tp * 16 + (size_t)TwP * 16 + (size_t)d.A * kDcaUT +
                             2 * d.A + 12 + 16 + tile) * sizeof(float);
        T2_REQUIRE(smem <= 160 * 1024, "attention_bwd (DCA): T_in too long for LDS (%zu bytes)", smem);
        if (smem > 64 * 1024)
            T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_dca_step_bwd_kernel), smem));
        hipLaunchKernelGGL(attention_dca_step_bwd_kernel, dim3(d.B, d.nstreams), dim3(NTB), smem, s, d);
        T2_LAUNCH_CHECK();
        return 0;
    }
    if (d.kind == 2) {                                               // GMM
        for (int i = 0; i < d.nstreams; ++i)
            T2_REQUIRE(d.st[i].w && d.st[i].gmm_w2 && d.st[i].gmm_b2 && d.st[i].mu && d.st[i].mu_carry && d.st[i].dw2_acc && d.st[i].db2_acc,
                       "attention_bwd (GMM): missing buffers");
        const int Tp = (Tmax + 3) & ~3;
        const size_t smem = ((size_t)d.E + d.A + 2 * Tp + 32 + NTB / 64 + 8 + (NTB / 64) * 16) * sizeof(float);
        T2_REQUIRE(smem <= 160 * 1024, "attention_bwd (GMM): T_in too long for LDS (%zu bytes)", smem);
        if (smem > 64 * 1024)
            T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_gmm_step_bwd_kernel), smem));
        hipLaunchKernelGGL(attention_gmm_step_bwd_kernel, dim3(d.B, d.nstreams), dim3(NTB), smem, s, d);
        T2_LAUNCH_CHECK();
        return 0;
    }
    if (d.kind == 1) {
        T2_REQUIRE(d.Kc % 2 == 1 && d.F >= 1 && d.F <= 32, "attention_bwd (LSA): location layer F=%d (<= 32) Kc=%d (odd) unsupported", d.F, d.Kc);
        for (int i = 0; i < d.nstreams; ++i)
            T2_REQUIRE(d.st[i].w && d.st[i].carry_cum && d.st[i].dconv_acc && d.st[i].ddense_acc && d.st[i].loc_conv && d.st[i].loc_dense,
                       "attention_bwd (LSA): missing buffers");
        const size_t smem_m = (size_t)lsa_mfma_smem(Tmax, d.A, d.E, d.F, d.Kc).total * sizeof(float);
        if (d.A % 32 == 0 && smem_m <= 160 * 1024 && (Tmax + 31) / 32 + d.A / 32 < NTL / 64) {     // matrix-core variant
            if (d.A <= 128) {
                T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_lsa_step_bwd_mfma_kernel<2>), smem_m));
                hipLaunchKernelGGL(attention_lsa_step_bwd_mfma_kernel<2>, dim3(d.B, d.nstreams), dim3(NTL), smem_m, s, d);
            } else {
                T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_lsa_step_bwd_mfma_kernel<4>), smem_m));
                hipLaunchKernelGGL(attention_lsa_step_bwd_mfma_kernel<4>, dim3(d.B, d.nstreams), dim3(NTL), smem_m, s, d);
            }
            T2_LAUNCH_CHECK();
            return 0;
        }
        const size_t smem = (size_t)lsa_bwd_smem(Tmax, d.A, d.E, d.F, d.Kc).total * sizeof(float);
        T2_REQUIRE(smem <= 160 * 1024, "attention_bwd (LSA): T_in=%d too long for LDS (%zu bytes > 160 KiB)", Tmax, smem);
        return d.A <= 128 ? launch_lsa_bwd<2>(d, smem, s) : launch_lsa_bwd<4>(d, smem, s);
    }
    T2_REQUIRE(d.nsplit >= 1 && d.nsplit <= 4, "attention_bwd: nsplit=%d", d.nsplit);
    for (int i = 0; i < d.nstreams; ++i) T2_REQUIRE(d.st[i].carry_out && d.st[i].carry_out != d.st[i].carry, "attention_bwd: carry_out must be a second buffer");
    const int chunk = (((Tmax + d.nsplit - 1) / d.nsplit) + 3) & ~3;
    const size_t smem = ((size_t)d.E + 2 * d.A + 3 * (size_t)(chunk + 4) + 2 * (NTB / 16) * (size_t)d.A) * sizeof(float);
    T2_REQUIRE(smem <= 160 * 1024, "attention_bwd: T_in too long for LDS (%zu bytes)", smem);
    const dim3 grid(d.B, d.nstreams, d.nsplit);
    if (d.A <= 128) {
        if (smem > 64 * 1024)
            T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_step_bwd_kernel<2>), smem));
        hipLaunchKernelGGL(attention_step_bwd_kernel<2>, grid, dim3(NTB), smem, s, d);
    } else {
        if (smem > 64 * 1024)
            T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(attention_step_bwd_kernel<4>), smem));
        hipLaunchKernelGGL(attention_step_bwd_kernel<4>, grid, dim3(NTB), smem, s, d);
    }
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
