// LSTM time-step kernels (forward step; backward pointwise; backward recurrent-input GEMM) for up
// to 4 independent cells per launch ("streams": the phone and sub-word attention LSTMs of
// Decoder.decode, or the directions of an encoder BiLSTM).
//
//   gates = pre[b,:] (+bias1+bias2) + sum_seg x_seg[b,:] . W_seg[n,:]^T        (skinny GEMM, M = B)
//   i,f,g,o = sigmoid, sigmoid, tanh, sigmoid ; c' = f*c + i*g ; h' = o*tanh(c')
//   h_out = dropout(h'), c_out = dropout(c')                                   (model.py:340-346,371-373)
//
// Work split (forward): one workgroup owns 8 hidden units = 32 gate rows of W for ALL batch rows,
// so every weight element is read exactly once per step chip-wide.  Operands are staged through LDS
// in wide K-stages with full-line global loads (16 B per lane, lanes along K): loading
// MFMA fragments straight from global memory touches 32 B of 32 different lines per instruction
// and thrashes the 32 KB L1.  LDS tiles stay row-major at pitch K+4: ds_write_b128 stores and
// ds_read_b128 fragment reads are both conflict-free.  The 8 waves split each stage's K (v_mfma_f32_32x32x2_f32, exact fp32 fma chains)
// and their partial tiles are summed through LDS in a fixed order.  Optionally the workgroup also
// emits its 8-unit partial of the attention query projection W_q h (attention.py:368) so that no
// separate launch is needed between the LSTM and the attention kernel.
#include <algorithm>
#include <numeric>

#include "kernels.h"

#ifdef T2_STAMPS
__device__ unsigned long long t2_stamps[32];
#define T2_STAMP(i)                                                                              \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0) t2_stamps[(d.nstreams == 2 ? 0 : 8) + i] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
extern "C" int t2_debug_read_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(t2_stamps), sizeof(unsigned long long) * n);
}
#else
#define T2_STAMP(i)
#endif

namespace t2 {

namespace {

constexpr int HU = 8;        // hidden units per workgroup
constexpr int NW = 8;        // waves per workgroup (split-K inside a stage)
constexpr int NTH = NW * 64;
constexpr int PP = 33;       // LDS pitch of a 32-wide partial tile

// One pipeline stage = BKT columns of K for all MT*32 batch rows (A) and the 32 output columns (B).
// BKT is as large as LDS allows: with ONE workgroup per CU the only way to cover the ~1.5 us
// L2-miss latency of the operand stream (activations were just written by another XCD, weights
// come from MALL/HBM) is to have ~100 KB per CU in flight, i.e. to request the whole next stage
// into registers before computing the current one.
template <int MT, int BKT> struct Tile {
    static constexpr int PA = BKT + 4;                         // row-major tiles, 16-byte aligned rows
    static constexpr int A_FLOATS = MT * 32 * PA;
    static constexpr int BF_FLOATS = 32 * PA;                  // forward B tile: [32 n][BKT]  (K-contiguous W rows)
    static constexpr int PBN = 36;                             // backward B tile: [BKT k][32 n] (N-contiguous W rows)
    static constexpr int BB_FLOATS = BKT * PBN;
    static constexpr int QA = MT * 32 * (BKT / 4) / NTH;       // float4 per thread, A tile
    static constexpr int QB = 32 * (BKT / 4) / NTH;            // float4 per thread, B tile (either layout)
    static constexpr int PART_FLOATS = (MT >= 2 ? 2 : 1) * NW * 32 * PP;      // the tail sums two batch tiles at a time
    static constexpr int FWD_FLOATS = (A_FLOATS + BF_FLOATS > PART_FLOATS ? A_FLOATS + BF_FLOATS : PART_FLOATS);
    static constexpr int BWD_FLOATS = (A_FLOATS + BB_FLOATS > NW * 32 * PP ? A_FLOATS + BB_FLOATS : NW * 32 * PP);
};

// Tile of R rows x BKT, K-contiguous source rows -> registers (16 B per lane, lanes along K: every
// 128-byte line is requested once) -> LDS row-major with ds_write_b128.
template <int R, int BKT, int Q, typename RowPtr>
__device__ __forceinline__ void load_rows(RowPtr rowptr, int k0, f32x4 (&regs)[Q]) {
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        // branch-free: rowptr clamps out-of-range rows to a valid one and reports validity; a per-lane
        // branch around each load would stop the compiler from issuing the stage's loads back to back
        const int q = threadIdx.x + i * NTH, row = q / (BKT / 4), k = (q % (BKT / 4)) * 4;
        bool ok;
        const float* p = rowptr(row, ok);
        const f32x4 x = *reinterpret_cast<const f32x4*>(p + k0 + k);
        regs[i] = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
template <int BKT, int Q>
__device__ __forceinline__ void store_rows(float* __restrict__ lds, const f32x4 (&regs)[Q]) {
    constexpr int P = BKT + 4;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const int q = threadIdx.x + i * NTH, row = q / (BKT / 4), k = (q % (BKT / 4)) * 4;
        *reinterpret_cast<f32x4*>(lds + row * P + k) = regs[i];
    }
}

// Forward stage: A [MT*32][PA], B [32][PA], both K-contiguous.  Wave w owns k in [w*BKT/8, (w+1)*BKT/8).
// Lane (r, hk) reads 4 consecutive k (ds_read_b128, conflict-free at pitch BKT+4) and feeds element j
// to MFMA j — A and B use the same k permutation, so each product pairs the same k.
template <int MT, int BKT>
__device__ __forceinline__ void compute_stage_fwd(const float* __restrict__ As, const float* __restrict__ Bs, int wave, int r, int hk,
                                                  f32x16 (&acc)[MT]) {
    constexpr int PA = BKT + 4;
#pragma unroll
    for (int kk = 0; kk < BKT / NW; kk += 8) {
        const int k = wave * (BKT / NW) + kk + 4 * hk;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bs + r * PA + k);
        f32x4 a4[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a4[m] = *reinterpret_cast<const f32x4*>(As + (m * 32 + r) * PA + k);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][j], b4[j], acc[m], 0, 0, 0);
    }
}
// Backward stage: A as above, B [BKT k][PBN] with n contiguous.
template <int MT, int BKT>
__device__ __forceinline__ void compute_stage_bwd(const float* __restrict__ As, const float* __restrict__ Bs, int wave, int r, int hk,
                                                  f32x16 (&acc)[MT]) {
    constexpr int PA = BKT + 4, PBN = Tile<MT, BKT>::PBN;
#pragma unroll
    for (int kk = 0; kk < BKT / NW; kk += 8) {
        const int k = wave * (BKT / NW) + kk + 4 * hk;
        float b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[(k + j) * PBN + r];
        f32x4 a4[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a4[m] = *reinterpret_cast<const f32x4*>(As + (m * 32 + r) * PA + k);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][j], b[j], acc[m], 0, 0, 0);
    }
}

// Pointwise operands that do not depend on the GEMM — pre-activations (+ biases), previous cell state and this
// unit group's slice of the query projection W_q — are requested right after the first operand stage so that their
// (cold, HBM) latency hides under the K loop.  Thread (mi, bl, uu) serves batch row (gi*G + mi)*32 + bl, unit u0 + uu
// of group gi (G = 2 batch tiles are finished per pass by the 512 threads).
template <int MT> struct TailRegs {
    static constexpr int G = MT >= 2 ? 2 : 1, NG = (MT + G - 1) / G;
    float pre[NG][4]; float cp[NG]; float wq[4];
};
template <int MT>
__device__ __forceinline__ void tail_prefetch(const LstmStepDesc& d, const LstmStream& st, int u0, TailRegs<MT>& tr) {
    constexpr int G = TailRegs<MT>::G, NG = TailRegs<MT>::NG;
    const int B = d.B, H = d.H;
    const int mi = threadIdx.x >> 8, bl = (threadIdx.x & 255) >> 3, u = u0 + (threadIdx.x & 7);
    if (mi < G) {
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int b = min((gi * G + mi) * 32 + bl, B - 1);            // clamped: rows >= B are never consumed
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = 0.f;
                if (st.pre) v += st.pre[(long)b * st.ldpre + g * H + u];      // wave-uniform conditions
                if (st.bias1) v += st.bias1[g * H + u];
                if (st.bias2) v += st.bias2[g * H + u];
                tr.pre[gi][g] = v;
            }
            tr.cp[gi] = st.c_prev ? st.c_prev[(long)b * st.ldc_prev + u] : 0.f;
        }
    }
    if (st.wq) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = min((int)threadIdx.x + k * NTH, st.A * HU - 1);
            tr.wq[k] = st.wq[(long)(i / HU) * H + u0 + (i % HU)];
        }
    }
}

// Everything after the K loop, shared by the fp32- and bf16-operand step kernels: sum the NW partial
// tiles in fixed order, gates, cell update, dropout, stores, optional query-projection partials.
template <int MT>
__device__ __forceinline__ void lstm_tail(const LstmStepDesc& d, const LstmStream& st, int u0, int wave, int r, int hk,
                                          f32x16 (&acc)[MT], float* __restrict__ part, float* __restrict__ hs, const TailRegs<MT>& tr) {
    constexpr int G = TailRegs<MT>::G, NG = TailRegs<MT>::NG;
    const int B = d.B, H = d.H;
    const RngKey kh = rng_key(d.seed, st.site_h), kc = rng_key(d.seed, st.site_c);
    const float scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    const int mi = threadIdx.x >> 8, bl = (threadIdx.x & 255) >> 3, uu = threadIdx.x & 7;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        if (gi > 0) __syncthreads();
        // lane holds column r, rows (e&3) + 8*(e>>2) + 4*hk
#pragma unroll
        for (int q = 0; q < G; ++q) {
            if (gi * G + q < MT) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    part[((q * NW + wave) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PP + r] = acc[gi * G + q < MT ? gi * G + q : 0][e];
            }
        }
        __syncthreads();
        const int m = gi * G + mi;
        const int b = m * 32 + bl, u = u0 + uu;
        if (mi < G && m < MT && b < B) {
            const float* pt = part + mi * NW * 32 * PP;
            float g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += pt[(w * 32 + bl) * PP + g * 8 + uu];
                g4[g] = sum + tr.pre[gi][g];
            }
            const bool active = !st.lengths || st.t < st.lengths[b];
            float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf(g4[2]), og = sigmoidf_(g4[3]);
            float cn = fg * tr.cp[gi] + ig * gg;
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
            if (st.h16_out) st.h16_out[(long)b * st.ldh16 + u] = (__bf16)ho;
            if (st.h16_out2) st.h16_out2[(long)b * st.ldh16_2 + u] = (__bf16)ho;
            hs[b * HU + uu] = ho;
        }
    }

    T2_STAMP(3);
    if (st.wq) {
        // partial query projection of this unit group: qpart[group][b][a] = sum_uu h[b,u0+uu] * Wq[a,u0+uu]
        __syncthreads();
        float* wqs = part;                            // [A][HU]
        const int A = st.A;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + k * NTH;
            if (i < A * HU) wqs[i] = tr.wq[k];
        }
        __syncthreads();
        float* qp = st.qpart + (long)blockIdx.x * B * A;
        for (int i = threadIdx.x; i < B * A; i += NTH) {
            const int b = i / A, a = i % A;
            float sum = 0.f;
#pragma unroll
            for (int uu2 = 0; uu2 < HU; ++uu2) sum += hs[b * HU + uu2] * wqs[a * HU + uu2];
            qp[i] = sum;
        }
    }
}

template <int MT, int BKT>
__global__ __launch_bounds__(NTH) void lstm_step_fwd_kernel(LstmStepDesc d) {
    using TL = Tile<MT, BKT>;
    const LstmStream& st = d.st[blockIdx.y];
    const int B = d.B, H = d.H;
    const int u0 = blockIdx.x * HU;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hk = lane >> 5;

    T2_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + TL::A_FLOATS;
    float* part = smem;                              // [NW][32][PP], aliases the staging area after the K loop
    float* hs = smem + TL::FWD_FLOATS;               // [MT*32][HU]   post-dropout h of this group

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    int nstages = 0;
    for (int s = 0; s < st.nseg; ++s) nstages += st.seg[s].k / BKT;

    f32x4 ra[TL::QA], rb[TL::QB];
    int seg = 0, kin = 0;                            // position of the NEXT stage to load
    auto load_next = [&]() {
        const LstmSeg sg = st.seg[seg];
        load_rows<MT * 32, BKT, TL::QA>([&](int row, bool& ok) { ok = row < B; return sg.x + (long)(ok ? row : 0) * sg.ldx; }, kin, ra);
        // column n = gate*8 + unit  ->  row (n/8)*H + u0 + n%8 of W
        load_rows<32, BKT, TL::QB>([&](int n, bool& ok) { ok = true; return sg.w + (long)((n >> 3) * H + u0 + (n & 7)) * sg.ldw; }, kin, rb);
        kin += BKT;
        if (kin >= sg.k) { kin = 0; ++seg; }
    };
    if (nstages > 0) load_next();
    // request the tail's operands now — but AFTER the first stage's operands (vector-memory operations complete
    // in order: stage 0 must not queue behind these cold reads)
    TailRegs<MT> tr;
    tail_prefetch<MT>(d, st, u0, tr);

    if (nstages > 0) {
        store_rows<BKT, TL::QA>(As, ra);
        store_rows<BKT, TL::QB>(Bs, rb);
    }
    __syncthreads();
    T2_STAMP(1);
    for (int c = 0; c < nstages; ++c) {
        const bool more = c + 1 < nstages;
        if (more) load_next();                       // the whole next stage is in flight during the MFMAs
        compute_stage_fwd<MT, BKT>(As, Bs, wave, r, hk, acc);
        __syncthreads();
        if (more) {
            store_rows<BKT, TL::QA>(As, ra);
            store_rows<BKT, TL::QB>(Bs, rb);
            __syncthreads();
        }
    }

    T2_STAMP(2);
    lstm_tail<MT>(d, st, u0, wave, r, hk, acc, part, hs, tr);
    T2_STAMP(4);
}

// ---------------------------------------------------------------------------------------------
// bf16-operand step (t2_set_precision(1), teacher-forced passes): the recurrent operands come from
// bf16 shadows — weights cast once per pass ([W_hh | W_ih[:, P:]] as ONE K-contiguous matrix) and
// the bf16 copies of h / ctx that the producing kernels write next to the fp32 ones — so a step
// moves half the bytes and the weight slices stay resident in the per-XCD L2s.  fp32 accumulate,
// fp32 gates / cell state; v_mfma_f32_32x32x16_bf16.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MT, int BKT> struct Tile16 {
    static constexpr int P = BKT + 8;                          // bf16 elements: (BKT+8)*2 B = odd number of 16-B slots
    static constexpr int A_ELEMS = MT * 32 * P, B_ELEMS = 32 * P;
    static constexpr int QA = MT * 32 * (BKT / 8) / NTH, QB = 32 * (BKT / 8) / NTH;
    static constexpr int STAGE_FLOATS = (A_ELEMS + B_ELEMS + 1) / 2;
    static constexpr int PART_FLOATS = (MT >= 2 ? 2 : 1) * NW * 32 * PP;
    static constexpr int SMEM_FLOATS = (STAGE_FLOATS > PART_FLOATS ? STAGE_FLOATS : PART_FLOATS);
};

template <int BKT, int Q, typename RowPtr>
__device__ __forceinline__ void load_rows16(RowPtr rowptr, int k0, bf16x8 (&regs)[Q]) {
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const int q = threadIdx.x + i * NTH, row = q / (BKT / 8), k = (q % (BKT / 8)) * 8;
        bool ok;
        const __bf16* p = rowptr(row, ok);
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(p + k0 + k);
        bf16x8 zz;
#pragma unroll
        for (int j = 0; j < 8; ++j) zz[j] = (__bf16)0.f;
        regs[i] = ok ? x : zz;
    }
}
template <int BKT, int Q>
__device__ __forceinline__ void store_rows16(__bf16* __restrict__ lds, const bf16x8 (&regs)[Q]) {
    constexpr int P = BKT + 8;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const int q = threadIdx.x + i * NTH, row = q / (BKT / 8), k = (q % (BKT / 8)) * 8;
        *reinterpret_cast<bf16x8*>(lds + row * P + k) = regs[i];
    }
}
template <int MT, int BKT>
__device__ __forceinline__ void compute_stage16(const __bf16* __restrict__ As, const __bf16* __restrict__ Bs, int wave, int r, int hk,
                                                f32x16 (&acc)[MT]) {
    constexpr int P = BKT + 8;
#pragma unroll
    for (int kk = 0; kk < BKT / NW; kk += 16) {
        const int k = wave * (BKT / NW) + kk + 8 * hk;
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + r * P + k);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + (m * 32 + r) * P + k);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
        }
    }
}

template <int MT, int BKT>
__global__ __launch_bounds__(NTH) void lstm_step_fwd_bf16_kernel(LstmStepDesc d) {
    using TL = Tile16<MT, BKT>;
    const LstmStream& st = d.st[blockIdx.y];
    const int B = d.B, H = d.H;
    const int u0 = blockIdx.x * HU;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hk = lane >> 5;

    T2_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);
    __bf16* Bs = As + TL::A_ELEMS;
    float* part = smem;
    float* hs = smem + TL::SMEM_FLOATS;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    // Two register stage buffers: stage c+2 is requested while stage c is computed and stage c+1 is still in
    // flight, so a workgroup keeps two full stages (~2 x 64 KB) outstanding — the K loop is a chain of L2/MALL
    // round trips and one stage of prefetch left the CU idle for most of each trip.
    const int nstages = st.k16 / BKT;
#ifdef T2_STAMPS
    {   // latency probes: one cold weight element, one cold activation element, one more of each (warm TLB?)
        const volatile __bf16* wp = st.w16 + (long)u0 * st.ldw16;
        float v0 = (float)wp[0];
        __builtin_amdgcn_s_waitcnt(0);
        T2_STAMP(5);
        const volatile __bf16* xp = st.x16;
        v0 += (float)xp[0];
        __builtin_amdgcn_s_waitcnt(0);
        T2_STAMP(6);
        v0 += (float)wp[4096] + (float)xp[4096];
        __builtin_amdgcn_s_waitcnt(0);
        T2_STAMP(7);
        if (v0 == 123.456f) hs[0] = v0;
    }
#endif
    bf16x8 ra0[TL::QA], rb0[TL::QB], ra1[TL::QA], rb1[TL::QB];
    auto load_stage = [&](int c, bf16x8 (&ra)[TL::QA], bf16x8 (&rb)[TL::QB]) {
        load_rows16<BKT, TL::QA>([&](int row, bool& ok) { ok = row < B; return st.x16 + (long)(ok ? row : 0) * st.ldx16; }, c * BKT, ra);
        load_rows16<BKT, TL::QB>([&](int n, bool& ok) { ok = true; return st.w16 + (long)((n >> 3) * H + u0 + (n & 7)) * st.ldw16; }, c * BKT, rb);
    };
    if (nstages > 0) load_stage(0, ra0, rb0);
    if (nstages > 1) load_stage(1, ra1, rb1);
    TailRegs<MT> tr;
    tail_prefetch<MT>(d, st, u0, tr);
    if (nstages > 0) { store_rows16<BKT, TL::QA>(As, ra0); store_rows16<BKT, TL::QB>(Bs, rb0); }
    __syncthreads();
    T2_STAMP(1);
    for (int c = 0; c < nstages; c += 2) {
        if (c + 2 < nstages) load_stage(c + 2, ra0, rb0);
        compute_stage16<MT, BKT>(As, Bs, wave, r, hk, acc);
        __syncthreads();
        if (c + 1 < nstages) {
            store_rows16<BKT, TL::QA>(As, ra1); store_rows16<BKT, TL::QB>(Bs, rb1);
            __syncthreads();
            if (c + 3 < nstages) load_stage(c + 3, ra1, rb1);
            compute_stage16<MT, BKT>(As, Bs, wave, r, hk, acc);
            __syncthreads();
            if (c + 2 < nstages) { store_rows16<BKT, TL::QA>(As, ra0); store_rows16<BKT, TL::QB>(Bs, rb0); __syncthreads(); }
        }
    }
    T2_STAMP(2);
    lstm_tail<MT>(d, st, u0, wave, r, hk, acc, part, hs, tr);
    T2_STAMP(4);
}

// ---------------------------------------------------------------------------------------------
// backward, part 1: pointwise.  One thread per (b, u).
//   dh  = direct sources + sum of recurrent partials (+ dq . Wq)          gradient on h_out(t)
//   dhn = dh * keep_h/(1-p) ;  dcn = dc_out * keep_c/(1-p) + dhn * o * (1 - tanh(cn)^2)
//   d(pre-activations) = { dcn*g*i(1-i), dcn*c_prev*f(1-f), dcn*i*(1-g^2), dhn*tanh(cn)*o(1-o) }
//   dc_state <- dcn * f                                                    gradient on c_out(t-1)
// ---------------------------------------------------------------------------------------------
// shared pointwise math of the two kernels below
struct PwIn { float dh, dc, ig, fg, gg, og, cn, cp; };
__device__ __forceinline__ void lstm_bwd_point(const LstmBwdPointDesc& d, const LstmBwdStream& st, int b, int u, PwIn in) {
    const int H = d.H;
    float dh = in.dh, dc = in.dc;
    if (d.drop_p > 0.f) {
        const uint32_t idx = st.idx_base + (uint32_t)b * st.idx_bstride + (uint32_t)u;
        const float scale = 1.0f / (1.0f - d.drop_p);
        dh = rng_keep(rng_key(d.seed, st.site_h), idx, d.drop_p) ? dh * scale : 0.f;
        dc = rng_keep(rng_key(d.seed, st.site_c), idx, d.drop_p) ? dc * scale : 0.f;
    }
    const float tc = tanhf(in.cn);
    const float dcn = dc + dh * in.og * (1.0f - tc * tc);
    const float d0 = dcn * in.gg * in.ig * (1.0f - in.ig), d1 = dcn * in.cp * in.fg * (1.0f - in.fg);
    const float d2 = dcn * in.ig * (1.0f - in.gg * in.gg), d3 = dh * tc * in.og * (1.0f - in.og);
    float* dg = st.dg + (long)b * st.lddg + u;
    dg[0] = d0; dg[H] = d1; dg[2 * H] = d2; dg[3 * H] = d3;
    if (st.dg16) {                                   // bf16 copy for the bf16-operand recurrent GEMM of this step
        __bf16* g16 = st.dg16 + (long)b * 4 * H + u;
        g16[0] = (__bf16)d0; g16[H] = (__bf16)d1; g16[2 * H] = (__bf16)d2; g16[3 * H] = (__bf16)d3;
    }
    st.dc_state[(long)b * H + u] = dcn * in.fg;
}
__device__ __forceinline__ PwIn lstm_bwd_load(const LstmBwdPointDesc& d, const LstmBwdStream& st, int b, int u) {
    const int H = d.H;
    PwIn in;
    float dh = 0.f;
    if (st.dh1) dh += st.dh1[(long)b * st.lddh1 + u];
    if (st.dh2) dh += st.dh2[(long)b * st.lddh2 + u];
    if (st.part && !d.first) {
        const float* p = st.part + (long)b * st.ldpart + st.part_col + u;
        float pv[8];                                  // all K-split partials requested at once (nparts <= 8)
#pragma unroll
        for (int z = 0; z < 8; ++z) pv[z] = z < st.nparts ? p[(long)z * st.part_stride] : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int z = 0; z < 8; ++z) acc += pv[z];
        dh += acc;
    }
    in.dh = dh;
    in.dc = d.first ? 0.f : st.dc_state[(long)b * H + u];
    const float* gp = st.gates + (long)b * st.ldgates + u;
    in.ig = gp[0]; in.fg = gp[H]; in.gg = gp[2 * H]; in.og = gp[3 * H];
    in.cn = st.c_new[(long)b * st.ldc_new + u];
    in.cp = st.c_prev ? st.c_prev[(long)b * st.ldc_prev + u] : 0.f;
    return in;
}

// no query projection (decoder / encoder LSTMs): one thread per (b, u)
__global__ __launch_bounds__(256) void lstm_bwd_pointwise_kernel(LstmBwdPointDesc d) {
    const LstmBwdStream& st = d.st[blockIdx.y];
    const int B = d.B, H = d.H;
    const long i = blockIdx.x * 256l + threadIdx.x;
    if (i >= (long)B * H) return;
    const int b = (int)(i / H), u = (int)(i % H);
    lstm_bwd_point(d, st, b, u, lstm_bwd_load(d, st, b, u));
}

// with the query-projection term dq . Wq (attention LSTMs): a workgroup owns PWU hidden units x PWB batch rows and
// stages its [A x PWU] slice of Wq in LDS once.  One thread per (b, u) re-read the whole 0.5 MB Wq once per batch
// row — 64 MB of L2->CU traffic per step, the per-CU L2 rate (~60 GB/s) made that the kernel's floor.
constexpr int PWU = 64, PWB = 8;
__global__ __launch_bounds__(256) void lstm_bwd_pointwise_q_kernel(LstmBwdPointDesc d) {
    const LstmBwdStream& st = d.st[blockIdx.z];
    const int B = d.B, H = d.H, A = st.A;
    const int ul = threadIdx.x & (PWU - 1), bs = threadIdx.x / PWU;            // bs in 0..3, rows bs and bs + 4
    const int u = blockIdx.x * PWU + ul, b0 = blockIdx.y * PWB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wqs = smem;                    // [A][PWU]
    float* dqs = smem + A * PWU;          // [PWB][A]
    // per-item operands first (cold), then the Wq slice (L2-resident)
    PwIn in[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) in[r] = lstm_bwd_load(d, st, min(b0 + bs + 4 * r, B - 1), u);
    if (st.dq) {
        for (int a = bs; a < A; a += 4) wqs[a * PWU + ul] = st.wq[(long)a * H + u];
        const int parts = st.dq_parts > 0 ? st.dq_parts : 1;
        for (int i = threadIdx.x; i < PWB * A; i += 256) {
            const float* row = st.dq + (long)min(b0 + i / A, B - 1) * st.lddq + i % A;
            float v = row[0];
            for (int p = 1; p < parts; ++p) v += row[p * A];
            dqs[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float* q = dqs + (bs + 4 * r) * A;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int a = 0; a < A; a += 4) {
                a0 += q[a] * wqs[a * PWU + ul]; a1 += q[a + 1] * wqs[(a + 1) * PWU + ul];
                a2 += q[a + 2] * wqs[(a + 2) * PWU + ul]; a3 += q[a + 3] * wqs[(a + 3) * PWU + ul];
            }
            in[r].dh += (a0 + a1) + (a2 + a3);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int b = b0 + bs + 4 * r;
        if (b < B) lstm_bwd_point(d, st, b, u, in[r]);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, part 2: part[z][b][n] = sum_{k in K-split z} dg[b,k] * W[k,n]   for the recurrent
// input columns n (ctx | h for the attention LSTMs, h for the decoder / encoder LSTMs).
// grid = (column tiles of 32, K-splits, streams), so the whole chip works on one step; same
// LDS-staged chunk pipeline as the forward step (A = dg rows, B = W rows k, n-contiguous).
// ---------------------------------------------------------------------------------------------
template <int MT, int BKT>
__global__ __launch_bounds__(NTH) void lstm_bwd_gemm_kernel(LstmBwdGemmDesc d) {
    using TL = Tile<MT, BKT>;
    const LstmBwdGemmStream& st = d.st[blockIdx.z];
    const int B = d.B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hk = lane >> 5;
    // locate this column tile
    int col0 = blockIdx.x * 32, sidx = 0, cbase = 0;
    while (sidx < st.nseg - 1 && col0 >= cbase + st.seg[sidx].ncols) { cbase += st.seg[sidx].ncols; ++sidx; }
    const LstmBwdSeg sg = st.seg[sidx];
    const int kspan = d.H4 / d.KS, kbeg = blockIdx.y * kspan;
    const int nstages = kspan / BKT;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + TL::A_FLOATS;
    float* part = smem;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    // B tile: BKT k-rows x 32 columns, n-contiguous: thread quad q -> (k = q/8, 4 n's)
    const float* wbase = sg.w + (long)kbeg * sg.ldw + (col0 - cbase);
    const bool wvec = (sg.ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(sg.w) & 15) == 0);
    f32x4 ra[TL::QA], rb[TL::QB];
    auto load_stage = [&](int c) {
        load_rows<MT * 32, BKT, TL::QA>([&](int row, bool& ok) { ok = row < B; return st.dg + (long)(ok ? row : 0) * st.lddg + kbeg; }, c * BKT, ra);
#pragma unroll
        for (int i = 0; i < TL::QB; ++i) {
            const int q = threadIdx.x + i * NTH, k = q >> 3, n = (q & 7) * 4;
            const float* p = wbase + (long)(c * BKT + k) * sg.ldw + n;
            if (wvec) rb[i] = *reinterpret_cast<const f32x4*>(p);
            else rb[i] = f32x4{p[0], p[1], p[2], p[3]};
        }
    };
    auto store_stage = [&]() {
        store_rows<BKT, TL::QA>(As, ra);
#pragma unroll
        for (int i = 0; i < TL::QB; ++i) {
            const int q = threadIdx.x + i * NTH, k = q >> 3, n = (q & 7) * 4;
            *reinterpret_cast<f32x4*>(Bs + k * TL::PBN + n) = rb[i];
        }
    };
    load_stage(0);
    store_stage();
    __syncthreads();
    for (int c = 0; c < nstages; ++c) {
        const bool more = c + 1 < nstages;
        if (more) load_stage(c + 1);
        compute_stage_bwd<MT, BKT>(As, Bs, wave, r, hk, acc);
        __syncthreads();
        if (more) { store_stage(); __syncthreads(); }
    }

    float* out = st.part + (long)blockIdx.y * B * d.NC;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m > 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e)
            part[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PP + r] = acc[m][e];
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * 32; i += NTH) {
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

// bf16-operand variant of the recurrent-input gradient: A = bf16 copy of dg(t), B = transposed bf16
// weight shadow wt16[n][k] (K-contiguous), K-split z covers k in [z*kspan, (z+1)*kspan).
template <int MT>
__global__ __launch_bounds__(NTH) void lstm_bwd_gemm_bf16_kernel(LstmBwdGemmDesc d) {
    constexpr int BKT = 256;
    using TL = Tile16<MT, BKT>;
    const LstmBwdGemmStream& st = d.st[blockIdx.z];
    const int B = d.B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hk = lane >> 5;
    const int col0 = blockIdx.x * 32;
    const int kspan = d.H4 / d.KS, kbeg = blockIdx.y * kspan, nstages = kspan / BKT;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);
    __bf16* Bs = As + TL::A_ELEMS;
    float* part = smem;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    bf16x8 ra[TL::QA], rb[TL::QB];
    auto load_stage = [&](int c) {
        load_rows16<BKT, TL::QA>([&](int row, bool& ok) { ok = row < B; return st.dg16 + (long)(ok ? row : 0) * d.H4 + kbeg; }, c * BKT, ra);
        load_rows16<BKT, TL::QB>([&](int n, bool& ok) { ok = true; return st.wt16 + (long)(col0 + n) * d.H4 + kbeg; }, c * BKT, rb);
    };
    load_stage(0);
    store_rows16<BKT, TL::QA>(As, ra); store_rows16<BKT, TL::QB>(Bs, rb);
    __syncthreads();
    for (int c = 0; c < nstages; ++c) {
        const bool more = c + 1 < nstages;
        if (more) load_stage(c + 1);
        compute_stage16<MT, BKT>(As, Bs, wave, r, hk, acc);
        __syncthreads();
        if (more) { store_rows16<BKT, TL::QA>(As, ra); store_rows16<BKT, TL::QB>(Bs, rb); __syncthreads(); }
    }
    float* out = st.part + (long)blockIdx.y * B * d.NC;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m > 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e)
            part[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PP + r] = acc[m][e];
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * 32; i += NTH) {
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

// dst[r*ldd + c] = bf16(src[r*lds + c])
__global__ void cast_rows_kernel(const float* __restrict__ src, long lds_, __bf16* __restrict__ dst, long ldd, int R, int C) {
    const size_t n = (size_t)R * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const size_t rr = i / C;
        dst[rr * ldd + c] = (__bf16)src[rr * lds_ + c];
    }
}
// dst[c*ldd + r] = bf16(src[r*lds + c])   (32x32 LDS tiles)
__global__ void cast_transpose_kernel(const float* __restrict__ src, long lds_, __bf16* __restrict__ dst, long ldd, int R, int C) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int rr = r0 + i, c = c0 + tx;
        tile[i][tx] = (rr < R && c < C) ? src[(size_t)rr * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, rr = r0 + tx;
        if (c < C && rr < R) dst[(size_t)c * ldd + rr] = (__bf16)tile[tx][i];
    }
}

template <int MT, int BKT> size_t fwd_smem() { return (size_t)(Tile<MT, BKT>::FWD_FLOATS + MT * 32 * HU) * sizeof(float); }
template <int MT, int BKT> size_t bwd_smem() { return (size_t)Tile<MT, BKT>::BWD_FLOATS * sizeof(float); }

template <typename K>
int allow_big_lds(K kernel, size_t smem) { return t2_allow_dynamic_lds(reinterpret_cast<const void*>(kernel), smem); }

// stage width by batch tile count (LDS budget) when every K extent allows it, else 64
#define LAUNCH_ONE(KERNEL, SMEM, MTV, BKV, GRID, BLOCK, STREAM, DESC)                                 \
    do {                                                                                              \
        const size_t sm_ = SMEM<MTV, BKV>();                                                          \
        int rc_ = allow_big_lds(KERNEL<MTV, BKV>, sm_);                                               \
        if (rc_) return rc_;                                                                          \
        hipLaunchKernelGGL((KERNEL<MTV, BKV>), GRID, BLOCK, sm_, STREAM, DESC);                       \
    } while (0)
#define LAUNCH_MT(KERNEL, SMEM, KDIV, GRID, BLOCK, STREAM, DESC)                                      \
    do {                                                                                              \
        if (MT <= 1) { if ((KDIV) % 256 == 0) LAUNCH_ONE(KERNEL, SMEM, 1, 256, GRID, BLOCK, STREAM, DESC); else LAUNCH_ONE(KERNEL, SMEM, 1, 64, GRID, BLOCK, STREAM, DESC); } \
        else if (MT <= 2) { if ((KDIV) % 256 == 0) LAUNCH_ONE(KERNEL, SMEM, 2, 256, GRID, BLOCK, STREAM, DESC); else LAUNCH_ONE(KERNEL, SMEM, 2, 64, GRID, BLOCK, STREAM, DESC); } \
        else if (MT <= 4) { if ((KDIV) % 128 == 0) LAUNCH_ONE(KERNEL, SMEM, 4, 128, GRID, BLOCK, STREAM, DESC); else LAUNCH_ONE(KERNEL, SMEM, 4, 64, GRID, BLOCK, STREAM, DESC); } \
        else LAUNCH_ONE(KERNEL, SMEM, 8, 64, GRID, BLOCK, STREAM, DESC);                              \
    } while (0)

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
            T2_REQUIRE(g.k % 64 == 0, "lstm_step: segment width %d must be a multiple of 64", g.k);
            T2_REQUIRE(g.ldx % 4 == 0 && g.ldw % 4 == 0 && ((uintptr_t)g.x & 15) == 0 && ((uintptr_t)g.w & 15) == 0,
                       "lstm_step: segment %d operands must be 16-byte aligned (ldx=%ld ldw=%ld)", j, g.ldx, g.ldw);
        }
        T2_REQUIRE(!st.wq || st.A * HU <= NW * 32 * PP, "lstm_step: attention dim %d too large", st.A);
    }
    const int MT = (d.B + 31) / 32;
    dim3 grid(d.H / HU, d.nstreams), block(NTH);
    bool bf = d.st[0].x16 != nullptr;
    if (bf) {
        for (int i = 0; i < d.nstreams; ++i)
            T2_REQUIRE(d.st[i].x16 && d.st[i].w16 && d.st[i].k16 % 256 == 0 && d.st[i].ldx16 % 8 == 0 && d.st[i].ldw16 % 8 == 0 &&
                       ((uintptr_t)d.st[i].x16 & 15) == 0 && ((uintptr_t)d.st[i].w16 & 15) == 0, "lstm_step: bad bf16 operands");
        T2_REQUIRE(MT <= 4, "lstm_step: the bf16-operand step supports B <= 128");
        const bool wide = d.st[0].k16 % 512 == 0 && (d.nstreams < 2 || d.st[1].k16 % 512 == 0);
        auto go = [&](auto kernel, size_t smem) -> int {
            T2_TRY_RC(allow_big_lds(kernel, smem));
            hipLaunchKernelGGL(kernel, grid, block, smem, s, d);
            return 0;
        };
        if (MT <= 1) {
            if (wide) T2_TRY_RC(go(lstm_step_fwd_bf16_kernel<1, 512>, (size_t)(Tile16<1, 512>::SMEM_FLOATS + 32 * HU) * 4));
            else T2_TRY_RC(go(lstm_step_fwd_bf16_kernel<1, 256>, (size_t)(Tile16<1, 256>::SMEM_FLOATS + 32 * HU) * 4));
        } else if (MT <= 2) {
            if (wide) T2_TRY_RC(go(lstm_step_fwd_bf16_kernel<2, 512>, (size_t)(Tile16<2, 512>::SMEM_FLOATS + 64 * HU) * 4));
            else T2_TRY_RC(go(lstm_step_fwd_bf16_kernel<2, 256>, (size_t)(Tile16<2, 256>::SMEM_FLOATS + 64 * HU) * 4));
        } else {                                         // 65..128 rows: 256-wide stages keep the A tile within LDS
            T2_TRY_RC(go(lstm_step_fwd_bf16_kernel<4, 256>, (size_t)(Tile16<4, 256>::SMEM_FLOATS + 128 * HU) * 4));
        }
        T2_LAUNCH_CHECK();
        return 0;
    }
    int kdiv = 0;                                    // gcd-like: every segment width must be a multiple of the stage width
    for (int i = 0; i < d.nstreams; ++i)
        for (int j = 0; j < d.st[i].nseg; ++j) kdiv = kdiv == 0 ? d.st[i].seg[j].k : std::gcd(kdiv, d.st[i].seg[j].k);
    if (kdiv == 0) kdiv = 256;
    LAUNCH_MT(lstm_step_fwd_kernel, fwd_smem, kdiv, grid, block, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

int lstm_bwd_pointwise(const LstmBwdPointDesc& d, hipStream_t s) {
    T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= kMaxLstmStreams, "lstm_bwd_pointwise: nstreams=%d", d.nstreams);
    bool hasq = false;
    int amax = 0;
    for (int i = 0; i < d.nstreams; ++i) {
        T2_REQUIRE(!d.st[i].dq || (d.st[i].A % 4 == 0 && d.st[i].A <= 256), "lstm_bwd_pointwise: attention dim %d unsupported", d.st[i].A);
        hasq = hasq || d.st[i].dq;
        if (d.st[i].dq) amax = std::max(amax, d.st[i].A);
    }
    if (hasq && d.H % PWU == 0) {
        const size_t smem = (size_t)amax * (PWU + PWB) * sizeof(float);
        T2_TRY_RC(allow_big_lds(lstm_bwd_pointwise_q_kernel, smem));
        hipLaunchKernelGGL(lstm_bwd_pointwise_q_kernel, dim3(d.H / PWU, (d.B + PWB - 1) / PWB, d.nstreams), dim3(256), smem, s, d);
    } else {
        T2_REQUIRE(!hasq, "lstm_bwd_pointwise: H=%d must be a multiple of %d with a query projection", d.H, PWU);
        const long n = (long)d.B * d.H;
        hipLaunchKernelGGL(lstm_bwd_pointwise_kernel, dim3((unsigned)((n + 255) / 256), d.nstreams), dim3(256), 0, s, d);
    }
    T2_LAUNCH_CHECK();
    return 0;
}

int lstm_bwd_ksplit(int H4) {
    int ks = 8;
    while (ks > 1 && H4 % (ks * 64) != 0) ks >>= 1;
    return ks;
}

int lstm_bwd_gemm(const LstmBwdGemmDesc& d, hipStream_t s) {
    T2_REQUIRE(d.nstreams >= 1 && d.nstreams <= kMaxLstmStreams, "lstm_bwd_gemm: nstreams=%d", d.nstreams);
    T2_REQUIRE(d.B >= 1 && d.B <= 256, "lstm_bwd_gemm: batch %d", d.B);
    T2_REQUIRE(d.KS >= 1 && d.H4 % (d.KS * 64) == 0, "lstm_bwd_gemm: 4H=%d not divisible by KS*64 (KS=%d)", d.H4, d.KS);
    int nc = 0;
    for (int j = 0; j < d.st[0].nseg; ++j) nc += d.st[0].seg[j].ncols;
    T2_REQUIRE(nc == d.NC && nc % 32 == 0, "lstm_bwd_gemm: column count %d (NC=%d) must be a multiple of 32", nc, d.NC);
    for (int i = 0; i < d.nstreams; ++i) {
        T2_REQUIRE(d.st[i].lddg % 4 == 0 && ((uintptr_t)d.st[i].dg & 15) == 0, "lstm_bwd_gemm: dg must be 16-byte aligned");
        for (int j = 0; j < d.st[i].nseg; ++j) T2_REQUIRE(d.st[i].seg[j].ncols % 32 == 0, "lstm_bwd_gemm: segment cols %d", d.st[i].seg[j].ncols);
    }
    const int MT = (d.B + 31) / 32;
    dim3 grid(d.NC / 32, d.KS, d.nstreams), block(NTH);
    if (d.st[0].dg16) {
        T2_REQUIRE(MT <= 4 && (d.H4 / d.KS) % 256 == 0, "lstm_bwd_gemm: bf16 variant needs B <= 128 and K-split spans of 256");
        if (MT <= 1) {
            const size_t smem = (size_t)Tile16<1, 256>::SMEM_FLOATS * 4;
            T2_TRY_RC(allow_big_lds(lstm_bwd_gemm_bf16_kernel<1>, smem));
            hipLaunchKernelGGL(lstm_bwd_gemm_bf16_kernel<1>, grid, block, smem, s, d);
        } else if (MT <= 2) {
            const size_t smem = (size_t)Tile16<2, 256>::SMEM_FLOATS * 4;
            T2_TRY_RC(allow_big_lds(lstm_bwd_gemm_bf16_kernel<2>, smem));
            hipLaunchKernelGGL(lstm_bwd_gemm_bf16_kernel<2>, grid, block, smem, s, d);
        } else {
            const size_t smem = (size_t)Tile16<4, 256>::SMEM_FLOATS * 4;
            T2_TRY_RC(allow_big_lds(lstm_bwd_gemm_bf16_kernel<4>, smem));
            hipLaunchKernelGGL(lstm_bwd_gemm_bf16_kernel<4>, grid, block, smem, s, d);
        }
        T2_LAUNCH_CHECK();
        return 0;
    }
    LAUNCH_MT(lstm_bwd_gemm_kernel, bwd_smem, d.H4 / d.KS, grid, block, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

int cast_rows_bf16(const float* src, long ld_src, __bf16* dst, long ld_dst, int R, int C, hipStream_t s) {
    const size_t n = (size_t)R * C;
    size_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)g), dim3(256), 0, s, src, ld_src, dst, ld_dst, R, C);
    T2_LAUNCH_CHECK();
    return 0;
}
int cast_transpose_bf16(const float* src, long ld_src, __bf16* dst, long ld_dst, int R, int C, hipStream_t s) {
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, src, ld_src, dst, ld_dst, R, C);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
