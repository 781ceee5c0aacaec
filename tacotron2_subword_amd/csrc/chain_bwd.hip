// Persistent backward chains: all steps (reverse time) of an LSTM recurrence's BPTT in ONE launch.
//
//   decoder-LSTM chain (autograd of model.py:371-373):  per step  P  dL/dh, dL/dc -> gate gradients dg(t)      (pointwise)
//                                                                 G  dx(t) = dg(t) . W_hh   -> dL/dh of step t-1 (skinny GEMM)
//
// As launches, every step re-fetched W_hh^T (8 MB of bf16) for the GEMM and paid two dependent launch + operand
// delivery floors (8.9 + 10.2 us at B = 64).  Here the transposed weights stay in registers for all steps:
//   G item (nt, kp): 32 output columns x one eighth of K = 4H (512 gate columns): 32 KB of W^T per workgroup, 256 items.
//       Its 8 waves split the 512 further; fragments of dg(t) come straight from the fragment-ordered exchange buffer
//       into MFMA registers; the 8 partial tiles are summed through LDS in fixed order and the [64 x 32] result goes out
//       as one K-split partial (write-through, 16 B per lane) in the layout the consumer reads contiguously.
//   P item (ug, rg): 16 hidden units x 32 batch rows, one (row, unit) per thread: sums the 8 K-split partials of dx(t+1)
//       in fixed order, adds the direct gradient, dropout masks, gate derivatives; keeps dL/dc in a register across
//       steps; publishes dg(t) as four bf16 MFMA fragments (one per gate) and stores the fp32 dg(t) rows the
//       weight-gradient GEMMs read afterwards.
// Hand-offs: chain_common.h (one arrival counter per consumer group: the two unit halves for G, the 32 column tiles for
// P).  Two all-to-all hops per step instead of two launches; numerics = the launch path (lstm.hip) up to summation order.
#include <algorithm>

#include "chain_common.h"

namespace t2 {

namespace {

using namespace chain;

constexpr int GNC = 32;          // output columns of a G item
constexpr int GKP = 8;           // K parts
constexpr int PU = 16;           // hidden units of a P item

template <int MT>
__global__ __launch_bounds__(NTH) void chain_bwd_lstm_kernel(ChainBwdDesc d) {
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hk = lane >> 5;
    const int B = d.B, H = d.H, K4 = 4 * H;
    const ChainBwdStream& S = d.st[0];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* abortw = reinterpret_cast<unsigned*>(smem);
    float* partL = smem + 4;                                  // [NWV][MT][32][PPR]   (G phase)
    float* dgL = smem + 4;                                    // [4 gates][32 rows][PU + 4]   (P phase; the phases alternate)

    // ---------------------------------------------------------------- items
    const int NT = H / GNC;                                   // column tiles
    const int KPW = K4 / GKP, KSW = KPW / NWV / 16;           // k per part, k-steps of 16 per wave (4 at H = 1024)
    const int nG = NT * GKP, nP = (H / PU) * MT;
    const bool hasG = wg < nG, hasP = wg < nP;
    const int nt = wg % NT, kp = wg / NT;                     // G item
    const int ug = wg / MT, rt = wg % MT;                     // P item: units [ug*16, +16), rows [rt*32, +32)
    const int u0 = ug * PU;
    const int KT = K4 / 16;
    auto rsX = __builtin_amdgcn_make_buffer_rsrc(d.X, 0, (int)(2u * KT * MT * 1024u), 0x00020000);
    auto rsP = __builtin_amdgcn_make_buffer_rsrc(d.PB, 0, (int)d.pb_bytes, 0x00020000);
    const unsigned pb_half = d.pb_bytes / 2;                  // one parity of the partial buffer
    unsigned* cntP = d.cnt + (size_t)(kp & 1) * CNT_STRIDE;   // G waits for the P items of its unit half
    unsigned* cntP_mine = d.cnt + (size_t)(u0 >= H / 2 ? 1 : 0) * CNT_STRIDE;
    unsigned* cntG_wait = d.cnt + (size_t)(2 + u0 / GNC) * CNT_STRIDE;    // P waits for the 8 K parts of its column tile
    unsigned* cntG_mine = d.cnt + (size_t)(2 + nt) * CNT_STRIDE;
    const unsigned nP_half = (unsigned)(H / 2 / PU * MT);

    if (tid == 0) *abortw = 0;

    // ---------------------------------------------------------------- G setup: W^T slice -> registers (once)
    // k range of wave w: [kp*KPW + w*KSW*16, +KSW*16); lane (j = r, hk) holds 8 consecutive k of output column nt*32 + j
    bf16x8 W[4];
    if (hasG) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            W[i] = *reinterpret_cast<const bf16x8*>(S.wt16 + (long)(nt * GNC + r) * S.ldwt + kp * KPW + (wave * KSW + min(i, KSW - 1)) * 16 + 8 * hk);
    }
    // ---------------------------------------------------------------- P state
    const int pb = rt * 32 + (tid >> 4), pu = u0 + (tid & 15);       // this thread's (row, unit)
    const bool resumed = d.t1 < d.T;                                  // a later launch of the same pass: state of step t1 is in memory
    float dc = (hasP && resumed && pb < B) ? S.dc_state[(long)pb * H + pu] : 0.f;
    float pin[7];                                                     // dh1, i, f, g, o, c_new, c_prev of the step to come
    auto load_pin = [&](int t, int tid) {
        const int b = min(rt * 32 + (tid >> 4), B - 1), u = u0 + (tid & 15);
        const long rb = (long)t * B + b;
        pin[0] = S.dh1[rb * S.lddh1 + u];
        const float* gp = S.gates + rb * K4 + u;
        pin[1] = gp[0]; pin[2] = gp[H]; pin[3] = gp[2 * H]; pin[4] = gp[3 * H];
        pin[5] = S.c_new[rb * H + u];
        pin[6] = t > 0 ? S.c_out[((long)(t - 1) * B + b) * H + u] : 0.f;
    };
    if (hasP) load_pin(d.t1 - 1, tid);
    __syncthreads();
    const RngKey kh = rng_key(d.seed, S.site_h), kc = rng_key(d.seed, S.site_c);
    const float dscale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;

    for (int t = d.t1 - 1; t >= d.t0; --t) {
        const unsigned ep = (unsigned)(d.t1 - 1 - t);                 // steps already done
        int tv = threadIdx.x;
        asm volatile("" : "+v"(tv));
        // ======================================================================================= P(t)
        if (hasP) {
            float in[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) in[i] = pin[i];
            float dh = in[0];
            if (ep > 0 || resumed) {
                if (ep > 0) {                                         // (a resumed launch finds step t1's partials complete)
                    if (wave == 0 && !poll_counter(cntG_wait, ep * (unsigned)GKP, d.err, 5u) && lane == 0) *abortw = 1;
                    __syncthreads();
                    if (*abortw) return;
                }
                // dx(t+1)[row, unit] = fixed-order sum of the 8 K-split partials (written by G(t+1) into parity (t+1)&1)
                const unsigned off = (unsigned)(((t + 1) & 1)) * pb_half + (unsigned)((((ug * MT + rt) * 32 + (tv >> 4)) * PU + (tv & 15)) * 4);
                const unsigned kstride = (unsigned)((H / PU) * MT * 32 * PU * 4);
                float pv[GKP];
#pragma unroll
                for (int z = 0; z < GKP; ++z) pv[z] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsP, off + z * kstride, 0, SC1));
                float acc = 0.f;
#pragma unroll
                for (int z = 0; z < GKP; ++z) acc += pv[z];
                dh += acc;
            }
            // pointwise BPTT (lstm.hip lstm_bwd_point): dropout masks, then the gate derivatives
            float dcs = (ep > 0 || resumed) ? dc : 0.f;
            if (d.drop_p > 0.f) {
                const uint32_t idx = (uint32_t)(((long)t * B + pb) * H + pu);
                dh = rng_keep(kh, idx, d.drop_p) ? dh * dscale : 0.f;
                dcs = rng_keep(kc, idx, d.drop_p) ? dcs * dscale : 0.f;
            }
            const float ig = in[1], fg = in[2], gg = in[3], og = in[4];
            const float tc = tanhf(in[5]);
            const float dcn = dcs + dh * og * (1.0f - tc * tc);
            float dgv[4] = {dcn * gg * ig * (1.0f - ig), dcn * in[6] * fg * (1.0f - fg), dcn * ig * (1.0f - gg * gg), dh * tc * og * (1.0f - og)};
            if (pb >= B) { dgv[0] = dgv[1] = dgv[2] = dgv[3] = 0.f; }
            dc = dcn * fg;
            if (ep > 0) __syncthreads();                              // (dgL aliases nothing of this phase, but G's partL of the same step follows)
#pragma unroll
            for (int g = 0; g < 4; ++g) dgL[(g * 32 + (tid >> 4)) * (PU + 4) + (tid & 15)] = dgv[g];
            __syncthreads();
            // dg(t) as four bf16 fragments: gate g covers k tile (g*H + u0)/16; lane (row r, half hk) holds 8 units
            if (wave < 4) {
                const float* hp = dgL + (wave * 32 + r) * (PU + 4) + hk * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(hp), hi = *reinterpret_cast<const f32x4*>(hp + 4);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsX,
                    (unsigned)(t & 1) * (unsigned)(KT * MT * 1024) + (unsigned)((((wave * H + u0) / 16) * MT + rt) * 1024 + lane * 16), 0, SC1);
            }
            publish(cntP_mine);
            {   // fp32 dg(t) rows for the weight-gradient GEMMs, then next step's operands (cold HBM rows) behind them
                const int b = rt * 32 + (tv >> 4), u = u0 + (tv & 15);
                if (b < B) {
                    float* gp = S.dg + ((long)t * B + b) * K4 + u;
                    gp[0] = dgv[0]; gp[H] = dgv[1]; gp[2 * H] = dgv[2]; gp[3 * H] = dgv[3];
                }
                if (t > d.t0) load_pin(t - 1, tv);
            }
        }
        // ======================================================================================= G(t)
        if (hasG && t > 0) {                                          // (dx(0) feeds nothing)
            if (wave == 0 && !poll_counter(cntP, (ep + 1) * nP_half, d.err, 6u) && lane == 0) *abortw = 1;
            __syncthreads();
            if (*abortw) return;
            f32x16 acc[MT];
            const unsigned xb = (unsigned)(t & 1) * (unsigned)(KT * MT * 1024) + (unsigned)lane * 16u;
            u32x4 af[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    af[m][i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xb + (unsigned)(((kp * KPW / 16 + wave * KSW + min(i, KSW - 1)) * MT + m) * 1024), 0, SC1);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < KSW) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[m][i]), W[i], acc[m], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    partL[((wave * MT + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PPR + r] = acc[m][e];
            }
            __syncthreads();
            // thread (row tile m, row, 4 columns): fixed-order sum of the 8 waves' partials, one 16-byte write-through store
            // into PB[parity t&1][kp][unit group of 16][row tile][row][16 units]
            if (tv < MT * 256) {
                const int m = tv >> 8, row = (tv & 255) >> 3, c4 = (tv & 7) * 4;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < NWV; ++w) sum += *reinterpret_cast<const f32x4*>(partL + ((w * MT + m) * 32 + row) * PPR + c4);
                const int ugo = nt * (GNC / PU) + c4 / PU;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rsP,
                    (unsigned)(t & 1) * pb_half + (unsigned)((((kp * (H / PU) + ugo) * MT + m) * 32 + row) * PU + (c4 % PU)) * 4u, 0, SC1);
            }
            publish(cntG_mine);
        }
    }
    if (hasP && d.t0 > 0 && pb < B) S.dc_state[(long)pb * H + pu] = dc;       // for the launch that continues at t0 - 1
}

}  // namespace

bool chain_bwd_plan(ChainBwdDesc& d) {
    if (d.kind != CHAIN_LSTM || d.H != 1024 || d.B < 1 || d.B > 64) return false;
    if (chain_device_cus() < 256) return false;
    return true;
}

size_t chain_bwd_exchange_bytes(const ChainBwdDesc& d, size_t* x_bytes, size_t* pb_bytes) {
    const size_t MT = (d.B + 31) / 32;
    *x_bytes = (size_t)2 * (4 * d.H / 16) * MT * 1024;
    *pb_bytes = (size_t)2 * GKP * (d.H / PU) * MT * 32 * PU * sizeof(float);
    return *x_bytes + *pb_bytes;
}

int chain_bwd(const ChainBwdDesc& d, hipStream_t s) {
    T2_REQUIRE(d.t1 > d.t0 && d.t0 >= 0, "chain_bwd: bad step range [%d,%d)", d.t0, d.t1);
    T2_REQUIRE(d.X && d.PB && d.cnt && d.err, "chain_bwd: exchange buffers missing");
    T2_REQUIRE(d.kind == CHAIN_LSTM, "chain_bwd: kind %d not covered", d.kind);
    const int MT = (d.B + 31) / 32;
    const size_t smem = (size_t)(4 + NWV * MT * 32 * PPR) * sizeof(float);
    const int grid = (d.H / GNC) * GKP;
    T2_CHECK_HIP(hipMemsetAsync(d.cnt, 0, kChainBwdCntBytes, s));
    if (MT == 1) {
        T2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_bwd_lstm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(chain_bwd_lstm_kernel<1>, dim3(grid), dim3(NTH), smem, s, d);
    } else {
        T2_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_bwd_lstm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(chain_bwd_lstm_kernel<2>, dim3(grid), dim3(NTH), smem, s, d);
    }
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
