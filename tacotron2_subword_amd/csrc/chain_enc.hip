// Persistent encoder BiLSTM chains (model.py:93-112, 120-123: the packed / unpacked bidirectional nn.LSTM of an Encoder):
// every time step of BOTH directions in ONE launch, forward and backward (BPTT), exact fp32 arithmetic.
//
// Why: as launches the two encoders cost 160 (forward) + 320 (backward) dependent kernels of 4-10 us per iteration —
// the last per-step launches of the training path, shorter than the launch gaps between them (round 2: 3.4 ms of kernel
// time, ~2 ms of wall time, ~480 host launches per iteration).  H = 256 makes the recurrent weights small: W_hh of one
// direction is 1 MB of fp32, i.e. 16 registers per thread when 32-64 workgroups share it, so unlike the decoder chains
// (chain.hip, bf16 shadows) these keep the weights in fp32 and use v_mfma_f32_32x32x2_f32 (an exact fma chain): the same
// kernel serves the fp32 parity mode and the bf16 mode.
//
// Forward.  Work item (direction s, unit group ug of 8 hidden units = 32 gate columns, row group rg of 32 batch rows):
//   gates[32 x 32] = pre[t] + h_{t-1}[32 x H] . W_hh[32 gate columns][H]^T, the 8 waves split K = H, partial tiles summed
//   through LDS in fixed order, gates / cell / length mask per (row, unit), h_t published.
//   What crosses workgroups: h_t (fp32, [unit chunk of 8][row][8]: a producer writes one contiguous KB, a consumer lane
//   reads its 16 consecutive k as 64 contiguous bytes), one arrival counter per (direction, row group).  One hop per step.
// Backward (reverse of the processing order).  Two kinds of item per (direction, row group), one hop each per step:
//   P (16 units x 32 rows): dL/dh = direct gradient + the 4 K-split partials of dx(t+1); gate derivatives; dL/dc in a
//     register; dg(t) out as fp32 [gate-column chunk of 16][row][16] and as the dpre rows the weight-gradient GEMMs read;
//   G (32 output units x one gate block of K = 4H): dx(t) = dg(t) . W_hh, W_hh^T slice in registers, 8 waves split K.
// Packed sequences (lengths given): an item past its length holds zero state and writes zero outputs / gates, so the
// reverse direction starts each item at its own last frame — pack_padded_sequence semantics; the backward needs no mask
// (zero gates give zero gradients).  Hand-off protocol, bounded spins and abort reports: chain_common.h.
#include <algorithm>

#include "chain_common.h"

namespace t2 {

namespace {

using namespace chain;

constexpr int EKW = 16;                    // K elements per lane: H = 16 * EKW = 256
constexpr int EH = 16 * EKW;
constexpr int EGKP = 4;                    // backward: K parts of the dx product = the four gate blocks
constexpr int EPU = 16;                    // backward: hidden units of a P item

__device__ __forceinline__ float acc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(NTH) void enc_chain_fwd_kernel(EncChainDesc d) {
    constexpr int H = EH, KW = EKW, NUG = H / 8;
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hk = lane >> 5;
    const int B = d.B, T = d.T, NRG = (B + 31) / 32, Bp = NRG * 32;
    const int s = wg / (NUG * NRG), rem = wg % (NUG * NRG), ug = rem / NRG, rg = rem % NRG;
    const int u0 = ug * 8, row0 = rg * 32;
    const bool rev = d.reverse[s] != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* abortw = reinterpret_cast<unsigned*>(smem);
    float* partL = smem + 4;                               // [NWV][32][PPR]
    float* hsL = smem + 4 + NWV * 32 * PPR;                // [32][8]
    if (tid == 0) *abortw = 0;

    // W_hh slice -> registers (once): B operand of MFMA i = W_hh[gate column r][k(i, hk)], k(i, hk) = wave*H/8 + hk*KW + i
    float wreg[KW];
    {
        const float* wr = d.w_hh[s] + (long)((r >> 3) * H + u0 + (r & 7)) * H + wave * (H / 8) + hk * KW;
#pragma unroll
        for (int i = 0; i < KW; i += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wr + i);
            wreg[i] = v[0]; wreg[i + 1] = v[1]; wreg[i + 2] = v[2]; wreg[i + 3] = v[3];
        }
    }
    const unsigned xs_bytes = (unsigned)(H * Bp * 4);      // h of one direction and parity
    auto rsX = __builtin_amdgcn_make_buffer_rsrc(d.X, 0, (int)(2u * d.ND * xs_bytes), 0x00020000);
    unsigned* cnt = d.cnt + (size_t)(s * NRG + rg) * CNT_STRIDE;

    const int bl = (tid & 255) >> 3, uu = tid & 7;
    const int b = row0 + bl, u = u0 + uu;                  // this thread's (row, unit) when tid < 256
    const bool cell = tid < 256;
    const int len = (cell && b < B) ? (d.lengths ? d.lengths[b] : T) : 0;
    float cst = 0.f;
    float pre_next[4];
    auto load_pre = [&](int n, int tid) {
        const int tt = rev ? T - 1 - n : n;
        const int bb = min(row0 + ((tid & 255) >> 3), B - 1), un = u0 + (tid & 7);
#pragma unroll
        for (int g = 0; g < 4; ++g) pre_next[g] = d.pre[s][((long)tt * B + bb) * 4 * H + g * H + un];
    };
    if (cell) load_pre(0, tid);
    __syncthreads();

    for (int n = 0; n < T; ++n) {
        int tv = threadIdx.x;
        asm volatile("" : "+v"(tv));                       // per-step opaque thread index: addresses stay inside the step (chain.hip)
        const int tt = rev ? T - 1 - n : n;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = pre_next[g];
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (n > 0) {                                       // h of the previous processed step (zero state before the first)
            if (wave == 0 && !poll_counter(cnt, (unsigned)n, (unsigned)NUG, d.err, 21u) && lane == 0) *abortw = 1;
            __syncthreads();
            if (*abortw) return;
            // this lane's 16 consecutive k of row r: two 8-unit chunks of 32 bytes
            const unsigned base = (unsigned)(((n - 1) & 1) * d.ND + s) * xs_bytes;
            const int chunk0 = (wave * (H / 8) + hk * KW) / 8;
            u32x4 av[KW / 4];
#pragma unroll
            for (int j = 0; j < KW / 8; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    av[2 * j + q] = __builtin_amdgcn_raw_buffer_load_b128(rsX, base + (unsigned)(((chunk0 + j) * Bp + row0 + (tv & 31)) * 32 + q * 16), 0, SC1);
#pragma unroll
            for (int q = 0; q < KW / 4; ++q) {
                const f32x4 fv = __builtin_bit_cast(f32x4, av[q]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[0], wreg[4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[1], wreg[4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[2], wreg[4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[3], wreg[4 * q + 3], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) partL[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PPR + r] = acc[e];
        __syncthreads();
        float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (cell) {
            float g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) sum += partL[(w * 32 + bl) * PPR + g * 8 + uu];
                g4[g] = sum + pre[g];
            }
            float ig = acc_sigmoid(g4[0]), fg = acc_sigmoid(g4[1]), gg = tanhf(g4[2]), og = acc_sigmoid(g4[3]);
            float cn = fg * cst + ig * gg;
            float hn = og * tanhf(cn);
            if (tt >= len) { ig = fg = gg = og = 0.f; cn = 0.f; hn = 0.f; }       // past the item's length (or a padding row): zero state
            cst = cn;
            sv[0] = ig; sv[1] = fg; sv[2] = gg; sv[3] = og; sv[4] = cn; sv[5] = hn;
            hsL[bl * 8 + uu] = hn;
        }
        __syncthreads();
        if (wave == 0) {                                   // h_t of this item: 32 rows x 8 units = one contiguous KB
            const f32x4 v = *reinterpret_cast<const f32x4*>(hsL + (lane >> 1) * 8 + (lane & 1) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsX,
                (unsigned)((n & 1) * d.ND + s) * xs_bytes + (unsigned)(((ug * Bp + row0 + (lane >> 1)) * 8 + (lane & 1) * 4) * 4), 0, SC1);
        }
        publish(cnt, (unsigned)ug);
        // saved activations and the module's output: plain stores in the slack before the next poll answers
        if (cell && b < B) {
            const long rb = (long)tt * B + b;
            float* gp = d.gates[s] + rb * 4 * H + u;
            gp[0] = sv[0]; gp[H] = sv[1]; gp[2 * H] = sv[2]; gp[3 * H] = sv[3];
            d.c[s][rb * H + u] = sv[4];
            d.h[s][rb * d.ldh + u] = sv[5];
        }
        if (cell && n + 1 < T) load_pre(n + 1, tv);
    }
}

// ------------------------------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(NTH) void enc_chain_bwd_kernel(EncChainBwdDesc d) {
    constexpr int H = EH, KW = EKW, K4 = 4 * H, NPG = H / EPU, NT = H / 32, NG = NT * EGKP, IPG = NG > NPG ? NG : NPG;
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hk = lane >> 5;
    const int B = d.B, T = d.T, NRG = (B + 31) / 32, Bp = NRG * 32;
    const int sg = wg / IPG, idx = wg % IPG, s = sg / NRG, rg = sg % NRG, row0 = rg * 32;
    const bool hasP = idx < NPG, hasG = idx < NG;
    const int pg = idx, nt = idx % NT, kp = idx / NT;
    const bool rev = d.reverse[s] != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* abortw = reinterpret_cast<unsigned*>(smem);
    float* partL = smem + 4;                               // [NWV][32][PPR]
    if (tid == 0) *abortw = 0;

    // G: W_hh^T slice -> registers (once): B operand of MFMA i = W_hh[k(i, hk)][output unit nt*32 + r], k in gate block kp
    float wreg[KW];
    if (hasG) {
        const float* wr = d.w_hh[s] + (long)(kp * H + wave * (H / 8) + hk * KW) * H + nt * 32 + r;
#pragma unroll
        for (int i = 0; i < KW; ++i) wreg[i] = wr[(long)i * H];
    }
    const unsigned dg_bytes = (unsigned)(K4 * Bp * 4), pb_bytes = (unsigned)(EGKP * H * Bp * 4);     // per direction and parity
    auto rsX = __builtin_amdgcn_make_buffer_rsrc(d.X, 0, (int)(2u * d.ND * dg_bytes), 0x00020000);
    auto rsP = __builtin_amdgcn_make_buffer_rsrc(d.PB, 0, (int)(2u * d.ND * pb_bytes), 0x00020000);
    unsigned* cbase = d.cnt + (size_t)((s * NRG + rg) * (1 + NT)) * CNT_STRIDE;
    unsigned* cntP = cbase;                                // arrivals of the group's P items
    unsigned* cntG_mine = cbase + (size_t)(1 + nt) * CNT_STRIDE;
    unsigned* cntG_wait = cbase + (size_t)(1 + pg / 2) * CNT_STRIDE;          // the column tile that covers this P item's 16 units

    const int prow = tid >> 4, pu = pg * EPU + (tid & 15);
    const int pb = row0 + prow;
    float dc = 0.f;
    float pin[7];                                          // dh, i, f, g, o, c, c_prev of the step to come
    auto load_pin = [&](int m, int tid) {
        const int n = T - 1 - m, tt = rev ? T - 1 - n : n;
        const int bb = min(row0 + (tid >> 4), B - 1), un = pg * EPU + (tid & 15);
        const long rb = (long)tt * B + bb;
        pin[0] = d.dh[s][rb * d.lddh + un];
        const float* gp = d.gates[s] + rb * K4 + un;
        pin[1] = gp[0]; pin[2] = gp[H]; pin[3] = gp[2 * H]; pin[4] = gp[3 * H];
        pin[5] = d.c[s][rb * H + un];
        const int tp = rev ? tt + 1 : tt - 1;              // the time index processed before tt
        pin[6] = n > 0 ? d.c[s][((long)tp * B + bb) * H + un] : 0.f;
    };
    if (hasP) load_pin(0, tid);
    __syncthreads();

    for (int m = 0; m < T; ++m) {                          // m-th step of the BPTT = forward step n = T-1-m
        int tv = threadIdx.x;
        asm volatile("" : "+v"(tv));
        const int n = T - 1 - m, tt = rev ? T - 1 - n : n;
        // =========================================================================================== P(m)
        if (hasP) {
            float in[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) in[i] = pin[i];
            float dh = in[0];
            if (m > 0) {
                if (wave == 0 && !poll_counter(cntG_wait, (unsigned)m, (unsigned)EGKP, d.err, 22u) && lane == 0) *abortw = 1;
                __syncthreads();
                if (*abortw) return;
                const unsigned off = (unsigned)(((m - 1) & 1) * d.ND + s) * pb_bytes + (unsigned)(((pg * Bp + row0 + (tv >> 4)) * EPU + (tv & 15)) * 4);
                float pv[EGKP];
#pragma unroll
                for (int z = 0; z < EGKP; ++z) pv[z] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsP, off + (unsigned)z * (unsigned)(H * Bp * 4), 0, SC1));
                float a = 0.f;
#pragma unroll
                for (int z = 0; z < EGKP; ++z) a += pv[z];
                dh += a;
            }
            const float ig = in[1], fg = in[2], gg = in[3], og = in[4];
            const float tc = tanhf(in[5]);
            const float dcn = dc + dh * og * (1.0f - tc * tc);
            float dgv[4] = {dcn * gg * ig * (1.0f - ig), dcn * in[6] * fg * (1.0f - fg), dcn * ig * (1.0f - gg * gg), dh * tc * og * (1.0f - og)};
            if (pb >= B) { dgv[0] = dgv[1] = dgv[2] = dgv[3] = 0.f; }
            dc = dcn * fg;
            // dg(m): gate g of this item's 16 units = chunk g*H/16 + pg of [chunk][row][16]
            const unsigned xo = (unsigned)((m & 1) * d.ND + s) * dg_bytes + (unsigned)(((pg * Bp + row0 + (tv >> 4)) * 16 + (tv & 15)) * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dgv[g]), rsX, xo + (unsigned)g * (unsigned)((H / 16) * Bp * 64), 0, SC1);
            publish(cntP, (unsigned)pg);
            if (pb < B) {                                  // dpre rows (weight-gradient and input-gradient GEMMs)
                float* gp = d.dpre[s] + ((long)tt * B + pb) * K4 + pu;
                gp[0] = dgv[0]; gp[H] = dgv[1]; gp[2 * H] = dgv[2]; gp[3 * H] = dgv[3];
            }
            if (m + 1 < T) load_pin(m + 1, tv);
        }
        // =========================================================================================== G(m)
        if (hasG && m + 1 < T) {                           // (dx of the first forward step feeds nothing)
            if (wave == 0 && !poll_counter(cntP, (unsigned)(m + 1), (unsigned)NPG, d.err, 23u) && lane == 0) *abortw = 1;
            __syncthreads();
            if (*abortw) return;
            const unsigned base = (unsigned)((m & 1) * d.ND + s) * dg_bytes;
            const int chunk = (kp * H + wave * (H / 8) + hk * KW) / 16;
            u32x4 av[KW / 4];
#pragma unroll
            for (int q = 0; q < KW / 4; ++q)
                av[q] = __builtin_amdgcn_raw_buffer_load_b128(rsX, base + (unsigned)(((chunk * Bp + row0 + (tv & 31)) * 16 + q * 4) * 4), 0, SC1);
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int q = 0; q < KW / 4; ++q) {
                const f32x4 fv = __builtin_bit_cast(f32x4, av[q]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[0], wreg[4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[1], wreg[4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[2], wreg[4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[3], wreg[4 * q + 3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) partL[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PPR + r] = acc[e];
            __syncthreads();
            if (tv < 256) {                                // (row, 4 columns): fixed-order sum of the 8 waves' partials
                const int row = tv >> 3, c4 = (tv & 7) * 4;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < NWV; ++w) sum += *reinterpret_cast<const f32x4*>(partL + (w * 32 + row) * PPR + c4);
                const int ch = nt * 2 + c4 / 16;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rsP,
                    (unsigned)((m & 1) * d.ND + s) * pb_bytes + (unsigned)kp * (unsigned)(H * Bp * 4) + (unsigned)(((ch * Bp + row0 + row) * 16 + (c4 & 15)) * 4), 0, SC1);
            }
            publish(cntG_mine, (unsigned)kp);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
bool enc_chain_covers(int ND, int B, int H) {
    return ND >= 1 && ND <= 2 && H == EH && B >= 1 && B <= 128 && chain_device_cus() >= 256 && chain_device_claim();
}

// floats of exchange space: [status 64 | counters | h or (dg + partials)]
size_t enc_chain_ws_floats(int ND, int B, int H, int backward) {
    const size_t NRG = (B + 31) / 32, Bp = NRG * 32;
    const size_t cnt = (size_t)ND * NRG * (backward ? 1 + H / 32 : 1) * CNT_STRIDE;
    const size_t x = backward ? (size_t)2 * ND * (4 * H * Bp + EGKP * H * Bp) : (size_t)2 * ND * H * Bp;
    return 64 + cnt + x + 64;
}

int enc_chain_fwd(EncChainDesc d, float* ws, size_t ws_floats, hipStream_t s) {
    T2_REQUIRE(enc_chain_covers(d.ND, d.B, d.H) && d.T >= 1, "enc_chain_fwd: shape not covered (ND=%d B=%d H=%d)", d.ND, d.B, d.H);
    T2_REQUIRE(ws && ws_floats >= enc_chain_ws_floats(d.ND, d.B, d.H, 0), "enc_chain_fwd: exchange space too small");
    const int NRG = (d.B + 31) / 32;
    d.err = reinterpret_cast<unsigned*>(ws);
    d.cnt = reinterpret_cast<unsigned*>(ws + 64);
    d.X = ws + 64 + (size_t)d.ND * NRG * CNT_STRIDE;
    T2_CHECK_HIP(hipMemsetAsync(ws, 0, (64 + (size_t)d.ND * NRG * CNT_STRIDE) * sizeof(float), s));
    const int grid = d.ND * (EH / 8) * NRG;
    const size_t smem = (size_t)(4 + NWV * 32 * PPR + 32 * 8) * sizeof(float);
    T2_TRY_RC(persistent_prepare(enc_chain_fwd_kernel, grid, smem));
    hipLaunchKernelGGL(enc_chain_fwd_kernel, dim3(grid), dim3(NTH), smem, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

int enc_chain_bwd(EncChainBwdDesc d, float* ws, size_t ws_floats, hipStream_t s) {
    T2_REQUIRE(enc_chain_covers(d.ND, d.B, d.H) && d.T >= 1, "enc_chain_bwd: shape not covered (ND=%d B=%d H=%d)", d.ND, d.B, d.H);
    T2_REQUIRE(ws && ws_floats >= enc_chain_ws_floats(d.ND, d.B, d.H, 1), "enc_chain_bwd: exchange space too small");
    const int NRG = (d.B + 31) / 32, Bp = NRG * 32, NT = EH / 32;
    const size_t ncnt = (size_t)d.ND * NRG * (1 + NT) * CNT_STRIDE;
    d.err = reinterpret_cast<unsigned*>(ws);
    d.cnt = reinterpret_cast<unsigned*>(ws + 64);
    d.X = ws + 64 + ncnt;
    d.PB = d.X + (size_t)2 * d.ND * 4 * EH * Bp;
    T2_CHECK_HIP(hipMemsetAsync(ws, 0, (64 + ncnt) * sizeof(float), s));
    const int ipg = std::max(NT * EGKP, EH / EPU);
    const int grid = d.ND * NRG * ipg;
    const size_t smem = (size_t)(4 + NWV * 32 * PPR) * sizeof(float);
    T2_TRY_RC(persistent_prepare(enc_chain_bwd_kernel, grid, smem));
    hipLaunchKernelGGL(enc_chain_bwd_kernel, dim3(grid), dim3(NTH), smem, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
