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
// Hand-offs: no arrival counters (round 3) — dg fragments and K-split partial words carry step tags (chain_common.h: step_tag,
// tag_f32), the exchange buffers are cleared per launch, consumers load until the tags are this step's.  Two all-to-all hops
// per step instead of two launches; numerics = the launch path (lstm.hip) up to summation order.
#include <algorithm>

#include "chain_common.h"

#ifdef T2_STAMPS
__device__ unsigned long long t2_chain_bwd_stamps[256 * 16];
extern "C" int t2_debug_read_chain_bwd_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(t2_chain_bwd_stamps), sizeof(unsigned long long) * n);
}
extern "C" int t2_debug_clear_chain_bwd_stamps(void) {
    static unsigned long long z[256 * 16];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(t2_chain_bwd_stamps), z, sizeof(z));
}
#define T2_BSTAMP(i)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); t2_stamp_lds[i] += n_ - stamp_last; stamp_last = n_; } \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define T2_BSTAMP(i)
#endif

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
    // (no arrival counters: dg fragments and K-split partial words carry step tags, consumers load until the tags are this step's)

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
    float bacc[4] = {0.f, 0.f, 0.f, 0.f};                           // this thread's (row, unit): sum over the steps of dg = bias gradient partial
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
                // (no counter poll: the partial words carry G(t+1)'s tag and are loaded until they show it; a resumed launch
                //  finds step t1's partials complete)
                // dx(t+1)[row, unit] = fixed-order sum of the 8 K-split partials (written by G(t+1) into parity (t+1)&1)
                const unsigned off = (unsigned)(((t + 1) & 1)) * pb_half + (unsigned)((((ug * MT + rt) * 32 + (tv >> 4)) * PU + (tv & 15)) * 4);
                const unsigned kstride = (unsigned)((H / PU) * MT * 32 * PU * 4);
                float pv[GKP];
                const unsigned want = step_tag(ep - 1);               // tag of G(t+1)'s partials (resumed launches: step t1's are drained)
                const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    unsigned okw = 1u;
#pragma unroll
                    for (int z = 0; z < GKP; ++z) {
                        const unsigned wv = __builtin_amdgcn_raw_buffer_load_b32(rsP, off + z * kstride, 0, SC1);
                        pv[z] = __builtin_bit_cast(float, wv);
                        okw &= ((wv & 1u) == want) ? 1u : 0u;
                    }
                    if (ep == 0 || __all(okw != 0u)) break;
                    if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if ((tv & 63) == 0) { report_abort(d.err, 15u); *abortw = 1; } break; }
                }
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
#pragma unroll
            for (int g = 0; g < 4; ++g) bacc[g] += dgv[g];
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
                u32x4 ow = __builtin_bit_cast(u32x4, o);
                ow.x = (ow.x & ~1u) | step_tag(ep);                   // the unit's tag: lowest mantissa bit of its first element
                __builtin_amdgcn_raw_buffer_store_b128(ow, rsX,
                    (unsigned)(t & 1) * (unsigned)(KT * MT * 1024) + (unsigned)((((wave * H + u0) / 16) * MT + rt) * 1024 + lane * 16), 0, SC1);
            }
            __syncthreads();                                          // (dgL is read; G's partial tiles alias it)
            if (*abortw) return;
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
            f32x16 acc[MT];
            const unsigned xb = (unsigned)(t & 1) * (unsigned)(KT * MT * 1024) + (unsigned)lane * 16u;
            u32x4 af[MT][4];
            {
                const unsigned want = step_tag(ep);
                const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    unsigned okw = 1u;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            af[m][i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xb + (unsigned)(((kp * KPW / 16 + wave * KSW + min(i, KSW - 1)) * MT + m) * 1024), 0, SC1);
                            okw &= ((af[m][i].x & 1u) == want) ? 1u : 0u;
                        }
                    if (__all(okw != 0u)) break;
                    if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if (lane == 0) { report_abort(d.err, 16u); *abortw = 1; } break; }
                }
            }
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
                const unsigned tg = step_tag(ep);                     // every word carries the tag (the consumer reads single words)
                sum[0] = tag_f32(sum[0], tg); sum[1] = tag_f32(sum[1], tg); sum[2] = tag_f32(sum[2], tg); sum[3] = tag_f32(sum[3], tg);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rsP,
                    (unsigned)(t & 1) * pb_half + (unsigned)((((kp * (H / PU) + ugo) * MT + m) * 32 + row) * PU + (c4 % PU)) * 4u, 0, SC1);
            }
            __syncthreads();
            if (*abortw) return;
        }
    }
    if (hasP && d.t0 > 0 && pb < B) S.dc_state[(long)pb * H + pu] = dc;       // for the launch that continues at t0 - 1
    if (S.dbias_part) {                                               // bias gradients: fixed-order sum over the row tile's 32 rows
        __syncthreads();
        if (hasP) {
#pragma unroll
            for (int g = 0; g < 4; ++g) dgL[(g * 32 + (tid >> 4)) * (PU + 4) + (tid & 15)] = bacc[g];
        }
        __syncthreads();
        if (hasP && tid < 4 * PU) {
            const int g = tid / PU, u = tid % PU;
            float sum = 0.f;
            for (int row = 0; row < 32; ++row) sum += dgL[(g * 32 + row) * (PU + 4) + u];
            float* o = S.dbias_part + (long)rt * K4 + g * H + u0 + u;
            *o = resumed ? *o + sum : sum;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Attention chain (autograd of model.py:322-369 with StepwiseMonotonicAttention, attention.py:291-398), reverse time.
// Three phases per step, each workgroup holding one item of each kind:
//   A item (b, s, split): half of the memory positions of one batch item and stream — attention.hip's
//       attention_step_bwd_kernel with its operands resident: the item's processed-memory rows (fp32), its encoder-memory
//       rows (bf16) and the d(processed memory) / dv accumulators stay in LDS for all steps, the carry of the recurrence
//       too; only the boundary carry crosses to the other split (one float per step, write-through).
//       total ctx gradient = two direct sources + 4 K-split partials of dx_ctx(t+1); g_j = dctx . memory_j (+ carry);
//       recurrence and energies backward; dq partial -> P items.
//   P item (s, ug, rt): 16 units x 32 rows of an attention LSTM: dh = direct + 4 K-split partials of dx_h(t+1) + dq . Wq
//       (Wq slice resident in LDS), gate derivatives, dL/dc in a register, dg(t) published as bf16 fragments.
//   G item (s, nt, kp): dx(t) = dg(t) . [W_ih[:,P:] | W_hh]: 64 of the 1536 output columns x one gate block (K = 1024) of
//       the transposed shadow resident in registers (64 VGPRs); the 8 waves split K, partial tiles summed through LDS.
// Hops per step: G(t+1) -> A(t) -> P(t) -> G(t).
//
// KIND = CHAIN_LSA (LocationSensitiveAttention, attention.py:26-85; the launch path's attention_lsa_step_bwd_mfma_kernel):
// same items, P and G shared.  The A item keeps, besides the above (no processed memory: the tanh tile u = tanh(q + pm +
// location term) and the location features of every step come from the forward chain's saved copies, layout.usave /
// locsave): the conv weights (zero-padded taps), dloc of the last step with `pad` halo rows on both sides, [w; cum] of two
// steps, the carried gradient on the cumulative weights, and the d(Wd) / d(Wc) tiles as MFMA accumulators in registers.
// Two quantities cross the position splits: the `pad` boundary rows of dloc (stored before the publish of step t+1, read at
// the start of step t; tagged 16-byte units, no counter poll) and the softmax dot S = sum_j w_j g_j (one tagged 8-byte
// write-through store per split and step); both areas are cleared per launch.  Everything that does not need step t's
// context gradient runs in front of the poll for it (halo rows, carried gradients, d(Wc) of step t+1, the saved tile's loads).
// ---------------------------------------------------------------------------------------------------------------
constexpr int ANC = 64;          // output columns of a G item (attention chain)
constexpr int AKP = 4;           // K parts = the four gate blocks

struct BwdLds { int ab, v, q, dctx, g, de, ps, ap, carry, dva, dqo, wq, pm, dpm, mem, cc, convw, dloc, wpad, ut, loc, red, TwP, scratch, total; };
__host__ __device__ inline BwdLds bwd_lds_of(int A, int E, int Tc, int MT, int kind = CHAIN_SMA, int F = 0, int Kc = 1) {
    BwdLds m; int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    const bool lsa = kind == CHAIN_LSA;
    const int pad = (Kc - 1) / 2;
    m.ab = take(4); m.v = take(A); m.q = take(A); m.dctx = take(E);
    m.g = take(Tc + 8); m.de = take(Tc + 8); m.ps = take(Tc + 8); m.ap = take(Tc + 8); m.carry = take(Tc + 8);
    m.dva = take(A); m.dqo = take(A);
    m.wq = take(PU * (A + 8));                                 // Wq slice of the P item as a bf16 hi/lo pair [2][PU][A + 8]
    m.pm = take(lsa ? 0 : Tc * A); m.dpm = take(Tc * A); m.mem = take((Tc + 1) * E / 2);
    // LSA: carried gradient on the cumulative weights, conv weights, dloc of the last step with `pad` halo rows on both
    // sides (pitch F + 1, column F stays zero), [w_prev; cum_prev] of two steps (ping-pong) with the same halos
    m.TwP = (Tc + 2 * pad + 8 + 3) & ~3;                    // (+8: the conv product's K is padded to a multiple of 16 with zero weights)
    m.cc = take(lsa ? Tc + 8 : 0); m.convw = take(lsa ? F * ((2 * Kc + 15) & ~15) : 0); m.dloc = take(lsa ? (Tc + 2 * pad) * (F + 1) : 0);
    m.wpad = take(lsa ? 4 * m.TwP : 0);
    // scratch shared by the phases; LSA's A phase: tanh / dpre tile [Tc][A + 4], location features [Tc][F + 1], reduction rows
    // (the tile's room first holds Q = dloc . Wc^T, [Tc + 2 pad][65], of the carried-gradient step)
    const int nut = Tc * (A + 4), nq = (Tc + 2 * pad) * 65;
    m.ut = 0; m.loc = ((nut > nq ? nut : nq) + 3) & ~3; m.red = m.loc + ((Tc * (F + 1) + 3) & ~3);
    const int sg = NWV * 32 * PPR, sp = 32 * (A + 4) + 4 * 32 * (PU + 4) + 4 * 32 * (PU + 1), sa = lsa ? m.red + 2 * NWV * A : 2 * 32 * A;
    m.scratch = take(sg > sp ? (sg > sa ? sg : sa) : (sp > sa ? sp : sa));
    m.total = o;
    return m;
}

template <int MT, int KIND>
__global__ __launch_bounds__(NTH) void chain_bwd_att_kernel(ChainBwdDesc d) {
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hk = lane >> 5;
    const int B = d.B, H = d.H, K4 = 4 * H, E = d.E, A = d.A, N = E + H;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // ---------------------------------------------------------------- items
    const int NTT = N / ANC, NTC = E / ANC;                   // column tiles per stream (24), of which context tiles (8)
    const int nGs = NTT * AKP, nPs = (H / PU) * MT, nAs = B * 2;
    const bool hasG = wg < d.NS * nGs, hasP = wg < d.NS * nPs, hasA = wg < d.NS * nAs;
    const int gs = hasG ? wg / nGs : 0, nt = (wg % nGs) % NTT, kp = (wg % nGs) / NTT;
    const int ps = hasP ? wg / nPs : 0, ug = (wg % nPs) / MT, rt = (wg % nPs) % MT, u0 = ug * PU;
    const int as = hasA ? wg / nAs : 0, ab_ = (wg % nAs) / 2, split = wg % 2;
    const ChainBwdStream& GS = d.st[gs];
    const ChainBwdStream& PS = d.st[ps];
    const ChainBwdStream& AS = d.st[as];
    const int Tin = AS.Tin;
    const int chunk = (((Tin + 1) / 2) + 3) & ~3;
    const int jb = min(split * chunk, Tin), je = min(jb + chunk, Tin), len = je - jb;
    const int ng = (KIND == CHAIN_SMA && je < Tin) ? len + 1 : len;   // g values computed here: positions [jb, jb + ng)
    const BwdLds M = bwd_lds_of(A, E, d.lds_Tc, MT, KIND, d.F, d.Kc);
    unsigned* abortw = reinterpret_cast<unsigned*>(smem + M.ab);
    float* vL = smem + M.v; float* qL = smem + M.q; float* dctxL = smem + M.dctx; float* gL = smem + M.g; float* deL = smem + M.de;
    float* psL = smem + M.ps; float* apL = smem + M.ap; float* carryL = smem + M.carry; float* dvaL = smem + M.dva; float* dqoL = smem + M.dqo;
    __bf16* wq16L = reinterpret_cast<__bf16*>(smem + M.wq); float* pmL = smem + M.pm; float* dpmL = smem + M.dpm;
    __bf16* memL = reinterpret_cast<__bf16*>(smem + M.mem);
    float* partL = smem + M.scratch;                           // G: [NWV][32][PPR]
    float* dqL = smem + M.scratch;                             // P: [32][A+4]
    float* dgL = smem + M.scratch + 32 * (A + 4);              // P: [4][32][PU+4]
    float* dpL = dgL + 4 * 32 * (PU + 4);                      // P: [4 K quarters][32][PU+1] partial dq . Wq
    float* redL = smem + M.scratch;                            // A: [2][32][A]

    const int KT = K4 / 16;
    const unsigned xs = (unsigned)(KT * MT * 1024);            // dg fragments of one stream
    auto rsX = __builtin_amdgcn_make_buffer_rsrc(d.X, 0, (int)(2u * d.NS * xs), 0x00020000);
    auto rsH = __builtin_amdgcn_make_buffer_rsrc(d.PB, 0, (int)d.pb_bytes, 0x00020000);
    auto rsC = __builtin_amdgcn_make_buffer_rsrc(d.PBC, 0, (int)d.pbc_bytes, 0x00020000);
    auto rsQ = __builtin_amdgcn_make_buffer_rsrc(d.DQX, 0, d.NS * 2 * B * A * 4, 0x00020000);
    // SMA: one boundary-carry float per (parity, stream, item).  LSA: [parity][stream][item][split][pad][F] halo rows of dloc, then
    // [parity][stream][item][split] (softmax-dot partial, step tag)
    auto rsK = __builtin_amdgcn_make_buffer_rsrc(d.CARRYX, 0, KIND == CHAIN_LSA ? (((d.Kc - 1) / 2) * d.F + 2) * 2 * d.NS * B * 2 * 4 : 2 * d.NS * B * 4, 0x00020000);
    const unsigned pbh_half = d.pb_bytes / 2, pbc_half = d.pbc_bytes / 2;
    const unsigned pbh_kp = (unsigned)((H / PU) * MT * 32 * PU * 4), pbh_s = pbh_kp * AKP;     // bytes per K part / per stream
    const unsigned pbc_kp = (unsigned)(B * E * 4), pbc_s = pbc_kp * AKP;
    // Hand-offs of this kernel carry NO counters: every payload is tagged — the lowest bit of every fp32 word of the K-split
    // partials, of the first word of every 16-byte unit elsewhere — with a bit of the step count (buffers with two parities: bit 1,
    // inverted = step_tag; the single dq buffer: bit 0), the exchange buffers are cleared per launch, and a consumer loads its
    // operands and loads them again until every tag is this step's.  One round trip per hop instead of two (counter poll, then
    // the loads), no drain and no atomic on the producer's side.  Every spin is bounded (SPIN_TICKS) like the polls it replaces.
    auto load_tagged4 = [&](auto rs, unsigned off, unsigned stride, unsigned want, float (&pv)[AKP], unsigned code) {
        const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            unsigned okw = 1u;
#pragma unroll
            for (int z = 0; z < AKP; ++z) {
                const unsigned wv = __builtin_amdgcn_raw_buffer_load_b32(rs, off + z * stride, 0, SC1);
                pv[z] = __builtin_bit_cast(float, wv);
                okw &= ((wv & 1u) == want) ? 1u : 0u;
            }
            if (__all(okw != 0u)) break;
            if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if ((threadIdx.x & 63) == 0) { report_abort(d.err, code); *abortw = 1; } break; }
            __builtin_amdgcn_s_sleep(2);
        }
    };

    if (tid == 0) *abortw = 0;

    // ---------------------------------------------------------------- G setup: W^T slice -> registers
    // wave w: column tile c2 = w & 1 of the item's two (columns nt*64 + c2*32 + r), K quarter kq = w >> 1 of the gate block
    // (k in [kp*1024 + kq*256, +256): 16 k-steps).  Four waves share a column tile, so the partial tiles of BOTH column tiles
    // meet in LDS in one round per row tile (K split over all eight waves took a round per (row tile, column tile): 4.9 us of G
    // per step were mostly those rounds)
    bf16x8 W[16];
    if (hasG) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            W[i] = KIND == CHAIN_LSA       // (LSA: column tile i / 8, k-steps wave*8 + i % 8 — its G phase splits K over all eight waves)
                       ? *reinterpret_cast<const bf16x8*>(GS.wt16 + (long)(nt * ANC + (i >> 3) * 32 + r) * GS.ldwt + kp * H + (wave * 8 + (i & 7)) * 16 + 8 * hk)
                       : *reinterpret_cast<const bf16x8*>(GS.wt16 + (long)(nt * ANC + (wave & 1) * 32 + r) * GS.ldwt + kp * H + ((wave >> 1) * 16 + i) * 16 + 8 * hk);
    }
    // ---------------------------------------------------------------- P setup
    const int pb = rt * 32 + (tid >> 4), pu = u0 + (tid & 15);
    float dc = 0.f;
    float bacc[4] = {0.f, 0.f, 0.f, 0.f};                           // P: sum over the steps of this thread's dg = bias gradient partial
    float pin[7];
    auto load_pin = [&](int t, int tid) {
        const int b = min(rt * 32 + (tid >> 4), B - 1), u = u0 + (tid & 15);
        const long rb = (long)t * B + b;
        pin[0] = PS.dh1[rb * PS.lddh1 + u];
        const float* gp = PS.gates + rb * K4 + u;
        pin[1] = gp[0]; pin[2] = gp[H]; pin[3] = gp[2 * H]; pin[4] = gp[3 * H];
        pin[5] = PS.c_new[rb * H + u];
        pin[6] = t > 0 ? PS.c_out[((long)(t - 1) * B + b) * H + u] : 0.f;
    };
    if (hasP) {
        for (int i = tid; i < A * PU; i += NTH) {                   // [hi|lo][unit][a]: x = hi + lo, both bf16
            const float x = PS.wq[(long)(i / PU) * H + u0 + i % PU];
            const __bf16 hi = (__bf16)x;
            wq16L[(i % PU) * (A + 8) + i / PU] = hi; wq16L[(PU + i % PU) * (A + 8) + i / PU] = (__bf16)(x - (float)hi);
        }
        load_pin(d.t1 - 1, tid);
    }
    // ---------------------------------------------------------------- A setup: resident rows, zeroed accumulators
    // LSA (attention.py:26-85) geometry and buffers
    const int F = d.F, Kc = d.Kc, F1 = F + 1, pad = (Kc - 1) / 2, TwP = M.TwP, UP = A + 4, WN = len + 2 * pad;
    const int CP = (2 * Kc + 15) & ~15;                              // pitch of the conv weights in LDS: [F][CP], columns >= 2 Kc zero
    float* wL = smem + M.ps;                                         // softmax weights of the step (own positions)
    float* ccL = smem + M.cc; float* convwL = smem + M.convw; float* dlocL = smem + M.dloc; float* wpadL = smem + M.wpad;
    float* utL = smem + M.scratch + M.ut; float* locL = smem + M.scratch + M.loc; float* lredL = smem + M.scratch + M.red;
    (void)F1; (void)TwP; (void)UP; (void)WN; (void)CP; (void)wL; (void)ccL; (void)convwL; (void)dlocL; (void)wpadL; (void)utL; (void)locL; (void)lredL;
    // one bf16 MFMA operand (8 consecutive K values of this lane) gathered from fp32 LDS values
    auto pack8 = [&](auto f) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)f(i);
        return o;
    };
    // n chained fp32 MFMAs whose operands come from LDS: requested 8 pairs at a time, so that a chunk's reads are in flight
    // while the previous chunk's MFMAs issue (a plain loop pays the LDS round trip in front of every MFMA)
    auto mfma_chain = [&](int n, auto fa, auto fb, f32x16 acc) {
        f32x16 acc2;                                                  // two interleaved chains: a dependent fp32 MFMA waits out the previous one
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
        for (int i0 = 0; i0 < n; i0 += 8) {
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = fa(i0 + u); bv[u] = fb(i0 + u); }   // (n % 8 == 0: no per-element condition — a load under a
                                                                                       //  run-time condition becomes a branch or a wait per element)
            __builtin_amdgcn_sched_barrier(0);                        // (at the register cap hipcc sinks each read to its MFMA otherwise)
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u + 1], bv[u + 1], acc2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
        return acc;
    };
    f32x16 accWd, accWc;                                             // LSA: d(Wd) tile of waves 2..5, d(Wc) tile of waves 2..3, summed over all steps
#pragma unroll
    for (int e = 0; e < 16; ++e) { accWd[e] = 0.f; accWc[e] = 0.f; }
    float ain[6];                                                   // per thread: q_a | p_j (LSA: w_j), a_prev_j (LSA: [w; cum](t-1) with halos), dalign_j | two direct dctx sources
    auto load_ain = [&](int t, int tid) {
        const long rb = (long)t * B + ab_;
        ain[0] = (KIND != CHAIN_LSA && tid < A) ? AS.qs[rb * A + tid] : 0.f;      // (LSA reads the saved tanh tile instead)
        const int j = jb + tid;
        const bool in = tid < ng;
        if constexpr (KIND == CHAIN_LSA) {
            ain[1] = tid < len ? AS.align[((long)ab_ * d.T + t) * Tin + j] : 0.f;
            // [w_{t-1}; cum_{t-1}] at positions jb - pad .. je + pad - 1 (zero outside the memory and at t = 0)
            const int c = tid / WN, x = tid - c * WN, jw = jb - pad + x;
            ain[2] = (tid < 2 * WN && t > 0 && jw >= 0 && jw < Tin) ? (c == 0 ? AS.align : AS.wcum)[((long)ab_ * d.T + t - 1) * Tin + jw] : 0.f;
        } else {
            ain[1] = (in && tid < len) ? AS.psel[((long)ab_ * d.T + t) * Tin + j] : 0.f;
            ain[2] = (in && tid < len) ? (t > 0 ? AS.align[((long)ab_ * d.T + t - 1) * Tin + j] : (j == 0 ? 1.f : 0.f)) : 0.f;
        }
        ain[3] = (in && AS.dalign) ? AS.dalign[((long)ab_ * d.T + t) * Tin + j] : 0.f;
        ain[4] = tid < E ? AS.dctx_a[rb * AS.lddctx_a + tid] : 0.f;
        ain[5] = tid < E ? AS.dctx_b[rb * AS.lddctx_b + tid] : 0.f;
    };
    if (hasA) {
        for (int a = tid; a < A; a += NTH) { vL[a] = AS.v[a]; dvaL[a] = 0.f; }
        for (int i = tid; i < len * (A / 4); i += NTH) {
            const int jl = i / (A / 4), a4 = (i % (A / 4)) * 4;
            if constexpr (KIND == CHAIN_SMA)
                *reinterpret_cast<f32x4*>(pmL + jl * A + a4) = *reinterpret_cast<const f32x4*>(AS.pm + ((long)ab_ * Tin + jb + jl) * A + a4);
            *reinterpret_cast<f32x4*>(dpmL + jl * A + a4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int i = tid; i < ng * (E / 4); i += NTH) {
            const int jl = i / (E / 4), c4 = (i % (E / 4)) * 4;
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(AS.memory + ((long)ab_ * Tin + jb + jl) * E + c4);
            bf16x4 o; o[0] = (__bf16)m4[0]; o[1] = (__bf16)m4[1]; o[2] = (__bf16)m4[2]; o[3] = (__bf16)m4[3];
            *reinterpret_cast<bf16x4*>(memL + jl * E + c4) = o;
        }
        for (int j = tid; j < chunk + 8; j += NTH) carryL[j] = 0.f;
        if constexpr (KIND == CHAIN_LSA) {
            for (int j = tid; j < chunk + 8; j += NTH) ccL[j] = 0.f;
            for (int i = tid; i < F * CP; i += NTH) { const int f = i / CP, ck = i - f * CP; convwL[i] = ck < 2 * Kc ? AS.loc_conv[f * 2 * Kc + ck] : 0.f; }
            for (int i = tid; i < (d.lds_Tc + 2 * pad) * F1; i += NTH) dlocL[i] = 0.f;
            for (int i = tid; i < 4 * TwP; i += NTH) wpadL[i] = 0.f;
        }
        load_ain(d.t1 - 1, tid);
    }
    __syncthreads();
    const RngKey kh = rng_key(d.seed, PS.site_h), kc = rng_key(d.seed, PS.site_c);
    const float dscale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;

#ifdef T2_STAMPS
    // (accumulators in LDS, 128 bytes in front of the dynamic carve: 32 more registers per thread made the stamped kernels spill and
    //  the stamps measure the spills)
    __shared__ __attribute__((aligned(16))) unsigned long long t2_stamp_lds[16];
    if (threadIdx.x < 16) t2_stamp_lds[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long stamp_last = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = d.t1 - 1; t >= d.t0; --t) {
        const unsigned ep = (unsigned)(d.t1 - 1 - t);
        int tv = threadIdx.x;
        asm volatile("" : "+v"(tv));
        // ======================================================================================= A(t)
        if (hasA) {
            if constexpr (KIND == CHAIN_SMA) {
            const int tid = tv, lane = tid & 63, wave = tid >> 6;
            T2_BSTAMP(15);
            float in[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) in[i] = ain[i];
            T2_BSTAMP(0);
            // total gradient on ctx(t): direct sources + the four K-split partials of dx_ctx(t+1) (tagged words: no counter poll)
            float boundary = 0.f;                                       // carry of position je, owned by the other split
            if (tid < E) {
                float v = in[4] + in[5];
                if (ep > 0) {
                    const unsigned off = (unsigned)((t + 1) & 1) * pbc_half + (unsigned)as * pbc_s + (unsigned)((ab_ * E + tid) * 4);
                    float pv[AKP];
                    load_tagged4(rsC, off, pbc_kp, step_tag(ep - 1), pv, 7u);       // (G(t+1)'s tag on every word)
                    v += (pv[0] + pv[1]) + (pv[2] + pv[3]);
                }
                dctxL[tid] = v;
                if (split == 0) AS.dctx_out[((long)t * B + ab_) * E + tid] = v;
            }
            if (tid == 0 && ep > 0 && je < Tin) {
                const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const unsigned wv = __builtin_amdgcn_raw_buffer_load_b32(rsK, (unsigned)((((t + 1) & 1) * d.NS + as) * B + ab_) * 4u, 0, SC1);
                    boundary = __builtin_bit_cast(float, wv);
                    if ((wv & 1u) == step_tag(ep - 1)) break;
                    if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { report_abort(d.err, 14u); *abortw = 1; break; }
                }
            }
            if (tid < A) qL[tid] = in[0];
            if (tid < len) { psL[tid] = in[1]; apL[tid] = in[2]; }
            if (tid == 0) { gL[ng] = 0.f; if (je < Tin) carryL[len] = boundary; }
            __syncthreads();
            if (*abortw) return;
            T2_BSTAMP(1);
            // g_j = dctx . memory_j + dalign_j + carry_j: one wave per position, lanes stride the E columns 8 at a time
            for (int jl = wave; jl < ng; jl += NWV) {
                float sum = 0.f;
                for (int c = lane * 8; c < E; c += 512) {
                    const bf16x8 mb = *reinterpret_cast<const bf16x8*>(memL + jl * E + c);
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(dctxL + c), d1 = *reinterpret_cast<const f32x4*>(dctxL + c + 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) sum += (float)mb[k] * d0[k] + (float)mb[4 + k] * d1[k];
                }
                sum = wave_sum(sum);
                if (lane == 0) {
                    float gsum = sum;
                    gL[jl] = gsum;                                      // (+ dalign / carry added by the owning thread below)
                }
            }
            __syncthreads();
            if (tid < ng) {
                float gsum = gL[tid] + in[3];
                if (ep > 0) gsum += carryL[tid];
                gL[tid] = gsum;
            }
            __syncthreads();
            if (tid < len) {
                const float p = psL[tid], gj = gL[tid], gn = gL[tid + 1];
                deL[tid] = apL[tid] * (gj - gn) * p * (1.0f - p);
                const float co = gj * p + gn * (1.0f - p);
                carryL[tid] = co;                                       // gradient on a_{t-1}[j], consumed at step t-1
                if (tid == 0 && split == 1)                             // position jb of this split = je of the other one
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tag_f32(co, step_tag(ep))), rsK, (unsigned)(((t & 1) * d.NS + as) * B + ab_) * 4u, 0, SC1);
            }
            __syncthreads();
            T2_BSTAMP(2);
            // energies backward: 16 lanes per position, each lane owns channels sub*4 + 64*k
            {
                const int gid = tid >> 4, sub = tid & 15;
                f32x4 dq[2], dv[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) { dq[k] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                for (int jl = gid; jl < len; jl += NTH / 16) {
                    const float dej = deL[jl];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int a = sub * 4 + 64 * k;
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(pmL + jl * A + a);
                        f32x4 acc = *reinterpret_cast<const f32x4*>(dpmL + jl * A + a);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float u = fast_tanh(qL[a + c] + pv[c]);
                            const float dpre = dej * vL[a + c] * (1.0f - u * u);
                            dq[k][c] += dpre; dv[k][c] += dej * u; acc[c] += dpre;
                        }
                        *reinterpret_cast<f32x4*>(dpmL + jl * A + a) = acc;
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {                       // the wave's 4 position groups (lanes 16 apart), fixed order
                        dq[k][c] += __shfl_xor(dq[k][c], 16, 64); dq[k][c] += __shfl_xor(dq[k][c], 32, 64);
                        dv[k][c] += __shfl_xor(dv[k][c], 16, 64); dv[k][c] += __shfl_xor(dv[k][c], 32, 64);
                    }
                if (lane < 16) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        *reinterpret_cast<f32x4*>(redL + wave * A + sub * 4 + 64 * k) = dq[k];
                        *reinterpret_cast<f32x4*>(redL + (NWV + wave) * A + sub * 4 + 64 * k) = dv[k];
                    }
                }
            }
            __syncthreads();
            if (tid < A) {
                float sq = 0.f, sv = 0.f;
#pragma unroll
                for (int k = 0; k < NWV; ++k) { sq += redL[k * A + tid]; sv += redL[(NWV + k) * A + tid]; }
                dqoL[tid] = sq;
                dvaL[tid] += sv;
            }
            __syncthreads();
            if (tid < A / 4) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(dqoL + tid * 4);
                u32x4 qw = __builtin_bit_cast(u32x4, q4);
                qw.x = (qw.x & ~1u) | ((ep & 1u) ^ 1u);                  // (the dq buffer has one parity: its tag flips every step)
                __builtin_amdgcn_raw_buffer_store_b128(qw, rsQ, (unsigned)((((as * 2 + split) * B + ab_) * A + tid * 4) * 4), 0, SC1);
            }
            T2_BSTAMP(3);
            __syncthreads();                                             // (no arrival counter: the dq units carry the step's tag)
            T2_BSTAMP(4);
            if (tid < A) AS.dq_out[((long)t * B + ab_) * 2 * A + split * A + tid] = dqoL[tid];
            if (t > d.t0) load_ain(t - 1, tid);
                    } else {
            const int tid = tv, lane = tid & 63, wave = tid >> 6, r = lane & 31, hk = lane >> 5;
            T2_BSTAMP(15);
            float in[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) in[i] = ain[i];
            float* wnew = wpadL + (t & 1) * 2 * TwP;                     // [w_{t-1}; cum_{t-1}] of THIS step, halos included
            const float* wold = wpadL + ((t + 1) & 1) * 2 * TwP;         // ... of step t+1 (whose dloc is in dlocL)
            const int njt = (len + 31) / 32;
            // ------------------------------------------------------------------------------------------------------------
            // In front of the poll (none of it needs the context gradient of this step): halo rows of dloc(t+1) from the
            // other split, gradient carried to w_t / cum_t, d(Wc) of step t+1, location features and tanh tile of step t.
            // ------------------------------------------------------------------------------------------------------------
            if (ep > 0 && tid < pad * (F / 4) && ((split == 0 && je < Tin) || (split == 1 && jb > 0))) {
                // the partner's `pad` rows next to the boundary.  No poll of its arrival counter in front (a second round trip):
                // the first word of every 16-byte unit carries the step tag of the step that wrote it (the area is cleared per
                // launch), and the rows were stored a whole P and G phase ago — the first load finds them
                const int row = tid / (F / 4), f4 = (tid % (F / 4)) * 4;
                const unsigned off = (unsigned)((((((t + 1) & 1) * d.NS + as) * B + ab_) * 2 + (1 - split)) * pad * F + row * F + f4) * 4u;
                u32x4 hw;
                const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    hw = __builtin_amdgcn_raw_buffer_load_b128(rsK, off, 0, SC1);
                    if ((hw.x & 1u) == step_tag(ep - 1)) break;
                    if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { report_abort(d.err, 11u); *abortw = 1; break; }
                }
                const f32x4 h4 = __builtin_bit_cast(f32x4, hw);
                float* dst = dlocL + (split == 0 ? len + pad + row : row) * F1 + f4;
                dst[0] = h4[0]; dst[1] = h4[1]; dst[2] = h4[2]; dst[3] = h4[3];
            }
            // the tanh tile u = tanh(q + pm + location term) and the location features of step t, own positions: saved by the
            // forward chain (chain.hip), requested here — behind the halo rows, vector loads complete in order — and in
            // flight underneath the carried-gradient products (cold HBM lines: ~2 us)
            // Tile layout in memory: [A][TinP], positions contiguous (what the forward's MFMA tiles store coalesced).  A wave
            // request covers 16 channel rows x 4 consecutive 16-byte pieces; the LDS tile is [position][A + 4], so the four
            // transposing writes of a lane land 2-way conflicted at worst.
            f32x4 ureg[4], lreg;
            const int TinP = (Tin + 3) & ~3, nj4 = (len + 3) >> 2;
            {
                const float* us = AS.usave + ((long)t * B + ab_) * A * TinP + jb;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int w = wave + NWV * k, a = (w & 7) * 16 + (lane & 15), j4 = min((w >> 3) * 4 + (lane >> 4), nj4 - 1);
                    ureg[k] = *reinterpret_cast<const f32x4*>(us + (long)a * TinP + 4 * j4);
                }
                const int i = min(tid, len * (F / 4) - 1);
                lreg = *reinterpret_cast<const f32x4*>(AS.locsave + (((long)t * B + ab_) * Tin + jb + i / (F / 4)) * F + (i % (F / 4)) * 4);
            }
            if (tid < len) wL[tid] = in[1];
            if (tid < 2 * WN) { const int c = tid / WN; wnew[c * TwP + (tid - c * WN)] = in[2]; }
            __syncthreads();
            T2_BSTAMP(11);
            if (ep > 0) {
                // gradient on [w_t; cum_t] through the location conv of step t+1, dwcat[c][i] = sum_{f,k} Wc[f][c][k] dloc[i - k + pad][f],
                // in two steps: Q[p][(c,k)] = sum_f dloc[p][f] Wc[f][(c,k)] on the matrix cores (rows p = own positions and both
                // halos: 3 x 2 tiles, one per wave 0,1,4..7), then dwcat[c][i] = the anti-diagonal sum_k Q[i + 2 pad - k][(c,k)].
                // bf16 operands, as every gradient product of this mode.  Waves 2,3: a d(Wc) column tile each.
                const int NP = len + 2 * pad, qi = wave < 2 ? wave : wave - 2, pt = qi >> 1, nq = qi & 1;
                if ((wave < 2 || wave >= 4) && pt * 32 < NP && nq * 32 < 2 * Kc) {
                    const float* ar = dlocL + min(pt * 32 + r, NP - 1) * F1 + 8 * hk;
                    const float* br = convwL + 8 * hk * CP + min(nq * 32 + r, 2 * Kc - 1);
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)                       // (F = 32)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8([&](int i) { return ar[16 * ks + i]; }),
                                                                        pack8([&](int i) { return br[(16 * ks + i) * CP]; }), acc, 0, 0, 0);
                    const int n = nq * 32 + r;
                    if (n < 2 * Kc) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int p = pt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk;
                            if (p < NP) utL[p * 65 + n] = acc[e];
                        }
                    }
                }
                if ((wave == 2 || wave == 3) && (wave - 2) * 32 < 2 * Kc) {
                    // d(Wc)[f][(c,k)] += sum_j dloc(t+1)[j][f] wcat(t+1)[c][j + k - pad]   (rows f, columns (c,k), K = own positions)
                    const int n = min((wave - 2) * 32 + r, 2 * Kc - 1), c = n / Kc, k = n - c * Kc;
                    const float* ar = dlocL + pad * F1 + min(r, F);
                    const float* br = wold + c * TwP + k;
                    for (int ks = 0; ks * 16 < len; ++ks) {              // (uniform trip count: every lane takes part in an MFMA)
                        const int j0 = ks * 16 + 8 * hk;
                        accWc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            pack8([&](int i) { const float x = ar[min(j0 + i, len - 1) * F1]; return j0 + i < len ? x : 0.f; }),
                            pack8([&](int i) { return br[min(j0 + i, len - 1)]; }), accWc, 0, 0, 0);
                    }
                }
            }
            __syncthreads();
            T2_BSTAMP(12);
            if (ep > 0 && tid < 2 * len) {
                const int c = tid / len, i = tid - c * len;
                const float* qp = utL + (i + 2 * pad) * 65 + c * Kc;
                float sum = 0.f;
                for (int k = 0; k < Kc; ++k) sum += qp[-k * 64];
                if (c == 0) carryL[i] = sum; else ccL[i] += sum;
            }
            __syncthreads();
            // the saved tile and features into LDS (the tile's room held Q until here)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int w = wave + NWV * k, a = (w & 7) * 16 + (lane & 15), j4 = (w >> 3) * 4 + (lane >> 4);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * j4 + c < len) utL[(4 * j4 + c) * UP + a] = ureg[k][c];
            }
            if (tid < len * (F / 4)) {
                float* lw = locL + (tid / (F / 4)) * F1 + (tid % (F / 4)) * 4;
                lw[0] = lreg[0]; lw[1] = lreg[1]; lw[2] = lreg[2]; lw[3] = lreg[3];
            }
            if (tid < len) locL[tid * F1 + F] = 0.f;                      // pad column (read by lanes r >= F of the d(Wd) product)
            // ------------------------------------------------------------------------------------------------------------
            T2_BSTAMP(13);
            __syncthreads();
            if (*abortw) return;
            T2_BSTAMP(0);
            if (tid < E) {
                float v = in[4] + in[5];
                if (ep > 0) {
                    const unsigned off = (unsigned)((t + 1) & 1) * pbc_half + (unsigned)as * pbc_s + (unsigned)((ab_ * E + tid) * 4);
                    float pv[AKP];
                    load_tagged4(rsC, off, pbc_kp, step_tag(ep - 1), pv, 7u);       // (G(t+1)'s tag on every word)
                    v += (pv[0] + pv[1]) + (pv[2] + pv[3]);
                }
                dctxL[tid] = v;
                if (split == 0) AS.dctx_out[((long)t * B + ab_) * E + tid] = v;
            }
            __syncthreads();
            T2_BSTAMP(1);
            // g_j = dctx . memory_j + dalign_j + carried gradients (own positions)
            for (int jl = wave; jl < len; jl += NWV) {
                float sum = 0.f;
                for (int c = lane * 8; c < E; c += 512) {
                    const bf16x8 mb = *reinterpret_cast<const bf16x8*>(memL + jl * E + c);
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(dctxL + c), d1 = *reinterpret_cast<const f32x4*>(dctxL + c + 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) sum += (float)mb[k] * d0[k] + (float)mb[4 + k] * d1[k];
                }
                sum = wave_sum(sum);
                if (lane == 0) gL[jl] = sum;
            }
            __syncthreads();
            float part = 0.f;
            if (tid < len) {
                float gsum = gL[tid] + in[3];
                if (ep > 0) gsum += carryL[tid] + ccL[tid];
                gL[tid] = gsum;
                part = wL[tid] * gsum;
            }
            // softmax backward needs S = sum_j w_j g_j over BOTH splits: each publishes its partial (value + step tag, one
            // write-through store) and reads the partner's
            part = wave_sum(part);
            if (lane == 0) lredL[wave] = part;
            __syncthreads();
            if (wave == 0) {
                float mine = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) mine += lredL[w];
                const unsigned sbase = (unsigned)(pad * F * 2 * d.NS * B * 2) * 4u;                 // behind the halo rows
                const unsigned soff = sbase + (unsigned)((((t & 1) * d.NS + as) * B + ab_) * 2) * 8u;
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                if (lane == 0) {
                    u32x2 o; o[0] = __builtin_bit_cast(unsigned, mine); o[1] = (unsigned)(t + 1);
                    __builtin_amdgcn_raw_buffer_store_b64(o, rsK, soff + (unsigned)split * 8u, 0, SC1);
                }
                float other = 0.f;
                if (len < Tin) {                                          // (a memory short enough for one split has no partner)
                    const unsigned long long t0c = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        const u32x2 v2 = __builtin_amdgcn_raw_buffer_load_b64(rsK, soff + (unsigned)(1 - split) * 8u, 0, SC1);
                        if (v2[1] == (unsigned)(t + 1)) { other = __builtin_bit_cast(float, v2[0]); break; }
                        if (__builtin_amdgcn_s_memrealtime() - t0c > SPIN_TICKS) { if (lane == 0) { report_abort(d.err, 12u); *abortw = 1; } break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (lane == 0) lredL[NWV] = split == 0 ? mine + other : other + mine;          // same order in both splits
            }
            __syncthreads();
            if (*abortw) return;
            const float sdot = lredL[NWV];
            if (tid < len) deL[tid] = wL[tid] * (gL[tid] - sdot);
            __syncthreads();
            T2_BSTAMP(2);
            // energies backward on the tanh tile: dpre = de_j v_a (1 - u^2) replaces u in place; d(pm), dq, dv
            {
                const int gid = tid >> 4, sub = tid & 15;
                f32x4 dq[2], dv[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) { dq[k] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                for (int jl = gid; jl < len; jl += NTH / 16) {
                    const float dej = deL[jl];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int a = sub * 4 + 64 * k;
                        const f32x4 u4 = *reinterpret_cast<const f32x4*>(utL + jl * UP + a);
                        f32x4 acc = *reinterpret_cast<const f32x4*>(dpmL + jl * A + a);
                        f32x4 dp;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float dpre = dej * vL[a + c] * (1.0f - u4[c] * u4[c]);
                            dq[k][c] += dpre; dv[k][c] += dej * u4[c]; acc[c] += dpre; dp[c] = dpre;
                        }
                        *reinterpret_cast<f32x4*>(dpmL + jl * A + a) = acc;
                        *reinterpret_cast<f32x4*>(utL + jl * UP + a) = dp;
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        dq[k][c] += __shfl_xor(dq[k][c], 16, 64); dq[k][c] += __shfl_xor(dq[k][c], 32, 64);
                        dv[k][c] += __shfl_xor(dv[k][c], 16, 64); dv[k][c] += __shfl_xor(dv[k][c], 32, 64);
                    }
                if (lane < 16) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        *reinterpret_cast<f32x4*>(lredL + wave * A + sub * 4 + 64 * k) = dq[k];
                        *reinterpret_cast<f32x4*>(lredL + (NWV + wave) * A + sub * 4 + 64 * k) = dv[k];
                    }
                }
            }
            __syncthreads();
            if (tid < A) {
                float sq = 0.f, sv = 0.f;
#pragma unroll
                for (int k = 0; k < NWV; ++k) { sq += lredL[k * A + tid]; sv += lredL[(NWV + k) * A + tid]; }
                dqoL[tid] = sq;
                dvaL[tid] += sv;
            }
            // dloc[j][f] = sum_a dpre[j][a] Wd[a][f] (waves 0..1, bf16 operands as every large product of this mode) and
            // d(Wd)[a][f] += sum_j dpre[j][a] loc[j][f] (waves 2..5), both straight off the tile
            if (wave < njt) {
                // Wd^T fragments (column f = r, channels 16i + 8hk ..): L2-resident, requested here (held across the tile pass they spill)
                bf16x8 wdt[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wdt[i] = *reinterpret_cast<const bf16x8*>(AS.wdt16 + (long)min(r, F - 1) * A + 16 * i + 8 * hk);
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                const float* ur = utL + min(wave * 32 + r, len - 1) * UP + 8 * hk;
#pragma unroll
                for (int i = 0; i < 8; ++i) {                            // (A = 128)
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(ur + 16 * i), hi = *reinterpret_cast<const f32x4*>(ur + 16 * i + 4);
                    bf16x8 af;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { af[c] = (__bf16)lo[c]; af[4 + c] = (__bf16)hi[c]; }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wdt[i], acc, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int jl = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk;
                    if (jl < len && r < F) dlocL[(pad + jl) * F1 + r] = acc[e];
                }
            } else if (wave >= 2 && wave < 6) {
                const float* ar = utL + (wave - 2) * 32 + r;
                const float* br = locL + min(r, F);
                for (int ks = 0; ks * 16 < len; ++ks) {
                    const int j0 = ks * 16 + 8 * hk;
                    accWd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        pack8([&](int i) { const float x = ar[min(j0 + i, len - 1) * UP]; return j0 + i < len ? x : 0.f; }),
                        pack8([&](int i) { return br[min(j0 + i, len - 1) * F1]; }), accWd, 0, 0, 0);
                }
            }
            __syncthreads();
            // hand-offs: dq partial of this split, the `pad` rows of dloc next to the boundary
            if (tid < A / 4) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(dqoL + tid * 4);
                u32x4 qw = __builtin_bit_cast(u32x4, q4);
                qw.x = (qw.x & ~1u) | ((ep & 1u) ^ 1u);                  // (the dq buffer has one parity: its tag flips every step)
                __builtin_amdgcn_raw_buffer_store_b128(qw, rsQ, (unsigned)((((as * 2 + split) * B + ab_) * A + tid * 4) * 4), 0, SC1);
            } else if (tid >= 64 && tid < 64 + pad * (F / 4)) {
                const int i = tid - 64, row = i / (F / 4), f4 = (i % (F / 4)) * 4;
                const float* src = dlocL + (pad + (split == 0 ? len - pad + row : row)) * F1 + f4;
                const f32x4 h4 = {tag_f32(src[0], step_tag(ep)), src[1], src[2], src[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h4), rsK,
                    (unsigned)(((((t & 1) * d.NS + as) * B + ab_) * 2 + split) * pad * F + row * F + f4) * 4u, 0, SC1);
            }
            T2_BSTAMP(3);
            __syncthreads();                                             // (no arrival counter: the dq units carry the step's tag)
            T2_BSTAMP(4);
            if (tid < A) AS.dq_out[((long)t * B + ab_) * 2 * A + split * A + tid] = dqoL[tid];
            if (t > d.t0) load_ain(t - 1, tid);
            }
        }
        // ======================================================================================= P(t)
        if (hasP) {
            float in[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) in[i] = pin[i];
            T2_BSTAMP(5);
            // dq rows of this row tile (two position splits summed), K-split partials of dx_h(t+1)
            {
                u32x4 qv[2][2];
                {
                    const unsigned want = (ep & 1u) ^ 1u;
                    const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        unsigned okw = 1u;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int idx = tv + i * NTH, row = idx / (A / 4), a4 = (idx % (A / 4)) * 4;
                            const int b = min(rt * 32 + row, B - 1);
#pragma unroll
                            for (int sp = 0; sp < 2; ++sp) {
                                qv[i][sp] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (unsigned)((((ps * 2 + sp) * B + b) * A + a4) * 4), 0, SC1);
                                okw &= ((qv[i][sp].x & 1u) == want) ? 1u : 0u;
                            }
                        }
                        if (__all(okw != 0u)) break;
                        if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if ((tv & 63) == 0) { report_abort(d.err, 8u); *abortw = 1; } break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                float pv[AKP] = {0.f, 0.f, 0.f, 0.f};
                if (ep > 0) {
                    const unsigned off = (unsigned)((t + 1) & 1) * pbh_half + (unsigned)ps * pbh_s + (unsigned)((((ug * MT + rt) * 32 + (tv >> 4)) * PU + (tv & 15)) * 4);
                    load_tagged4(rsH, off, pbh_kp, step_tag(ep - 1), pv, 9u);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = tv + i * NTH, row = idx / (A / 4), a4 = (idx % (A / 4)) * 4;
                    *reinterpret_cast<f32x4*>(dqL + row * (A + 4) + a4) = __builtin_bit_cast(f32x4, qv[i][0]) + __builtin_bit_cast(f32x4, qv[i][1]);
                }
                in[0] += (pv[0] + pv[1]) + (pv[2] + pv[3]);
            }
            __syncthreads();
            if (*abortw) return;
            float dh = in[0];
            {   // + dq . Wq[:, unit]   (attention.py:68: the query projection's input gradient): a [32 rows x A] . [A x 16 units] product
                // on the matrix cores, K quarter per wave (waves 0..3), both operands as bf16 hi + lo (lo.lo dropped: 2^-16
                // relative) — the thread-per-(row, unit) dot product it replaces read 64 x 16 bytes of LDS per thread and step
                if (wave < 4) {
                    const float* ar = dqL + r * (A + 4) + wave * 32 + 8 * hk;
                    const __bf16* bw = wq16L + (r & 15) * (A + 8) + wave * 32 + 8 * hk;
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {                     // (A = 128: 32 per wave)
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(ar + 16 * ks), hi4 = *reinterpret_cast<const f32x4*>(ar + 16 * ks + 4);
                        bf16x8 ah, al;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            ah[c] = (__bf16)lo4[c]; al[c] = (__bf16)(lo4[c] - (float)ah[c]);
                            ah[4 + c] = (__bf16)hi4[c]; al[4 + c] = (__bf16)(hi4[c] - (float)ah[4 + c]);
                        }
                        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(bw + 16 * ks), wl = *reinterpret_cast<const bf16x8*>(bw + PU * (A + 8) + 16 * ks);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
                    }
                    if (r < PU) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) dpL[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * (PU + 1) + r] = acc[e];
                    }
                }
                __syncthreads();
                const float* dp = dpL + (tid >> 4) * (PU + 1) + (tid & 15);
                dh += (dp[0] + dp[32 * (PU + 1)]) + (dp[2 * 32 * (PU + 1)] + dp[3 * 32 * (PU + 1)]);
            }
            float dcs = ep > 0 ? dc : 0.f;
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
#pragma unroll
            for (int g = 0; g < 4; ++g) { dgL[(g * 32 + (tid >> 4)) * (PU + 4) + (tid & 15)] = dgv[g]; bacc[g] += dgv[g]; }
            __syncthreads();
            if (wave < 4) {
                const float* hp = dgL + (wave * 32 + r) * (PU + 4) + hk * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(hp), hi = *reinterpret_cast<const f32x4*>(hp + 4);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
                u32x4 ow = __builtin_bit_cast(u32x4, o);
                ow.x = (ow.x & ~1u) | step_tag(ep);
                __builtin_amdgcn_raw_buffer_store_b128(ow, rsX,
                    (unsigned)((t & 1) * d.NS + ps) * xs + (unsigned)((((wave * H + u0) / 16) * MT + rt) * 1024 + lane * 16), 0, SC1);
            }
            T2_BSTAMP(6);
            __syncthreads();                                             // (no arrival counter: the fragments carry the step's tag)
            T2_BSTAMP(7);
            {
                const int b = rt * 32 + (tv >> 4), u = u0 + (tv & 15);
                if (b < B) {
                    float* gp = PS.dg + ((long)t * B + b) * K4 + u;
                    gp[0] = dgv[0]; gp[H] = dgv[1]; gp[2 * H] = dgv[2]; gp[3 * H] = dgv[3];
                }
                if (t > d.t0) load_pin(t - 1, tv);
            }
        }
        // ======================================================================================= G(t)
        if (hasG && t > 0) {
            T2_BSTAMP(8);
            const unsigned xb = (unsigned)((t & 1) * d.NS + gs) * xs + (unsigned)lane * 16u;
            if constexpr (KIND == CHAIN_LSA) {
                // LSA keeps two MFMA accumulators of weight gradients in registers for all steps: no room for the 16 fragments of a
                // K quarter next to them (tried: 24-64 spilled registers, 24.7 -> 25.7 us per step).  K split over the eight waves,
                // one reduction round per (row tile, column tile), fragments requested one row tile at a time
                u32x4 af[8];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    {
                        const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
                        for (;;) {                                           // (tagged fragments: load until all eight are this step's)
                            unsigned okw = 1u;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                af[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xb + (unsigned)(((kp * (H / 16) + wave * 8 + i) * MT + m) * 1024), 0, SC1);
                                okw &= ((af[i].x & 1u) == step_tag(ep)) ? 1u : 0u;
                            }
                            if (__all(okw != 0u)) break;
                            if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if (lane == 0) { report_abort(d.err, 10u); *abortw = 1; } break; }
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        f32x16 acc;
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), W[c * 8 + i], acc, 0, 0, 0);
                        if (m + c > 0) __syncthreads();
#pragma unroll
                        for (int e = 0; e < 16; ++e) partL[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PPR + r] = acc[e];
                        __syncthreads();
                        if (tv < 256) {
                            const int row = tv >> 3, c4 = (tv & 7) * 4;
                            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int w = 0; w < NWV; ++w) sum += *reinterpret_cast<const f32x4*>(partL + (w * 32 + row) * PPR + c4);
                            const int col = nt * ANC + c * 32 + c4;                 // column of [ctx | h]
                            const unsigned tg = step_tag(ep);                       // every word carries the tag (the consumers read single words)
                            sum[0] = tag_f32(sum[0], tg); sum[1] = tag_f32(sum[1], tg); sum[2] = tag_f32(sum[2], tg); sum[3] = tag_f32(sum[3], tg);
                            if (nt < NTC) {
                                const int b = m * 32 + row;
                                if (b < B)
                                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rsC,
                                        (unsigned)(t & 1) * pbc_half + (unsigned)gs * pbc_s + (unsigned)kp * pbc_kp + (unsigned)((b * E + col) * 4), 0, SC1);
                            } else {
                                const int u = col - E;
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rsH,
                                    (unsigned)(t & 1) * pbh_half + (unsigned)gs * pbh_s + (unsigned)kp * pbh_kp +
                                    (unsigned)(((((u / PU) * MT + m) * 32 + row) * PU + (u % PU)) * 4), 0, SC1);
                            }
                        }
                    }
                }
            } else {
                const int gkq = __builtin_amdgcn_readfirstlane(tv >> 7), gc2 = __builtin_amdgcn_readfirstlane((tv >> 6) & 1);   // (scalar: the 16 fragment offsets then sit in SGPRs)
                // row tiles one after the other: the next tile's fragments are in flight underneath the current tile's reduction round
                u32x4 af[16];
                auto load_m = [&](int m) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        af[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xb + (unsigned)(((kp * (H / 16) + gkq * 16 + i) * MT + m) * 1024), 0, SC1);
                };
                auto ensure_m = [&](int m) {                              // tagged fragments: load again until all sixteen are this step's
                    const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        unsigned okw = 1u;
#pragma unroll
                        for (int i = 0; i < 16; ++i) okw &= ((af[i].x & 1u) == step_tag(ep)) ? 1u : 0u;
                        if (__all(okw != 0u)) break;
                        if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if (lane == 0) { report_abort(d.err, 10u); *abortw = 1; } break; }
                        __builtin_amdgcn_s_sleep(2);
                        load_m(m);
                    }
                };
                load_m(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ensure_m(m);
                    f32x16 acc, acc2;                                        // two interleaved chains
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), W[i], acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i + 1]), W[i + 1], acc2, 0, 0, 0);
                    }
                    if (m + 1 < MT) load_m(m + 1);
                    if (m > 0) __syncthreads();
#pragma unroll
                    for (int e = 0; e < 16; ++e) partL[((gkq * 2 + gc2) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PPR + r] = acc[e] + acc2[e];
                    __syncthreads();
                    {   // thread (column tile c, row, 4 columns): fixed-order sum of the four K quarters
                        const int c = tv >> 8, row = (tv & 255) >> 3, c4 = (tv & 7) * 4;
                        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q = 0; q < 4; ++q) sum += *reinterpret_cast<const f32x4*>(partL + ((q * 2 + c) * 32 + row) * PPR + c4);
                        const int col = nt * ANC + c * 32 + c4;                 // column of [ctx | h]
                        const unsigned tg = step_tag(ep);                       // every word carries the tag (the consumers read single words)
                        sum[0] = tag_f32(sum[0], tg); sum[1] = tag_f32(sum[1], tg); sum[2] = tag_f32(sum[2], tg); sum[3] = tag_f32(sum[3], tg);
                        if (nt < NTC) {
                            const int b = m * 32 + row;
                            if (b < B)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rsC,
                                    (unsigned)(t & 1) * pbc_half + (unsigned)gs * pbc_s + (unsigned)kp * pbc_kp + (unsigned)((b * E + col) * 4), 0, SC1);
                        } else {
                            const int u = col - E;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rsH,
                                (unsigned)(t & 1) * pbh_half + (unsigned)gs * pbh_s + (unsigned)kp * pbh_kp +
                                (unsigned)(((((u / PU) * MT + m) * 32 + row) * PU + (u % PU)) * 4), 0, SC1);
                        }
                    }
                }
            }
            T2_BSTAMP(9);
            __syncthreads();                                             // (no arrival counter: every partial word carries the step's tag)
            if (*abortw) return;
            T2_BSTAMP(10);
        }
    }
#ifdef T2_STAMPS
    if (tid == 0)
        for (int i = 0; i < 16; ++i) t2_chain_bwd_stamps[wg * 16 + i] += t2_stamp_lds[i];
#endif
    // ---------------------------------------------------------------- P epilogue: bias gradients (fixed-order sum over the tile's rows)
    if (PS.dbias_part) {
        __syncthreads();
        if (hasP) {
#pragma unroll
            for (int g = 0; g < 4; ++g) dgL[(g * 32 + (tid >> 4)) * (PU + 4) + (tid & 15)] = bacc[g];
        }
        __syncthreads();
        if (hasP && tid < 4 * PU) {
            const int g = tid / PU, u = tid % PU;
            float sum = 0.f;
            for (int row = 0; row < 32; ++row) sum += dgL[(g * 32 + row) * (PU + 4) + u];
            PS.dbias_part[(long)rt * K4 + g * H + u0 + u] = sum;
        }
        __syncthreads();
    }
    // ---------------------------------------------------------------- A epilogue: the accumulators leave LDS
    if (hasA) {
        if constexpr (KIND == CHAIN_LSA) {
            // location-layer weight gradients of this (split, item): the MFMA accumulators of waves 2..5 / 2..3
            if (wave >= 2 && wave < 6 && r < F) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int a = (wave - 2) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk;
                    AS.ddense_acc[(((long)split * B + ab_) * A + a) * F + r] = accWd[e];
                }
            }
            const int n = (wave - 2) * 32 + r;
            if (wave >= 2 && wave < 4 && n < 2 * Kc) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int f = (e & 3) + 8 * (e >> 2) + 4 * hk;
                    if (f < F) AS.dconv_acc[(((long)split * B + ab_) * F + f) * 2 * Kc + n] = accWc[e];
                }
            }
        }
        for (int a = tid; a < A; a += NTH) AS.dv_acc[((long)split * B + ab_) * A + a] = dvaL[a];
        for (int i = tid; i < len * (A / 4); i += NTH) {
            const int jl = i / (A / 4), a4 = (i % (A / 4)) * 4;
            *reinterpret_cast<f32x4*>(AS.dpm_acc + ((long)ab_ * Tin + jb + jl) * A + a4) = *reinterpret_cast<const f32x4*>(dpmL + jl * A + a4);
        }
    }
}

}  // namespace

bool chain_bwd_plan(ChainBwdDesc& d) {
    if (d.H != 1024 || d.B < 1 || d.B > 64) return false;
    if (chain_device_cus() < 256 || !chain_device_claim()) return false;
    if (d.kind == CHAIN_LSTM) return true;
    if ((d.kind != CHAIN_SMA && d.kind != CHAIN_LSA) || d.E != 512 || d.A != 128 || d.NS < 1 || d.NS > 2) return false;
    const bool lsa = d.kind == CHAIN_LSA;
    if (lsa && (d.F != 32 || d.Kc < 3 || d.Kc > 31 || d.Kc % 2 == 0)) return false;     // (F = 32: the reference's attention_location_n_filters)
    const int pad = lsa ? (d.Kc - 1) / 2 : 0;
    int tc = 0;
    for (int s = 0; s < d.NS; ++s) {
        const int Tin = d.st[s].Tin, chunk = (((Tin + 1) / 2) + 3) & ~3;
        if (Tin < 16) return false;                                  // (two position splits per item, as the launch path at these sizes)
        // LSA: both splits hold at least the `pad` rows they hand to each other; one [positions x A] tile row pair per wave
        if (lsa && (Tin - chunk < pad || chunk < pad || chunk > 64)) return false;
        if (lsa && (!d.st[s].usave || !d.st[s].locsave)) return false;   // the forward chain's tanh tile / location features
        tc = std::max(tc, chunk);
    }
    d.lds_Tc = tc;
    return (size_t)bwd_lds_of(d.A, d.E, tc, (d.B + 31) / 32, d.kind, d.F, d.Kc).total * sizeof(float) <= 160 * 1024;
}

size_t chain_bwd_exchange_bytes(const ChainBwdDesc& d, size_t* x_bytes, size_t* pb_bytes) {
    const size_t MT = (d.B + 31) / 32;
    *x_bytes = (size_t)2 * (4 * d.H / 16) * MT * 1024;
    *pb_bytes = (size_t)2 * GKP * (d.H / PU) * MT * 32 * PU * sizeof(float);
    return *x_bytes + *pb_bytes;
}

size_t chain_bwd_lsa_tagged_bytes(const ChainBwdDesc& d) {
    return ((size_t)(((d.Kc - 1) / 2) * d.F + 2) * 2 * d.NS * d.B * 2 * sizeof(float) + 255) & ~(size_t)255;
}

size_t chain_bwd_att_exchange_bytes(const ChainBwdDesc& d, size_t* x_bytes, size_t* pbh_bytes, size_t* pbc_bytes, size_t* dqx_bytes, size_t* carry_bytes) {
    const size_t MT = (d.B + 31) / 32, al = 255;
    *x_bytes = (size_t)2 * d.NS * (4 * d.H / 16) * MT * 1024;
    *pbh_bytes = (size_t)2 * d.NS * AKP * (d.H / PU) * MT * 32 * PU * sizeof(float);
    *pbc_bytes = ((size_t)2 * d.NS * AKP * d.B * d.E * sizeof(float) + al) & ~al;
    *dqx_bytes = ((size_t)d.NS * 2 * d.B * d.A * sizeof(float) + al) & ~al;
    *carry_bytes = ((size_t)2 * d.NS * d.B * sizeof(float) + al) & ~al;
    if (d.kind == CHAIN_LSA)      // halo rows + softmax-dot slots (cleared per launch), then the bf16 Wd^T copies [NS][F][A]
        *carry_bytes = chain_bwd_lsa_tagged_bytes(d) + (((size_t)d.NS * d.F * d.A * sizeof(__bf16) + al) & ~al);
    return *x_bytes + *pbh_bytes + *pbc_bytes + *dqx_bytes + *carry_bytes;
}

int chain_bwd(const ChainBwdDesc& d, hipStream_t s) {
    T2_REQUIRE(d.t1 > d.t0 && d.t0 >= 0, "chain_bwd: bad step range [%d,%d)", d.t0, d.t1);
    T2_REQUIRE(d.X && d.PB && d.cnt && d.err, "chain_bwd: exchange buffers missing");
    const int MT = (d.B + 31) / 32;
    T2_CHECK_HIP(hipMemsetAsync(d.cnt, 0, kChainBwdCntBytes, s));
    if (d.kind == CHAIN_SMA || d.kind == CHAIN_LSA) {
        T2_REQUIRE(d.t0 == 0 && d.t1 == d.T, "chain_bwd: the attention chain runs its whole step range in one launch");
        T2_REQUIRE(d.PBC && d.DQX && d.CARRYX, "chain_bwd: exchange buffers missing");
        const size_t smem = (size_t)bwd_lds_of(d.A, d.E, d.lds_Tc, MT, d.kind, d.F, d.Kc).total * sizeof(float);
        const int grid = std::max(std::max(d.NS * (d.E + d.H) / ANC * AKP, d.NS * (d.H / PU) * MT), d.NS * d.B * 2);
        {   // every exchange buffer carries step tags (tag 0 = not written yet): clear them
            size_t xb, ph, pc, dq, cr;
            chain_bwd_att_exchange_bytes(d, &xb, &ph, &pc, &dq, &cr);
            const size_t carry = d.kind == CHAIN_LSA ? chain_bwd_lsa_tagged_bytes(d) : cr;      // (LSA: the caller's bf16 Wd^T copies follow)
            unsigned char* c0 = reinterpret_cast<unsigned char*>(d.cnt);
            if (d.X == c0 + kChainBwdCntBytes && d.PB == d.X + xb && d.PBC == d.PB + ph && reinterpret_cast<unsigned char*>(d.DQX) == d.PBC + pc &&
                reinterpret_cast<unsigned char*>(d.CARRYX) == reinterpret_cast<unsigned char*>(d.DQX) + dq) {
                T2_CHECK_HIP(hipMemsetAsync(d.X, 0, xb + ph + pc + dq + carry, s));             // (one region, as c_api.hip lays it out)
            } else {
                T2_CHECK_HIP(hipMemsetAsync(d.X, 0, xb, s));
                T2_CHECK_HIP(hipMemsetAsync(d.PB, 0, ph, s));
                T2_CHECK_HIP(hipMemsetAsync(d.PBC, 0, pc, s));
                T2_CHECK_HIP(hipMemsetAsync(d.DQX, 0, dq, s));
                T2_CHECK_HIP(hipMemsetAsync(d.CARRYX, 0, carry, s));
            }
        }
        auto launch = [&](auto kernel) -> int {
            T2_TRY_RC(persistent_prepare(kernel, grid, smem));
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(NTH), smem, s, d);
            return 0;
        };
        int rc;
        if (d.kind == CHAIN_SMA) rc = MT == 1 ? launch(chain_bwd_att_kernel<1, CHAIN_SMA>) : launch(chain_bwd_att_kernel<2, CHAIN_SMA>);
        else rc = MT == 1 ? launch(chain_bwd_att_kernel<1, CHAIN_LSA>) : launch(chain_bwd_att_kernel<2, CHAIN_LSA>);
        if (rc) return rc;
        T2_LAUNCH_CHECK();
        return 0;
    }
    T2_REQUIRE(d.kind == CHAIN_LSTM, "chain_bwd: kind %d not covered", d.kind);
    T2_REQUIRE(d.t1 == d.T, "chain_bwd: the decoder-LSTM chain runs its whole step range in one launch (tagged hand-offs count steps from the launch)");
    {   // tagged hand-offs: fragments and partials start out with tag 0 (= invalid for the first two steps)
        size_t xb = 0, pb = 0;
        chain_bwd_exchange_bytes(d, &xb, &pb);
        T2_CHECK_HIP(hipMemsetAsync(d.X, 0, xb + pb, s));
    }
    const size_t smem = (size_t)(4 + NWV * MT * 32 * PPR) * sizeof(float);
    const int grid = (d.H / GNC) * GKP;
    if (MT == 1) {
        T2_TRY_RC(persistent_prepare(chain_bwd_lstm_kernel<1>, grid, smem));
        hipLaunchKernelGGL(chain_bwd_lstm_kernel<1>, dim3(grid), dim3(NTH), smem, s, d);
    } else {
        T2_TRY_RC(persistent_prepare(chain_bwd_lstm_kernel<2>, grid, smem));
        hipLaunchKernelGGL(chain_bwd_lstm_kernel<2>, dim3(grid), dim3(NTH), smem, s, d);
    }
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
