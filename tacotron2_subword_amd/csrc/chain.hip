// Persistent decoder-chain kernel: many consecutive time steps of the teacher-forced recurrences in ONE launch.
//
//   chain A (model.py:322-369, attention.py:7-85,291-398): both attention LSTMs + attention (SMA or LSA)
//   chain B (model.py:371-373):                            the decoder LSTM over hoisted input pre-activations
//
// Why: as separate launches every step re-fetched its recurrent weights (25 MB of bf16 for chain A) and each launch
// spent 5-9 us receiving operands through one CU (profiles/README.md, round 1).  Here a workgroup keeps ITS slice of
// the recurrent weights in registers for all steps (256 workgroups x 96 KB = the whole [W_hh | W_ih[:,P:]] of both
// attention LSTMs), its cell state in registers, its attention state (previous alignment, cumulative weights) and as
// much of its item's processed memory / encoder memory as fits in LDS.  What crosses workgroups per step is only
//   h_t   (bf16, MFMA-fragment order)        units -> every workgroup of the same stream and row group
//   q_t   (fp32 partial query projections)   units -> the item's attention workgroups
//   ctx_t (bf16, MFMA-fragment order)        items -> every LSTM workgroup of the same stream and row group
// exchanged through L2-bypassing write-through stores (sc1) and sc1 loads.  Decode loop: one arrival counter per (stream,
// row group, kind) — MI355X_MICROARCH.md "Valid forms": every payload store drained by every storing wave before the
// workgroup barrier, one lane signals with an agent-scope atomic, one wave polls with relaxed agent loads, the other waves
// load behind the workgroup barrier (scripts/persist_probe.hip measures this pattern).  Teacher-forced chains (round 3): NO
// counters — every 16-byte unit carries a step tag in the lowest bit of its first word, the buffers are cleared per launch,
// a consumer loads until its units show the step's tag (see TAG / TAGQ in the kernel): one round trip per hop, two hops per step.
//
// Work items (one workgroup of 512 threads per CU, grid = 256):
//   L item (s, ug, rg): stream s, unit group ug (8*UT hidden units = 32*UT gate columns), row group rg (32*RT batch rows).
//       gates[32*RT x 32*UT] = pre[t] + [h_{t-1} | ctx_{t-1}] . W^T : the 8 waves split K, operands straight from the
//       fragment-ordered exchange buffer into MFMA registers (one coalesced 1 KB load per fragment), partial tiles summed
//       through LDS in fixed order, gates / cell / dropout per (row, unit), h_t published, fp32 query partial via
//       v_mfma_f32_32x32x2_f32 on the fp32 h_t.
//   A item (b, s, part): batch item b, stream s, 1/CS of the context columns.  Every part recomputes the (cheap)
//       energies / alignment so that no exchange is needed inside an item; each part sums its own context columns.
// Numerics = the launch-per-step bf16 path (lstm.hip / attention.hip) up to summation order, except: the energy tanh
// and the SMA sigmoid use the hardware exp (abs. error ~1e-7), and context rows resident in LDS are bf16 copies of the
// memory (every consumer of ctx in bf16 mode rounds it to bf16 anyway).
#include <algorithm>
#include <type_traits>

#include "chain_common.h"

// Diagnostic build (T2_EXTRA_HIPCC_FLAGS=-DT2_STAMPS, scripts/chain_stamps.py): thread 0 of every workgroup adds the
// realtime-counter ticks (100 MHz) of each segment of a step into t2_chain_stamps[workgroup][segment].
#ifdef T2_STAMPS
__device__ unsigned long long t2_chain_stamps[256 * 16];
extern "C" int t2_debug_read_chain_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(t2_chain_stamps), sizeof(unsigned long long) * n);
}
extern "C" int t2_debug_clear_chain_stamps(void) {
    static unsigned long long z[256 * 16];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(t2_chain_stamps), z, sizeof(z));
}
#define T2_CSTAMP(i)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); t2_stamp_lds[i] += n_ - stamp_last; stamp_last = n_; } \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define T2_CSTAMP(i)
#endif

namespace t2 {

namespace {

using namespace chain;

struct Geo {                                // derived on both sides from the descriptor
    int MT, NRG, NUG, nL, nA, KT; unsigned xs_bytes;
};
__host__ __device__ inline Geo geo_of(const ChainDesc& d, int UT, int RT) {
    Geo g;
    g.MT = (d.B + 31) / 32; g.NRG = (g.MT + RT - 1) / RT; g.NUG = d.H / (8 * UT);
    g.nL = d.NS * g.NUG * g.NRG; g.nA = d.kind == CHAIN_LSTM ? 0 : d.NS * d.B * d.CS;
    g.KT = (d.H + (d.kind == CHAIN_LSTM ? 0 : d.E) + (d.dec ? d.P : 0)) / 16;
    g.xs_bytes = (unsigned)g.KT * g.MT * 1024u;
    return g;
}

// LDS carve (floats).  Persistent part first, then a scratch area shared by the two phases.
struct Lds { int ab, v, ap, cum, q, e, an, cs, pm, mem, convw, dense, loc, wpad, pa, lsp, scratch, total; };
__host__ __device__ inline Lds lds_of(const ChainDesc& d, int UT, int RT, int Tin, int Jp, int Jm) {
    Lds m; int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    const int Tp = (Tin + 3) & ~3, EC = d.kind == CHAIN_LSTM ? 0 : d.E / d.CS;
    m.ab = take(4);
    if (d.kind != CHAIN_LSTM) {
        m.v = take(d.A + 4); m.ap = take(Tp + 4); m.cum = take(Tp + 4); m.q = take(d.A); m.e = take(Tp + 4); m.an = take(Tp + 4); m.cs = take(EC);
        m.pm = take(Jp * (d.kind == CHAIN_LSA ? d.A + 1 : d.A)); m.mem = take(Jm * EC / 2);    // (LSA reads pm position-major: odd pitch)
        // decode loop: the projection / prenet workgroups keep their weight fragments (64 KB) in this area instead of the
        // attention rows, however short the memories are
        if (d.dec && o - m.pm < 16384) take(16384 - (o - m.pm));
    } else { m.v = m.ap = m.cum = m.q = m.e = m.an = m.cs = m.pm = m.mem = o; }
    m.convw = m.dense = o;
    // location layer weights, resident, each as a bf16 hi/lo pair: conv taps [2][F][2 KP + 8] (KP = Kc rounded up to 8, zero taps
    // behind Kc), dense layer [2][A][F + 8]
    if (d.kind == CHAIN_LSA) { m.convw = take(d.F * (2 * ((d.Kc + 7) & ~7) + 8)); m.dense = take(d.A * (d.F + 8)); }
    const int lpart = NWV * 32 * PPR, lhs = RT * 32 * (UT * 8 + 4), lq = d.kind == CHAIN_LSTM ? 0 : RT * 32 * (d.A + 4);
    const int lphase = (lpart > lq ? lpart : lq) + lhs;
    int aphase = d.kind == CHAIN_LSTM ? 0 : 16 * d.A + NWV * EC;
    m.loc = m.wpad = m.pa = m.lsp = 0;
    if (d.kind == CHAIN_LSA) {                               // per-step location features behind the reduction buffers
        const int TwP = (Tin + d.Kc - 1 + 8 + 3) & ~3;
        m.loc = aphase;                                           // (the fp32 copy of the features is gone: the bf16 pair below is what is read)
        m.wpad = aphase; aphase += 2 * TwP;
        m.pa = aphase; aphase += (d.A / 32) * ((Tin + 3) & ~3);   // energy partials of the channel tiles [A/32][Tp]
        m.lsp = aphase; aphase += (Tin * (d.F + 8) + 3) & ~3;     // the features again as a bf16 hi/lo pair [2][Tin][F + 8]
    }
    m.scratch = take(lphase > aphase ? lphase : aphase);
    m.loc += m.scratch; m.wpad += m.scratch; m.pa += m.scratch; m.lsp += m.scratch;
    m.total = o;
    return m;
}

template <int UT, int RT, int KH, int KC, int KIND, int KPN = 0>
__global__ __launch_bounds__(NTH) void chain_fwd_kernel(ChainDesc d) {
    constexpr int NTILE = UT * RT, NSLOT = (NTILE + 1) / 2, HSP = UT * 8 + 4;
    constexpr bool DEC = KPN > 0;                       // decode loop: prenet segment in the LSTM product + decoder LSTM, projections, prenets in the launch
    // Teacher-forced chains (every kind): NO arrival counters.  Fragments of step t carry bit 1 of t (inverted) in the lowest bit of
    // their first word, the exchange buffers are cleared per launch, and a consumer loads its fragments and loads them again until
    // they show the step's tag (bounded): one round trip per hop instead of a counter poll followed by the loads, no drain and no
    // atomic on the producer's side.  (Round 3, first try: tags checked BEHIND the polls — the attention kinds, at the 256-register
    // cap, spilled 32-58 registers and got slower; without the polls they spill 3-8 and gain 0.6 us per step.)  The decode loop
    // keeps the drained protocol with counters (its 8-byte dec_h pieces and five hops are not converted).
    constexpr bool TAG = !DEC;
    // Attention kinds, teacher-forced: the QUERY PARTIALS alone are tagged (bit 0 of the step count, inverted, in the first word of every
    // 16-byte unit; one buffer, rewritten every step, cleared per launch).  The A items then do not poll the h counter at all: they
    // load the partials and load again until the tags are this step's — one round trip instead of two behind the L items' publish,
    // and not behind its drain.  (The L items poll the h counter themselves, inside their wait for the contexts.)
    constexpr bool TAGQ = KIND != CHAIN_LSTM;             // (the decode loop too: its A items have just finished their L items — a short wait)
    auto tag_x = [](int step) { return (unsigned)(((step >> 1) & 1) ^ 1); };
    auto tag_q = [](int step) { return (unsigned)((step & 1) ^ 1); };
    const Geo G = geo_of(d, UT, RT);
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hk = lane >> 5;
    const int B = d.B, H = d.H, A = d.A;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // ---------------------------------------------------------------- items of this workgroup
    const bool hasL = wg < G.nL, hasA = KIND != CHAIN_LSTM && wg < G.nA;
    int ls = 0, ug = 0, rg = 0;
    if (hasL) { ls = wg / (G.NUG * G.NRG); const int rem = wg % (G.NUG * G.NRG); ug = rem / G.NRG; rg = rem % G.NRG; }
    int as = 0, ab_ = 0, part = 0;
    if (hasA) { as = wg / (B * d.CS); const int rem = wg % (B * d.CS); ab_ = rem / d.CS; part = rem % d.CS; }
    const ChainStream& LS = d.st[ls];
    const ChainStream& AS = d.st[as];
    const int u0 = ug * 8 * UT, row0 = rg * 32 * RT;
    const int arg = ab_ / (32 * RT), arow = ab_ % (32 * RT);         // row group / local row of the A item
    const int EC = KIND == CHAIN_LSTM ? 0 : d.E / d.CS, c0 = part * EC;
    const int Tin = hasA ? AS.Tin : 4, Tp = (Tin + 3) & ~3;
    // decode loop: the few workgroups that also own a projection / prenet item keep THOSE weights in the LDS area of the
    // attention operands (as ready-made MFMA B fragments) and read their attention rows from L2 instead
    const bool prole = DEC && ((wg >= 128 && wg < 128 + (d.M + 1 + 15) / 16) || (wg >= 192 && wg < 192 + d.NS * (d.P / 16)));
    const int Jp = (hasA && !prole) ? (as ? d.Jp[1] : d.Jp[0]) : 0, Jm = (hasA && !prole) ? (as ? d.Jm[1] : d.Jm[0]) : 0;
    const Lds M = lds_of(d, UT, RT, d.lds_Tin, d.lds_Jp, d.lds_Jm);   // one carve for every workgroup (largest stream)
    unsigned* abortw = reinterpret_cast<unsigned*>(smem + M.ab);
    float* vL = smem + M.v; float* apL = smem + M.ap; float* cumL = smem + M.cum; float* qL = smem + M.q;
    float* eL = smem + M.e; float* anL = smem + M.an; float* csL = smem + M.cs;
    float* pmL = smem + M.pm; __bf16* memL = reinterpret_cast<__bf16*>(smem + M.mem);
    float* partL = smem + M.scratch;                                   // [NWV][32][PPR]      (L phase)
    float* qsL = smem + M.scratch;                                     // [RT*32][A+4]        (L phase, after the reduce)
    const int lq = KIND == CHAIN_LSTM ? 0 : RT * 32 * (A + 4);
    float* hsL = smem + M.scratch + (NWV * 32 * PPR > lq ? NWV * 32 * PPR : lq);   // [RT*32][HSP]
    float* redL = smem + M.scratch;                                    // [16][A]             (A phase)
    float* credL = smem + M.scratch + 16 * A;                          // [NWV][EC]
    __bf16* convwL = reinterpret_cast<__bf16*>(smem + M.convw); __bf16* denseL = reinterpret_cast<__bf16*>(smem + M.dense);    // LSA: bf16 [hi|lo][F][2KP+8], [hi|lo][A][F+8]
    __bf16* lspL = reinterpret_cast<__bf16*>(smem + M.lsp);                                            // LSA: bf16 [hi|lo][Tin][F+8]
    float* wpadL = smem + M.wpad; float* paL = smem + M.pa;   // LSA: [2][TwP], [A/32][Tp]
    (void)convwL; (void)denseL; (void)wpadL; (void)paL; (void)lspL;
    (void)cumL; (void)qsL; (void)redL; (void)credL; (void)csL; (void)eL; (void)anL; (void)qL; (void)vL; (void)pmL; (void)memL;

    auto rsX = __builtin_amdgcn_make_buffer_rsrc(d.X, 0, (int)(2u * d.NS * G.xs_bytes), 0x00020000);
    auto rsQ = __builtin_amdgcn_make_buffer_rsrc(d.Q, 0, (int)d.q_bytes, 0x00020000);
    unsigned* cntH_L = d.cnt + (size_t)((ls * G.NRG + rg) * 2 + 0) * CNT_STRIDE;
    unsigned* cntC_L = d.cnt + (size_t)((ls * G.NRG + rg) * 2 + 1) * CNT_STRIDE;
    unsigned* cntH_A = d.cnt + (size_t)((as * G.NRG + arg) * 2 + 0) * CNT_STRIDE;
    unsigned* cntC_A = d.cnt + (size_t)((as * G.NRG + arg) * 2 + 1) * CNT_STRIDE;
    const unsigned rows_in_rg = (unsigned)min(32 * RT, B - row0);      // valid rows of the L item's row group
    const unsigned nA_per_step = rows_in_rg * (unsigned)d.CS;

    if (tid == 0) *abortw = 0;

    // ---------------------------------------------------------------- L setup: weight slice -> registers (once)
    bf16x8 W[UT][KH + KC + KPN];
    float wqf[UT * 4];
    float cst[NSLOT];
    if (hasL) {
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int i = 0; i < KH + KC + KPN; ++i) {
                const int kt = i < KH ? wave * KH + i : i < KH + KC ? H / 16 + wave * KC + (i - KH) : (H + d.E) / 16 + wave * KPN + (i - KH - KC);
                const long row = (long)(r >> 3) * H + u0 + ut * 8 + (r & 7);
                W[ut][i] = *reinterpret_cast<const bf16x8*>(LS.w16 + row * LS.ldw16 + kt * 16 + 8 * hk);
            }
        if (KIND != CHAIN_LSTM) {
#pragma unroll
            for (int m = 0; m < UT * 4; ++m) wqf[m] = LS.wq[(long)((wave & 3) * 32 + r) * H + u0 + 2 * m + hk];
        }
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int tp = 2 * sl + (tid >> 8);
            float c = 0.f;
            if (tp < NTILE && d.t0 > 0) {
                const int b = row0 + (tp / UT) * 32 + ((tid & 255) >> 3), u = u0 + (tp % UT) * 8 + (tid & 7);
                if (b < B) c = DEC ? d.att_c[ls][(long)b * H + u] : LS.c_out[((long)(d.t0 - 1) * B + b) * H + u];
            }
            cst[sl] = c;
        }
    }
    // ---------------------------------------------------------------- A setup: item constants + state -> LDS (once)
    int alen = 0;
    if (hasA) {
        for (int a = tid; a < A; a += NTH) vL[a] = AS.v[a];
        if (tid == 0) { float sv = 0.f; for (int a = 0; a < A; ++a) sv += AS.v[a]; vL[A] = sv; }     // sum(v): e = sum(v) - 2 sum_a v_a / (exp(2u_a) + 1)
        for (int j = tid; j < Tp + 4; j += NTH) {
            float ap = 0.f, cm = 0.f;
            if (j < Tin) {
                if (d.t0 > 0) {
                    ap = AS.align[((long)ab_ * d.T + (d.t0 - 1)) * Tin + j];
                    if (KIND == CHAIN_LSA) cm = AS.wcum[((long)ab_ * d.T + (d.t0 - 1)) * Tin + j];
                } else if (KIND == CHAIN_SMA && j == 0) ap = 1.f;                     // attention.py:324-328
            }
            apL[j] = ap; cumL[j] = cm;
        }
        for (int i = tid; i < Jp * (A / 4); i += NTH) {
            const int j = i / (A / 4), a4 = (i % (A / 4)) * 4;
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(AS.pm + ((long)ab_ * Tin + j) * A + a4);
            if (KIND == CHAIN_LSA) { float* pr = pmL + j * (A + 1) + a4; pr[0] = p4[0]; pr[1] = p4[1]; pr[2] = p4[2]; pr[3] = p4[3]; }
            else *reinterpret_cast<f32x4*>(pmL + j * A + a4) = p4;
        }
        for (int i = tid; i < Jm * (EC / 4); i += NTH) {
            const int j = i / (EC / 4), c4 = (i % (EC / 4)) * 4;
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(AS.memory + ((long)ab_ * Tin + j) * d.E + c0 + c4);
            bf16x4 o; o[0] = (__bf16)m4[0]; o[1] = (__bf16)m4[1]; o[2] = (__bf16)m4[2]; o[3] = (__bf16)m4[3];
            *reinterpret_cast<bf16x4*>(memL + j * EC + c4) = o;
        }
        if (KIND == CHAIN_LSA) {
            const int F = d.F, Kc = d.Kc, KP = (Kc + 7) & ~7, WP = 2 * KP + 8;
            for (int i = tid; i < F * 2 * KP; i += NTH) {      // K index of the conv product: c*KP + k (8 consecutive never straddle the two channels)
                const int f = i / (2 * KP), ck = i - f * 2 * KP, c = ck / KP, k = ck - c * KP;
                const float x = k < Kc ? AS.loc_conv[(f * 2 + c) * Kc + k] : 0.f;
                const __bf16 hi = (__bf16)x;
                convwL[f * WP + ck] = hi; convwL[(F + f) * WP + ck] = (__bf16)(x - (float)hi);
            }
            for (int i = tid; i < A * F; i += NTH) {          // x = hi + lo with both halves in bf16: three bf16 products then carry ~16 mantissa bits
                const float x = AS.loc_dense[i];
                const __bf16 hi = (__bf16)x;
                denseL[(i / F) * (F + 8) + i % F] = hi; denseL[(A + i / F) * (F + 8) + i % F] = (__bf16)(x - (float)hi);
            }
        }
        alen = AS.lengths ? AS.lengths[ab_] : Tin;
        if (d.max_pos > 0) alen = min(alen, d.max_pos);
    }
    __syncthreads();

    const float dscale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    const RngKey kh = rng_key(d.seed, LS.site_h), kc = rng_key(d.seed, LS.site_c), kn = rng_key(d.seed, AS.site_noise);

    float pre_next[NSLOT][4];
    auto load_pre = [&](int t, int tid) {
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int tp = 2 * sl + (tid >> 8);
            const int b = min(row0 + (tp / UT) * 32 + ((tid & 255) >> 3), B - 1), u = u0 + (tp % UT) * 8 + (tid & 7);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre_next[sl][g] = !(hasL && tp < NTILE) ? 0.f : DEC ? d.bias1[ls][g * H + u] + d.bias2[ls][g * H + u]      // (constant over the steps)
                                                                    : LS.pre[((long)t * B + b) * 4 * H + g * H + u];
        }
    };
    load_pre(d.t0, tid);

    // K slice of wave w: k tiles [w*KH, (w+1)*KH) of the h part, [H/16 + w*KC, ...) of the ctx part.  `step` selects the
    // exchange buffer holding h / ctx of step `step`.
    f32x16 acc[UT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ut][e] = 0.f;
    };
    auto frag_offset = [&](int step, int rt, int kt) {
        const int rtg = min(rg * RT + rt, G.MT - 1);
        return (unsigned)((step & 1) * d.NS + ls) * G.xs_bytes + (unsigned)lane * 16u + (unsigned)((kt * G.MT + rtg) * 1024);
    };
    auto gemm_part = [&](int step, int rt, int kt0, int i0, auto nk) {
        constexpr int NK = decltype(nk)::value;
        u32x4 af[NK > 0 ? NK : 1];
        const bool chk = TAG && step >= d.t0;               // (fragments of step t0 - 1: a finished launch or the zero state)
        const bool rowpad = min(rg * RT + rt, G.MT - 1) * 32 + (int)(threadIdx.x & 31) >= B;
        const unsigned long long tsp = chk ? __builtin_amdgcn_s_memrealtime() : 0ull;
        for (;;) {
            unsigned okw = 1u;
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                af[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, frag_offset(step, rt, kt0 + i), 0, SC1);
                okw &= ((af[i].x & 1u) == tag_x(step) || rowpad) ? 1u : 0u;   // (ctx units of padding rows are never written; h fragments: every row is)
            }
            if (!chk || __all(okw != 0u)) break;
            if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if ((threadIdx.x & 63) == 0) { report_abort(d.err, 17u); *abortw = 1; } break; }
        }
#pragma unroll
        for (int i = 0; i < NK; ++i)
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
                acc[ut] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), W[ut][i0 + i], acc[ut], 0, 0, 0);
    };
    // With one row tile per item (EARLY) the h fragments are requested BEFORE the wait for the contexts (h_{t-1} has long
    // been complete: the attention phase waited for it) and consumed after it, so their latency hides in that wait.
    constexpr bool EARLY = RT == 1;
    u32x4 hf[EARLY ? KH : 1];
    auto issue_h = [&](int step) {
#pragma unroll
        for (int i = 0; i < (EARLY ? KH : 1); ++i) hf[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, frag_offset(step, 0, wave * KH + i), 0, SC1);
    };
    auto mfma_h = [&](int step) {
        if (TAG && step >= d.t0) {                          // requested before the wait for the contexts: long landed, but checked like every fragment
            const unsigned long long tsp = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                unsigned okw = 1u;
#pragma unroll
                for (int i = 0; i < (EARLY ? KH : 1); ++i) okw &= ((hf[i].x & 1u) == tag_x(step)) ? 1u : 0u;
                if (__all(okw != 0u)) break;
                if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if ((threadIdx.x & 63) == 0) { report_abort(d.err, 18u); *abortw = 1; } break; }
                issue_h(step);
            }
        }
        zero_acc();
#pragma unroll
        for (int i = 0; i < (EARLY ? KH : 1); ++i)
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
                acc[ut] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hf[i]), W[ut][i], acc[ut], 0, 0, 0);
    };


    // ---------------------------------------------------------------- decode loop: decoder LSTM (D), projections (P1), prenets (P2)
    // D item: 4 hidden units (16 gate columns) x all rows, K = 2(H+E) + Hd split over the 8 waves (512 each: one source segment
    // per wave), v_mfma_f32_16x16x32_bf16; P1 item: 16 of the n_mel + 1 (gate) output columns, K = Hd + 2E; P2 item: stream s,
    // 16 of the P second-layer columns (it recomputes the whole first layer).  Weights of all three stay in registers.
    constexpr int DU = 4, DNW = 16;                             // D: units per item, k blocks of 32 per wave
    const int Hd = DEC ? d.Hd : 16, Pn = DEC ? d.P : 16, Mm = DEC ? d.M : 16, Ee = d.E;
    // (projection and prenet items sit on workgroups 128.. / 192..: with the context in two column parts those own no attention item)
    const bool hasD = DEC && wg < Hd / DU, hasP1 = DEC && wg >= 128 && wg < 128 + (Mm + 1 + 15) / 16, hasP2 = DEC && wg >= 192 && wg < 192 + d.NS * (Pn / 16);
    const int du0 = wg * DU, p1g = wg - 128, p2s = hasP2 ? (wg - 192) / (Pn / 16) : 0, p2c = hasP2 ? (wg - 192) % (Pn / 16) : 0;
    bf16x8 WD[DEC ? DNW : 1];
    bf16x8* wfragL = reinterpret_cast<bf16x8*>(smem + M.pm);      // P1: [wave][8 blocks][lane]; P2: W1 [wave][2][3][lane] then W2 [wave][lane]
    float dcst = 0.f, dbias[4] = {0.f, 0.f, 0.f, 0.f};
    auto cvt8 = [&](const float* p, int n_valid) {              // 8 consecutive floats -> bf16x8, zero past n_valid
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)(j < n_valid ? p[j] : 0.f);
        return o;
    };
    if (DEC) {
        const int j16 = lane & 15, kq = lane >> 4;
        if (hasD) {
#pragma unroll
            for (int i = 0; i < DNW; ++i)
                WD[i] = *reinterpret_cast<const bf16x8*>(d.wd16 + (long)((j16 >> 2) * Hd + du0 + (j16 & 3)) * d.ldwd + wave * (DNW * 32) + i * 32 + 8 * kq);
            if (tid < 32 * DU) {
                const int u = du0 + (tid & 3);
#pragma unroll
                for (int g = 0; g < 4; ++g) dbias[g] = d.dbias1[g * Hd + u] + d.dbias2[g * Hd + u];
                const int b = tid >> 2;
                if (d.t0 > 0 && b < B) dcst = d.dec_c[(long)b * Hd + u];
            }
        }
        if (hasP1) {
            const int col = p1g * 16 + j16, WO = Hd + d.NS * Ee;
            const float* wr = col < Mm ? d.proj_w + (long)col * WO : d.gate_w;
#pragma unroll
            for (int i = 0; i < 8; ++i) wfragL[(wave * 8 + i) * 64 + lane] = cvt8(wr + wave * 256 + i * 32 + 8 * kq, col <= Mm ? 8 : 0);
        }
        if (hasP2) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int k = i * 32 + 8 * kq;
                    wfragL[((wave * 2 + c) * 3 + i) * 64 + lane] = cvt8(d.pw1[p2s] + (long)((wave * 2 + c) * 16 + j16) * Mm + min(k, Mm - 8), k < Mm ? 8 : 0);
                }
            wfragL[(NWV * 6 + wave) * 64 + lane] = cvt8(d.pw2[p2s] + (long)(p2c * 16 + j16) * Pn + wave * 32 + 8 * kq, 8);
        }
    }
    auto rsD = __builtin_amdgcn_make_buffer_rsrc(d.XD, 0, DEC ? 2 * (Hd / 16) * 1024 : 0, 0x00020000);
    auto rsM = __builtin_amdgcn_make_buffer_rsrc(d.XM, 0, DEC ? 2 * 8 * 1024 : 0, 0x00020000);
    float* partDL = smem + M.scratch;                           // [NWV][32][20]  K-split partials of a 16-column item
    float* stgL = smem + M.scratch + NWV * 32 * 20;            // [32][20] staging of a finished 32 x 16 tile
    __bf16* h1L = reinterpret_cast<__bf16*>(smem + M.scratch + NWV * 32 * 20 + 32 * 20);   // P2: first-layer output [32][Pn + 8] bf16
    // fragment address of the 16x16x32 A operand (rows i + 16*rt16, 8 k at kofs..) inside a fragment-ordered buffer
    auto frag16 = [&](int lane, int kofs, int rt16) {         // (lane: the step's opaque copy, so that these offsets are not hoisted)
        const int j16 = lane & 15, kq = lane >> 4, k = kofs + 8 * kq;
        return (unsigned)((((k >> 4) * G.MT) * 64 + ((k >> 3) & 1) * 32 + j16 + 16 * rt16) * 16);
    };

    float sv[NSLOT][7];
    // saved activations (backward pass) and module-level outputs: plain stores, always issued right AFTER the loads the
    // step needs next (vector-memory operations complete in order: a store in front of a load delays the load)
    auto store_L_saved = [&](int t, int tid) {
        if (DEC) {                                              // inference keeps nothing but the state the next launch resumes from
            if (t == d.t1 - 1) {
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) {
                    const int tp = 2 * sl + (tid >> 8);
                    const int b = row0 + (tp / UT) * 32 + ((tid & 255) >> 3), u = u0 + (tp % UT) * 8 + (tid & 7);
                    if (tp < NTILE && b < B) d.att_c[ls][(long)b * H + u] = sv[sl][5];
                }
            }
            return;
        }
#ifndef T2_CHAIN_NOSAVE
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int tp = 2 * sl + (tid >> 8);
            const int b = row0 + (tp / UT) * 32 + ((tid & 255) >> 3), u = u0 + (tp % UT) * 8 + (tid & 7);
            if (tp < NTILE && b < B) {
                const long rb = (long)t * B + b;
                float* gp = LS.gates + rb * 4 * H + u;
                gp[0] = sv[sl][0]; gp[H] = sv[sl][1]; gp[2 * H] = sv[sl][2]; gp[3 * H] = sv[sl][3];
                LS.c_new[rb * H + u] = sv[sl][4];
                LS.c_out[rb * H + u] = sv[sl][5];
                LS.h_out[rb * LS.ldh + u] = sv[sl][6];
                LS.h16_out[rb * LS.ldh16 + u] = (__bf16)sv[sl][6];
            }
        }
#endif
    };
    auto store_A_saved = [&](int t, int tid) {                        // context of step t (still in csL)
        if (DEC) return;
#ifndef T2_CHAIN_NOSAVE
        const long rb = (long)t * B + ab_;
        for (int c = tid; c < EC; c += NTH) {
            const float x = csL[c];
            d.din[rb * d.WD + AS.coff + c0 + c] = x;
            d.dout[rb * d.WO + AS.ctx2off + c0 + c] = x;
            d.din16[rb * d.WD + AS.coff + c0 + c] = (__bf16)x;
        }
        // LSA: the step's location features (hi + lo of the bf16 pair still in LDS: the values the energies were computed from), for
        // the backward chain — which then repeats neither the location conv nor the tanh tile (chain_bwd.hip); the item's parts take
        // the rows in turn
        if (KIND == CHAIN_LSA && AS.locsave) {
            const int F = d.F, DP = F + 8;
            float* ls = AS.locsave + rb * Tin * F;
            for (int i = tid; i < Tin * (F / 4); i += NTH) {
                const int j = i / (F / 4), f4 = (i % (F / 4)) * 4;
                if (j % d.CS != part) continue;
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(lspL + j * DP + f4), lo = *reinterpret_cast<const bf16x4*>(lspL + (Tin + j) * DP + f4);
                __builtin_nontemporal_store(f32x4{(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]},
                                            reinterpret_cast<f32x4*>(ls + j * F + f4));
            }
        }
#endif
    };
#ifdef T2_STAMPS
    // (accumulators in LDS, 128 bytes in front of the dynamic carve: 32 more registers per thread made the stamped kernels spill and
    //  the stamps measure the spills)
    __shared__ __attribute__((aligned(16))) unsigned long long t2_stamp_lds[16];
    if (threadIdx.x < 16) t2_stamp_lds[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long stamp_last = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = d.t0; t < d.t1; ++t) {
        const unsigned ep = (unsigned)(t - d.t0);
        const bool more = t + 1 < d.t1;
        // An opaque copy of the thread index per step: every address that is rebuilt from it inside the step stays inside
        // the step.  Left to itself hipcc hoists dozens of per-thread 64-bit addresses out of the time loop and keeps them
        // live next to the 96 weight registers (spills).
        int tv = threadIdx.x;
        asm volatile("" : "+v"(tv));
        const unsigned xout = (unsigned)((t & 1) * d.NS) * G.xs_bytes;
        // ======================================================================================= L(t)
        if (hasL) {
            // pre-activations of this step were requested one step ago (a cold HBM read issued here would hold back every
            // exchange load of the step behind it)
            float pre[NSLOT][4];
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl)
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[sl][g] = pre_next[sl][g];
            T2_CSTAMP(15);
            if ((!EARLY || !hasA || TAGQ) && !TAG) {   // (decode loop: an attention item's workgroup has polled this counter in its A phase;
                                                        //  decoder-LSTM chain: no counter at all, the h fragments are loaded until they carry this step's tag)
                if (wave == 0 && !poll_counter(cntH_L, ep, (unsigned)G.NUG, d.err, 1u) && lane == 0) *abortw = 1;
                __syncthreads();
                if (*abortw) return;
            }
            T2_CSTAMP(0);
            if (EARLY) issue_h(t - 1);
            if (KC > 0 && !TAG) {
                if (wave == 0) {
                    bool ok = poll_counter(cntC_L, ep, nA_per_step, d.err, 2u);
                    if (ok && DEC) ok = poll_counter(d.cnt + (size_t)(10 + ls) * CNT_STRIDE, ep, (unsigned)(d.P / 16), d.err, 11u);   // prenet output of this step
                    if (!ok && lane == 0) *abortw = 1;
                }
                __syncthreads();
                if (*abortw) return;
            }
            T2_CSTAMP(2);
            if (EARLY) mfma_h(t - 1);
            // ---- fixed-order sum of the 8 K-split partial tiles, gates, cell update, dropout (model.py:340-346, 371-373)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (!EARLY) { zero_acc(); gemm_part(t - 1, rt, wave * KH, 0, std::integral_constant<int, KH>{}); }
                if (KC > 0) gemm_part(t - 1, rt, H / 16 + wave * KC, KH, std::integral_constant<int, KC>{});
                if (DEC) gemm_part(t - 1, rt, (H + d.E) / 16 + wave * KPN, KH + KC, std::integral_constant<int, KPN>{});
                T2_CSTAMP(14);
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) {
                    const int tp = rt * UT + ut;
                    if (tp > 0) __syncthreads();
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        partL[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * PPR + r] = acc[ut][e];
                    __syncthreads();
                    if ((tid >> 8) == (tp & 1)) {
                        const int sl = tp >> 1;
                        const int bl = (tid & 255) >> 3, uu = tid & 7;
                        const int b = row0 + rt * 32 + bl, u = u0 + ut * 8 + uu;
                        float g4[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float sum = 0.f;
#pragma unroll
                            for (int w = 0; w < NWV; ++w) sum += partL[(w * 32 + bl) * PPR + g * 8 + uu];
                            g4[g] = sum + pre[sl][g];
                        }
                        const float ig = fast_sigmoid(g4[0]), fg = fast_sigmoid(g4[1]), gg = fast_tanh(g4[2]), og = fast_sigmoid(g4[3]);
                        const float cn = fg * cst[sl] + ig * gg;
                        const float hn = og * fast_tanh(cn);
                        float ho = hn, co = cn;
                        if (d.drop_p > 0.f) {
                            const uint32_t idx = (uint32_t)(((long)t * B + b) * H + u);
                            ho = rng_keep(kh, idx, d.drop_p) ? hn * dscale : 0.f;
                            co = rng_keep(kc, idx, d.drop_p) ? cn * dscale : 0.f;
                        }
                        if (b >= B) { ho = 0.f; co = 0.f; }
                        cst[sl] = co;
                        sv[sl][0] = ig; sv[sl][1] = fg; sv[sl][2] = gg; sv[sl][3] = og; sv[sl][4] = cn; sv[sl][5] = co; sv[sl][6] = ho;
                        hsL[(rt * 32 + bl) * HSP + ut * 8 + uu] = ho;
                    }
                }
            }
            __syncthreads();
            T2_CSTAMP(3);
            // ---- h_t in fragment order (bf16) for the next step's GEMMs
            if (wave < RT && rg * RT + wave < G.MT) {
                const int rtg = rg * RT + wave;
                if (UT == 2 || hk == 0) {
                    const int kh8 = UT == 2 ? hk : ((u0 >> 3) & 1);           // which half of the 16-wide k tile
                    const float* hp = hsL + (wave * 32 + r) * HSP + (UT == 2 ? hk * 8 : 0);
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(hp), hi = *reinterpret_cast<const f32x4*>(hp + 4);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
                    u32x4 ow = __builtin_bit_cast(u32x4, o);
                    if (TAG) ow.x = (ow.x & ~1u) | tag_x(t);
                    __builtin_amdgcn_raw_buffer_store_b128(ow, rsX,
                        xout + (unsigned)ls * G.xs_bytes + (unsigned)((((u0 >> 4) * G.MT + rtg) * 64 + kh8 * 32 + r) * 16), 0, SC1);
                }
            }
            if (KIND != CHAIN_LSTM) {
                // ---- fp32 partial of the query projection over this unit group (attention.py:68,368): exact fma chains
                if (wave < 4 * RT) {
                    const int ct = wave & 3, rt = wave >> 2;
                    f32x16 qa;
#pragma unroll
                    for (int e = 0; e < 16; ++e) qa[e] = 0.f;
#pragma unroll
                    for (int m = 0; m < UT * 4; ++m)
                        qa = __builtin_amdgcn_mfma_f32_32x32x2f32(hsL[(rt * 32 + r) * HSP + 2 * m + hk], wqf[m], qa, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 16; ++e) qsL[(rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * (A + 4) + ct * 32 + r] = qa[e];
                }
                __syncthreads();
                const unsigned qb = (unsigned)(((ls * G.NRG + rg) * G.NUG + ug) * (32 * RT)) * (unsigned)(A * 4);
                for (int i = tv; i < RT * 32 * (A / 4); i += NTH) {
                    const int row = i / (A / 4), a4 = (i % (A / 4)) * 4;
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(qsL + row * (A + 4) + a4);
                    u32x4 vw = __builtin_bit_cast(u32x4, v4);
                    if (TAGQ) vw.x = (vw.x & ~1u) | tag_q(t);          // (absolute step: the decode loop's launches continue each other)
                    __builtin_amdgcn_raw_buffer_store_b128(vw, rsQ, qb + (unsigned)((row * A + a4) * 4), 0, SC1);
                }
            }
            T2_CSTAMP(4);
            if (TAG) { __syncthreads(); if (*abortw) return; } else publish(cntH_L, (unsigned)ug);   // (TAG: no arrival counter, the fragments carry the step's tag)
            T2_CSTAMP(5);
            // saved activations: issued here, in the slack before the next poll is answered (issuing scattered stores costs
            // the wave hundreds of cycles; behind the next phase's loads they sat on the critical path: measured +1.2 us/step)
            store_L_saved(t, tv);
            if (!DEC && !hasA && more) load_pre(t + 1, tv);
        }
        // ======================================================================================= A(t)
        if (hasA) {
            const int tid = tv, lane = tid & 63, wave = tid >> 6;
            T2_CSTAMP(6);
            if (KIND == CHAIN_LSA) {
                // ---- location features of [w_{t-1}; cum_{t-1}] (attention.py:7-23): conv (2 -> F channels, Kc taps, zero padded)
                // on the matrix cores (exact fp32 fma chains).  It needs nothing of step t, so it runs HERE, in front of the poll:
                // the workgroup would otherwise sit out the L phase's publish latency.  (The dense layer follows below, fused
                // with the energies.)
                const int F = d.F, Kc = d.Kc, pad = (Kc - 1) / 2, Tw = Tin + Kc - 1, TwP = (Tw + 8 + 3) & ~3, KP = (Kc + 7) & ~7, WP = 2 * KP + 8;
                for (int i = tid; i < 2 * TwP; i += NTH) {
                    const int c = i / TwP, j = i % TwP - pad;
                    wpadL[i] = (j >= 0 && j < Tin) ? (c == 0 ? apL[j] : cumL[j]) : 0.f;
                }
                __syncthreads();
                const int r_ = lane & 31, h_ = lane >> 5;
                const int njt = (Tin + 31) / 32, nft = (F + 31) / 32;
                // loc[j][f] = sum_{c,k} Wc[f][c][k] wpad[c][j + k]: a [Tin x 2KP] . [2KP x F] product whose A operand is read
                // straight out of the padded weights (Toeplitz).  Both operands as bf16 hi + lo (lo.lo dropped: 2^-16 relative):
                // 3 bf16 MFMAs per 16 K where the exact fp32 chain took 8 of twice the cycles; the taps' pitch (2KP + 8 halves)
                // spreads the lanes' 16-byte reads over the banks (the fp32 taps at pitch 64 put all 32 rows on one bank)
                for (int tile = wave; tile < njt * nft; tile += NWV) {
                    const int jt = tile / nft, ft = tile % nft;
                    const float* xr = wpadL + min(jt * 32 + r_, Tin - 1);
                    const __bf16* wh = convwL + min(ft * 32 + r_, F - 1) * WP + 8 * h_;
                    f32x16 acc;
#pragma unroll
                    for (int e2 = 0; e2 < 16; ++e2) acc[e2] = 0.f;
                    for (int k0 = 0; k0 < 2 * KP; k0 += 16) {
                        const int ck = k0 + 8 * h_, c = ck >= KP ? 1 : 0;
                        const float* xp = xr + c * TwP + ck - c * KP;    // (taps k >= Kc are zero; the window reads stay inside the padded row)
                        float xv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) xv[u] = xp[u];
                        bf16x8 ah, al;
#pragma unroll
                        for (int u = 0; u < 8; ++u) { ah[u] = (__bf16)xv[u]; al[u] = (__bf16)(xv[u] - (float)ah[u]); }
                        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wh + k0), bl = *reinterpret_cast<const bf16x8*>(wh + F * WP + k0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                    }
                    const int f = ft * 32 + r_;
                    if (f < F) {
#pragma unroll
                        for (int e2 = 0; e2 < 16; ++e2) {
                            const int row = jt * 32 + (e2 & 3) + 8 * (e2 >> 2) + 4 * h_;
                            if (row < Tin) {
                                const float x = acc[e2];
                                const __bf16 hi = (__bf16)x;
                                lspL[row * (F + 8) + f] = hi; lspL[(Tin + row) * (F + 8) + f] = (__bf16)(x - (float)hi);
                            }
                        }
                    }
                }
            }
            // (LSA: the last wave polls — it has no tile of the location conv above (4 tiles at Tin <= 128), so the poll's round trip
            // runs underneath the other waves' MFMA chains instead of behind them)
            if (!TAGQ && wave == (KIND == CHAIN_LSA ? NWV - 1 : 0)) {   // h_t of the item's row group, and (one request) of the L item's for step t+1
                const bool ok = (EARLY && hasL && more) ? poll_counters2(cntH_A, ep + 1, (unsigned)G.NUG, cntH_L, ep + 1, (unsigned)G.NUG, d.err, 3u)
                                                        : poll_counter(cntH_A, ep + 1, (unsigned)G.NUG, d.err, 3u);
                if (!ok && lane == 0) *abortw = 1;
            }
            if (!TAGQ || KIND == CHAIN_LSA) __syncthreads();       // (LSA: the conv's bf16 pair is complete behind this barrier)
            if (*abortw) return;
            T2_CSTAMP(7);
            // ---- query = ordered sum of the unit groups' partials.  Request order: query partials (needed now), h fragments
            // of the next step (needed in a moment), then the stores of finished work, then next step's cold pre-activations.
            {
                const int pg = tid >> 5, a4 = (tid & 31) * 4;
                const unsigned qb = (unsigned)((as * G.NRG + arg) * G.NUG) * (unsigned)(32 * RT * A * 4) + (unsigned)((arow * A + a4) * 4);
                constexpr int QU = 4;
                u32x4 pv[QU];
                // (tagged hand-off: a partial that still shows last step's tag has not landed yet: load the batch again)
                auto load_q = [&](int i0) {
                    const unsigned long long tsp = TAGQ ? __builtin_amdgcn_s_memrealtime() : 0ull;
                    for (;;) {
                        unsigned okw = 1u;
#pragma unroll
                        for (int k = 0; k < QU; ++k) {
                            pv[k] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, qb + (unsigned)min(i0 + 16 * k, G.NUG - 1) * (unsigned)(32 * RT * A * 4), 0, SC1);
                            okw &= ((pv[k].x & 1u) == tag_q(t)) ? 1u : 0u;
                        }
                        if (!TAGQ || __all(okw != 0u)) break;
                        if (__builtin_amdgcn_s_memrealtime() - tsp > SPIN_TICKS) { if ((tid & 63) == 0) { report_abort(d.err, 19u); *abortw = 1; } break; }
                    }
                };
                load_q(pg);
                f32x4 accq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < QU; ++k) if (pg + 16 * k < G.NUG) accq += __builtin_bit_cast(f32x4, pv[k]);
                for (int i0 = pg + 16 * QU; i0 < G.NUG; i0 += 16 * QU) {       // more than 64 unit groups (8 units per item)
                    load_q(i0);
#pragma unroll
                    for (int k = 0; k < QU; ++k) if (i0 + 16 * k < G.NUG) accq += __builtin_bit_cast(f32x4, pv[k]);
                }
                *reinterpret_cast<f32x4*>(redL + pg * A + a4) = accq;
                // the next step's pre-activations (cold HBM rows): requested here, where this wave needs nothing from
                // memory until the context is published (vector-memory operations complete in order)
                if (!DEC && more) load_pre(t + 1, tid);
            }
            __syncthreads();
            if (tid < A) {
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) sum += redL[k * A + tid];
                qL[tid] = sum;
                if (!DEC && part == 0) AS.qs[((long)t * B + ab_) * A + tid] = sum;
            }
            __syncthreads();
            T2_CSTAMP(8);
            if (KIND == CHAIN_LSA) {
                // ---- LSA energies e_j = v . tanh(q + pm_j + dense(loc_j)) (attention.py:20-23, 73), dense layer and tanh fused:
                // a wave takes a [32 channels x 32 positions] tile, pa^T = Wd . loc^T on the matrix cores (split-bf16 operands, below)
                // with the POSITIONS on the lanes, so each lane sums its 16 channel rows in registers; the partials of the
                // A/32 channel tiles meet in LDS ([A/32][Tp]).  pm rows are LDS-resident with an odd pitch (conflict-free
                // position-major reads).
                const int njt = (Tin + 31) / 32, nat = A / 32, TinP = (Tin + 3) & ~3;
                const int r_ = lane & 31, h_ = lane >> 5;
                constexpr float K2 = 2.0f * 1.44269504088896341f;
                for (int tile = wave; tile < njt * nat; tile += NWV) {
                    const int jt = tile % njt, at = tile / njt;
                    const int j = min(jt * 32 + r_, Tin - 1);
                    // pa^T tile = Wd . loc^T with both operands split into bf16 hi + lo (lo.lo dropped: 2^-16 relative): 3 bf16 MFMAs
                    // per 16 features where the exact fp32 chain took 8 fp32 MFMAs of twice the cycles — that chain was this phase's
                    // long pole (two waves per SIMD: 1.7 us per step on the fp32 pipe)
                    const int DP = d.F + 8;
                    const __bf16* dh = denseL + (at * 32 + r_) * DP + 8 * h_;
                    const __bf16* lh = lspL + j * DP + 8 * h_;
                    f32x16 acc;
#pragma unroll
                    for (int e2 = 0; e2 < 16; ++e2) acc[e2] = 0.f;
                    for (int k0 = 0; k0 < d.F; k0 += 16) {               // (F % 16 == 0: chain_plan)
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(dh + k0), al = *reinterpret_cast<const bf16x8*>(dh + A * DP + k0);
                        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(lh + k0), bl = *reinterpret_cast<const bf16x8*>(lh + Tin * DP + k0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                    }
                    const float* pr = pmL + j * (A + 1) + at * 32 + 4 * h_;
                    const float* qr = qL + at * 32 + 4 * h_;
                    const float* vr = vL + at * 32 + 4 * h_;
                    float sum = 0.f;
                    // u = tanh(x) = 1 - 2 / (exp(2x) + 1): the tiles of the [Tin x A] tile are saved by the item's parts in turn, for
                    // the backward chain.
                    // Layout [T][B][A][TinP] (channel-major, positions contiguous): a store instruction then covers two channel
                    // rows x 32 consecutive positions = two 128-byte lines (position-major rows put every lane on a line of its own:
                    // measured +1.7 us per step).  Every wave stores one of its tiles.
                    float* up = (!DEC && AS.usave && (tile / NWV + wave) % d.CS == part && jt * 32 + r_ < Tin)
                                    ? AS.usave + (((long)t * B + ab_) * A + at * 32 + 4 * h_) * TinP + j : nullptr;
#pragma unroll
                    for (int e2 = 0; e2 < 16; ++e2) {
                        const int ao = (e2 & 3) + 8 * (e2 >> 2);
                        const float x = qr[ao] + pr[ao] + acc[e2];
                        const float sg = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(K2 * x) + 1.0f);
                        sum += vr[ao] * sg;
                        if (up) __builtin_nontemporal_store(1.0f - 2.0f * sg, up + ao * TinP);   // (read once, by the backward pass: keep it out of the L2's way)
                    }
                    sum += __shfl_xor(sum, 32, 64);
                    if (h_ == 0 && jt * 32 + r_ < Tin) paL[at * Tp + jt * 32 + r_] = sum;
                }
                __syncthreads();
                for (int jj = tid; jj < Tin; jj += NTH) {
                    float sum = 0.f;
                    for (int at = 0; at < nat; ++at) sum += paL[at * Tp + jj];
                    eL[jj] = vL[A] - 2.0f * sum;
                }
            } else {
            // ---- energies e_j = v . tanh(q + pm_j) = sum(v) - 2 sum_a v_a / (exp(2 (q_a + pm_ja)) + 1): 16 lanes per position,
            // 8 channels per lane, two positions in flight per lane group
            {
                const int gid = tid >> 4, sub = tid & 15;
                f32x4 qv[2], vv[2];
                float vsum = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    qv[k] = *reinterpret_cast<const f32x4*>(qL + sub * 4 + 64 * k);
                    vv[k] = *reinterpret_cast<const f32x4*>(vL + sub * 4 + 64 * k);
                    qv[k] *= 2.0f * 1.44269504088896341f;                  // exp(2u) = exp2(2 log2(e) u)
                    vsum += (vv[k][0] + vv[k][1]) + (vv[k][2] + vv[k][3]);
                }
                constexpr float K2 = 2.0f * 1.44269504088896341f;
                // LDS-resident rows and L2 rows go through separate loops: one loop with a choice per row makes hipcc
                // select between the two pointers and read both through flat_load
                auto dot = [&](f32x4 (&pv)[2], int j) {
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            sum += vv[k][c] * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fmaf(pv[k][c], K2, qv[k][c])) + 1.0f);
                    return vsum - 2.0f * sum;
                };
                auto finish = [&](int j, float s0) {
                    s0 += __shfl_xor(s0, 8, 64); s0 += __shfl_xor(s0, 4, 64);
                    s0 += __shfl_xor(s0, 2, 64); s0 += __shfl_xor(s0, 1, 64);
                    if (sub == 0) eL[j] = s0;
                };
                const int nres = min(Jp, Tin);
                int j = gid;
                for (; j < nres; j += 2 * (NTH / 16)) {                  // resident rows, two positions in flight
                    const int j1 = j + NTH / 16;
                    f32x4 p0[2], p1[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        p0[k] = *reinterpret_cast<const f32x4*>(pmL + j * A + sub * 4 + 64 * k);
                        p1[k] = *reinterpret_cast<const f32x4*>(pmL + min(j1, nres - 1) * A + sub * 4 + 64 * k);
                    }
                    finish(j, dot(p0, j));
                    if (j1 < nres) finish(j1, dot(p1, min(j1, nres - 1)));
                }
                for (j = nres + gid; j < Tin; j += NTH / 16) {           // rows beyond the LDS budget: from L2
                    const float* pr = AS.pm + ((long)ab_ * Tin + j) * A + sub * 4;
                    f32x4 p0[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) p0[k] = *reinterpret_cast<const f32x4*>(pr + 64 * k);
                    finish(j, dot(p0, j));
                }
            }
            }
            __syncthreads();
            T2_CSTAMP(9);
            if (KIND == CHAIN_SMA) {
                // p = sigmoid(e + noise) ; a_t[j] = a_{t-1}[j] p_j + a_{t-1}[j-1] (1 - p_{j-1})      (attention.py:337-348)
                float maskv = AS.mask_value;
                asm volatile("" : "+v"(maskv));             // opaque: otherwise hipcc selects between &eL[j] and &mask_value and reads through flat_load
                for (int j = tid; j < Tin; j += NTH) {
                    float ev = eL[j];
                    if (j >= alen) ev = maskv;
                    if (d.noise_std > 0.f) ev += d.noise_std * rng_normal(kn, (uint32_t)(((long)t * B + ab_) * Tin + j));
                    const float p = fast_sigmoid(ev);
                    eL[j] = p;
                    if (!DEC && part == 0) AS.psel[((long)ab_ * d.T + t) * Tin + j] = p;
                }
                __syncthreads();
                for (int j = tid; j < Tin; j += NTH) {
                    float a = apL[j] * eL[j];
                    if (j > 0) a += apL[j - 1] * (1.0f - eL[j - 1]);
                    anL[j] = a;
                    if (part == 0) AS.align[((long)ab_ * d.T + t) * Tin + j] = a;
                }
            } else {
                // masked softmax over the positions, cumulative weights (attention.py:76-85, model.py:358-361)
                float maskv = AS.mask_value;
                asm volatile("" : "+v"(maskv));
                float mx = -INFINITY;
                for (int j = tid; j < Tin; j += NTH) {
                    float ev = eL[j];
                    if (j >= alen) ev = maskv;
                    eL[j] = ev;
                    mx = fmaxf(mx, ev);
                }
                mx = wave_max(mx);
                if (lane == 0) credL[wave] = mx;
                __syncthreads();
                mx = credL[0];
#pragma unroll
                for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, credL[w]);
                float sum = 0.f;
                for (int j = tid; j < Tin; j += NTH) { const float x = __expf(eL[j] - mx); eL[j] = x; sum += x; }
                sum = wave_sum(sum);
                __syncthreads();
                if (lane == 0) credL[wave] = sum;
                __syncthreads();
                sum = credL[0];
#pragma unroll
                for (int w = 1; w < NWV; ++w) sum += credL[w];
                const float inv = 1.0f / sum;
                for (int j = tid; j < Tin; j += NTH) {
                    const float w = eL[j] * inv;
                    anL[j] = w;
                    const float cm = cumL[j] + w;
                    cumL[j] = cm;
                    if (part == 0) {
                        AS.align[((long)ab_ * d.T + t) * Tin + j] = w;
                        AS.wcum[((long)ab_ * d.T + t) * Tin + j] = cm;
                    }
                }
            }
            __syncthreads();
            T2_CSTAMP(10);
            // ---- context columns [c0, c0 + EC): wave w sums positions j = w, w + 8, ...; lanes stride the columns 4 at a time;
            // resident rows (bf16 in LDS) and L2 rows (fp32) in separate loops
            {
                const int ncl = EC / 4, nres = min(Jm, Tin);
                for (int cb = 0; cb < ncl; cb += 64) {
                    const int cl = cb + lane;
                    if (cl < ncl) {
                        f32x4 accc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                        for (int j = wave; j < nres; j += NWV) {
                            const bf16x4 mb = *reinterpret_cast<const bf16x4*>(memL + j * EC + cl * 4);
                            accc += anL[j] * f32x4{(float)mb[0], (float)mb[1], (float)mb[2], (float)mb[3]};
                        }
#pragma unroll 4
                        for (int j = nres + wave; j < Tin; j += NWV)
                            accc += anL[j] * *reinterpret_cast<const f32x4*>(AS.memory + ((long)ab_ * Tin + j) * d.E + c0 + cl * 4);
                        *reinterpret_cast<f32x4*>(credL + wave * EC + cl * 4) = accc;
                    }
                }
            }
            __syncthreads();
            for (int c = tid; c < EC; c += NTH) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) sum += credL[w * EC + c];
                csL[c] = sum;
            }
            for (int j = tid; j < Tin; j += NTH) apL[j] = anL[j];          // recurrent state for the next step
            __syncthreads();
            // ---- ctx_t in fragment order (bf16): 8 columns per lane
            if (tid < EC / 8) {
                const int pc = c0 / 8 + tid;                                 // 8-column piece of the context row
                const f32x4 lo = *reinterpret_cast<const f32x4*>(csL + tid * 8), hi = *reinterpret_cast<const f32x4*>(csL + tid * 8 + 4);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
                const int kt = H / 16 + pc / 2;
                u32x4 ow = __builtin_bit_cast(u32x4, o);
                if (TAG) ow.x = (ow.x & ~1u) | tag_x(t);
                __builtin_amdgcn_raw_buffer_store_b128(ow, rsX,
                    xout + (unsigned)as * G.xs_bytes + (unsigned)(((kt * G.MT + ab_ / 32) * 64 + (pc & 1) * 32 + (ab_ & 31)) * 16), 0, SC1);
            }
            T2_CSTAMP(11);
            if (TAG) { __syncthreads(); if (*abortw) return; } else publish(cntC_A, (unsigned)(arow * d.CS + part));
            T2_CSTAMP(12);
            store_A_saved(t, tid);
        }
        if (DEC) {
            const int tid = tv, lane = tid & 63, wave = tid >> 6, r = lane & 31, hk = lane >> 5;     // rebuilt from the opaque copy
            const int j16 = lane & 15, kq = lane >> 4;
            const unsigned xcur = (unsigned)((t & 1) * d.NS) * G.xs_bytes;           // h_t / ctx_t of both streams
            unsigned* cntD = d.cnt + (size_t)8 * CNT_STRIDE;
            unsigned* cntM = d.cnt + (size_t)9 * CNT_STRIDE;
            // =================================================================================== D(t): decoder LSTM (model.py:371-373)
            T2_CSTAMP(13);
            if (hasD) {
                // K order of [W_ih | W_hh]: [h0 (H) | ctx0 (E) | h1 (H) | ctx1 (E) | dec_h (Hd)]; wave w covers k in [512 w, 512 w + 512).
                // Every wave polls the one counter ITS slice hangs on (its loads are sc1 and follow its own poll): h_t and
                // dec_h_{t-1} were published long ago, so six of the eight waves have their fragments in flight while the two
                // ctx waves still wait for the attention phase
                {
                    const int kgw = wave * (DNW * 32), sg0 = H + Ee, sg1 = 2 * (H + Ee);
                    const bool fd = kgw >= sg1;
                    const int stw = kgw < sg0 ? 0 : 1;
                    const bool isc = !fd && kgw - stw * sg0 >= H;
                    const unsigned* cp = fd ? cntD : d.cnt + (size_t)(2 * stw + (isc ? 1 : 0)) * CNT_STRIDE;
                    const unsigned wsteps = fd ? ep : ep + 1, wprod = fd ? (unsigned)(Hd / DU) : isc ? (unsigned)(B * d.CS) : (unsigned)G.NUG;
                    if (!poll_counter(cp, wsteps, wprod, d.err, 12u) && lane == 0) *abortw = 1;
                }
                T2_CSTAMP(1);
                const int kg = wave * (DNW * 32);
                const int seg0 = H + Ee, seg1 = 2 * (H + Ee);
                const bool from_d = kg >= seg1;
                const int st = kg < seg0 ? 0 : 1;
                const int lk = from_d ? kg - seg1 : kg - st * seg0;                  // k inside the source buffer ([h | ctx] of a stream)
                const unsigned base = from_d ? (unsigned)(((t + 1) & 1) * (Hd / 16) * 1024) : xcur + (unsigned)st * G.xs_bytes;
                f32x4 accd[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int i0 = 0; i0 < DNW; i0 += 4) {
                    u32x4 af[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            af[i][m] = from_d ? __builtin_amdgcn_raw_buffer_load_b128(rsD, base + frag16(lane, lk + (i0 + i) * 32, m), 0, SC1)
                                              : __builtin_amdgcn_raw_buffer_load_b128(rsX, base + frag16(lane, lk + (i0 + i) * 32, m), 0, SC1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            accd[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i][m]), WD[i0 + i], accd[m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) partDL[(wave * 32 + m * 16 + kq * 4 + e) * 20 + j16] = accd[m][e];
                __syncthreads();
                if (*abortw) return;
                if (tid < 32 * DU) {
                    const int row = tid >> 2, uu = tid & 3;
                    float g4[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < NWV; ++w) sum += partDL[(w * 32 + row) * 20 + g * 4 + uu];
                        g4[g] = sum + dbias[g];
                    }
                    const float ig = fast_sigmoid(g4[0]), fg = fast_sigmoid(g4[1]), gg = fast_tanh(g4[2]), og = fast_sigmoid(g4[3]);
                    const float cn = fg * dcst + ig * gg;
                    dcst = cn;                                                       // (inference: no dropout on the carried state)
                    stgL[row * 20 + uu] = row < B ? og * fast_tanh(cn) : 0.f;
                    if (t == d.t1 - 1 && row < B) d.dec_c[(long)row * Hd + du0 + uu] = cn;
                }
                __syncthreads();
                if (tid < 32) {              // dec_h_t: 4 units of a row = 8 bytes of the row's fragment piece
                    const f32x4 h4 = *reinterpret_cast<const f32x4*>(stgL + tid * 20);
                    bf16x4 o; o[0] = (__bf16)h4[0]; o[1] = (__bf16)h4[1]; o[2] = (__bf16)h4[2]; o[3] = (__bf16)h4[3];
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsD,
                        (unsigned)((t & 1) * (Hd / 16) * 1024) + (unsigned)((((du0 >> 4) * 64) + ((du0 >> 3) & 1) * 32 + tid) * 16 + ((du0 & 7) >> 2) * 8), 0, SC1);
                }
                T2_CSTAMP(6);
                publish(cntD, (unsigned)wg);
            }
            // =================================================================================== P1(t): mel / gate projections + stop rule (model.py:382-388, 461-480)
            if (hasP1) {
                // K order of the projections: [dec_h (Hd) | ctx0 (E) | ctx1 (E)]; wave w covers k in [256 w, 256 w + 256); as in
                // D(t) every wave polls the counter of its own slice, so the ctx waves load while dec_h_t is still being made
                const int kg = wave * 256;
                const bool from_d = kg < Hd;
                const int st = kg < Hd + Ee ? 0 : 1;
                if (!poll_counter(from_d ? cntD : d.cnt + (size_t)(2 * st + 1) * CNT_STRIDE,
                                  ep + 1, from_d ? (unsigned)(Hd / DU) : (unsigned)(B * d.CS), d.err, 13u) && lane == 0) *abortw = 1;
                const int lk = from_d ? kg : H + (kg - Hd - st * Ee);                // ctx sits behind h in a stream's fragments
                const unsigned base = from_d ? (unsigned)((t & 1) * (Hd / 16) * 1024) : xcur + (unsigned)st * G.xs_bytes;
                f32x4 accp[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int i0 = 0; i0 < 8; i0 += 4) {
                    u32x4 af[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            af[i][m] = from_d ? __builtin_amdgcn_raw_buffer_load_b128(rsD, base + frag16(lane, lk + (i0 + i) * 32, m), 0, SC1)
                                              : __builtin_amdgcn_raw_buffer_load_b128(rsX, base + frag16(lane, lk + (i0 + i) * 32, m), 0, SC1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bf16x8 wb = wfragL[(wave * 8 + i0 + i) * 64 + lane];
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            accp[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i][m]), wb, accp[m], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) partDL[(wave * 32 + m * 16 + kq * 4 + e) * 20 + j16] = accp[m][e];
                __syncthreads();
                if (*abortw) return;
                {
                    const int row = tid >> 4, col = p1g * 16 + (tid & 15);
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < NWV; ++w) sum += partDL[(w * 32 + row) * 20 + (tid & 15)];
                    float val = 0.f;
                    if (col < Mm) {
                        val = sum + d.proj_b[col];
                        if (row < B) d.mel_out[(long)row * d.ldmel + (long)t * Mm + col] = val;
                    } else if (col == Mm && row < B) {
                        const float gt = sum + d.gate_b[0];
                        d.gate_out[(long)row * d.ldgate + t] = gt;
                        if (d.stop_index && d.stop_index[row] < 0 && 1.0f / (1.0f + expf(-gt)) > d.thr) { d.stop_index[row] = t; atomicAdd(d.done, 1); }
                    }
                    stgL[row * 20 + (tid & 15)] = (col < Mm && row < B) ? val : 0.f;            // next step's prenet input (teacher = own output)
                }
                __syncthreads();
                if (wave == 0) {             // mel_t columns [16 p1g, +16) as one bf16 fragment (k tile p1g of the prenets' K)
                    const float* hp = stgL + r * 20 + hk * 8;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(hp), hi = *reinterpret_cast<const f32x4*>(hp + 4);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsM, (unsigned)(((t & 1) * 8 + p1g) * 1024 + lane * 16), 0, SC1);
                }
                publish(cntM, (unsigned)p1g);
            }
            // =================================================================================== P2(t): both prenet layers of frame t+1 (model.py:13-24, 470-471)
            if (hasP2 && t + 1 < d.T) {
                if (wave == 0 && !poll_counter(cntM, ep + 1, (unsigned)((Mm + 1 + 15) / 16), d.err, 14u) && lane == 0) *abortw = 1;
                __syncthreads();
                if (*abortw) return;
                const RngKey k1 = rng_key(d.seed, d.psite1[p2s]), k2 = rng_key(d.seed, d.psite2[p2s]);
                const float pscale = d.pdrop > 0.f ? 1.0f / (1.0f - d.pdrop) : 1.0f;
                const uint32_t ibase = (uint32_t)((long)(t + 1) * B * Pn);             // keep-bit index = (t+1)*B*P + b*P + n, as infer.hip
                // layer 1: [32 x M] . W1^T -> 16 columns x 2 per wave; ReLU + dropout; bf16 rows in LDS
                u32x4 mf[3][2];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        mf[i][m] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (unsigned)((t & 1) * 8 * 1024) + frag16(lane, i * 32, m), 0, SC1);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 3; ++i) a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, mf[i][m]), wfragL[((wave * 2 + c) * 3 + i) * 64 + lane], a1, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = m * 16 + kq * 4 + e, n = (wave * 2 + c) * 16 + j16;
                            float v = fmaxf(a1[e], 0.f);
                            if (d.pdrop > 0.f) v = rng_keep(k1, ibase + (uint32_t)(row * Pn + n), d.pdrop) ? v * pscale : 0.f;
                            h1L[row * (Pn + 8) + n] = (__bf16)v;
                        }
                    }
                __syncthreads();
                // layer 2: this item's 16 columns, wave w takes k block w of the P inputs
                f32x4 a2[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const bf16x8 hv = *reinterpret_cast<const bf16x8*>(h1L + (m * 16 + j16) * (Pn + 8) + wave * 32 + 8 * kq);
                    a2[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hv, wfragL[(NWV * 6 + wave) * 64 + lane], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) partDL[(wave * 32 + m * 16 + kq * 4 + e) * 20 + j16] = a2[m][e];
                }
                __syncthreads();
                {
                    const int row = tid >> 4, n = p2c * 16 + (tid & 15);
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < Pn / 32; ++w) sum += partDL[(w * 32 + row) * 20 + (tid & 15)];
                    float v = fmaxf(sum, 0.f);
                    if (d.pdrop > 0.f) v = rng_keep(k2, ibase + (uint32_t)(row * Pn + n), d.pdrop) ? v * pscale : 0.f;
                    stgL[row * 20 + (tid & 15)] = row < B ? v : 0.f;
                }
                __syncthreads();
                if (wave == 0) {             // p2(t+1) columns [16 p2c, +16): one fragment behind h and ctx in the stream's exchange buffer
                    const float* hp = stgL + r * 20 + hk * 8;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(hp), hi = *reinterpret_cast<const f32x4*>(hp + 4);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsX,
                        xcur + (unsigned)p2s * G.xs_bytes + (unsigned)((((H + Ee) / 16 + p2c) * G.MT) * 1024 + lane * 16), 0, SC1);
                }
                publish(d.cnt + (size_t)(10 + p2s) * CNT_STRIDE, (unsigned)p2c);
            }
        }
    }
#ifdef T2_STAMPS
    if (tid == 0 && (KIND != CHAIN_LSTM) == (T2_STAMPS != 2))     // -DT2_STAMPS=1: the attention chain, =2: the decoder-LSTM chain
        for (int i = 0; i < 16; ++i) t2_chain_stamps[wg * 16 + i] += t2_stamp_lds[i];
#endif
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int g_chain_cus[16] = {0};

int chain_device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int& c = g_chain_cus[dev & 15];
    if (c == 0) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        c = p.multiProcessorCount;
    }
    return c;
}

// Picks the tiling for a shape; returns false when the persistent kernel does not cover it.
bool chain_plan(ChainDesc& d) {
    if (d.H != 1024 || d.B < 1 || d.B > 128) return false;
    if (d.kind != CHAIN_LSTM && (d.E != 512 || d.A != 128 || d.NS < 1 || d.NS > 2)) return false;
    if (d.kind == CHAIN_LSA && (d.F + 1 > 64 || d.F % 16 != 0 || d.A % 32 != 0)) return false;
    const int MT = (d.B + 31) / 32;
    if (d.dec && (MT != 1 || d.NS != 2 || d.kind == CHAIN_LSTM || d.P != 256 || d.Hd != 1024 || d.M + 1 > 96 || d.M % 8 != 0)) return false;
    // decoder-LSTM chain: one row tile per item while the items still fit the chip (B <= 64: 128 unit groups x 2 row groups = 256
    // workgroups; with both tiles in one item half the CUs idled and every step did twice the work per workgroup)
    if (d.kind == CHAIN_LSTM) { d.UT = 1; d.RT = (d.H / 8) * MT <= 256 ? 1 : 2; d.CS = 1; d.NS = 1; }
    else {
        d.RT = MT <= 2 ? 1 : 2;
        d.UT = (d.NS * (d.H / 8) * ((MT + d.RT - 1) / d.RT) <= 256) ? 1 : 2;
        d.CS = (!d.dec && d.NS * d.B * 4 <= 256) ? 4 : d.NS * d.B * 2 <= 256 ? 2 : 1;      // (decode: 2, see the D / P items)
    }
    const Geo g = geo_of(d, d.UT, d.RT);
    if (g.nL > 256 || g.nA > 256) return false;
    if (chain_device_cus() < 256 || !chain_device_claim()) return false;
    // LDS residency: processed-memory rows first, then (bf16) memory rows, in what the largest stream leaves free
    const int budget = (160 * 1024 - 256) / 4;
    int tmax = 4;
    for (int s = 0; s < d.NS && d.kind != CHAIN_LSTM; ++s) tmax = std::max(tmax, d.st[s].Tin);
    d.lds_Tin = tmax; d.lds_Jp = 0; d.lds_Jm = 0;
    const int fixed = lds_of(d, d.UT, d.RT, tmax, 0, 0).total;
    if (fixed > budget) return false;
    if (d.kind != CHAIN_LSTM) {
        const int EC = d.E / d.CS;
        int left = budget - fixed + (d.dec ? 16384 : 0);                        // (dec: `fixed` holds the 64 KB pad the rows will take the place of)
        const int pmp = d.kind == CHAIN_LSA ? d.A + 1 : d.A;
        d.lds_Jp = std::min(tmax, left / pmp); left -= d.lds_Jp * pmp;
        if (d.kind == CHAIN_LSA && (d.lds_Jp < tmax || (d.dec && g.nA > 128))) return false;   // LSA energies read pm from LDS only
        d.lds_Jm = std::min(tmax, left / (EC / 2)) & ~1;
        for (int s = 0; s < d.NS; ++s) { d.Jp[s] = std::min(d.st[s].Tin, d.lds_Jp); d.Jm[s] = std::min(d.st[s].Tin, d.lds_Jm); }
        if (getenv("T2_CHAIN_NO_RESIDENT")) { if (d.kind != CHAIN_LSA) d.Jp[0] = d.Jp[1] = 0; d.Jm[0] = d.Jm[1] = 0; }
        if (lds_of(d, d.UT, d.RT, tmax, d.lds_Jp, d.lds_Jm).total > budget) return false;
    }
    return true;
}

size_t chain_exchange_bytes(const ChainDesc& d, size_t* x_bytes, size_t* q_bytes) {
    const Geo g = geo_of(d, d.UT, d.RT);
    *x_bytes = (size_t)2 * d.NS * g.xs_bytes;
    *q_bytes = d.kind == CHAIN_LSTM ? 16 : (size_t)d.NS * g.NRG * g.NUG * 32 * d.RT * d.A * 4;
    return *x_bytes + *q_bytes;
}

template <int UT, int RT, int KC, int KIND, int KPN = 0>
static int chain_launch(const ChainDesc& d, hipStream_t s) {
    const Lds m = lds_of(d, UT, RT, d.lds_Tin, d.lds_Jp, d.lds_Jm);
    const size_t smem = (size_t)m.total * sizeof(float);
    auto kernel = chain_fwd_kernel<UT, RT, 8, KC, KIND, KPN>;
    const Geo g = geo_of(d, UT, RT);
    const int grid = KPN > 0 ? 256 : std::max(g.nL, g.nA);
    T2_TRY_RC(persistent_prepare(kernel, grid, smem));
    T2_CHECK_HIP(hipMemsetAsync(d.cnt, 0, KPN > 0 ? kChainCntBytes : (size_t)d.NS * g.NRG * 2 * CNT_STRIDE * sizeof(unsigned), s));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(NTH), smem, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

int chain_fwd(const ChainDesc& d, hipStream_t s) {
    T2_REQUIRE(d.t1 > d.t0 && d.t0 >= 0, "chain_fwd: bad step range [%d,%d)", d.t0, d.t1);
    T2_REQUIRE(d.X && d.cnt && d.err && (d.kind == CHAIN_LSTM || d.Q), "chain_fwd: exchange buffers missing");
    if (d.kind == CHAIN_LSTM) {
        if (d.RT == 1) return chain_launch<1, 1, 0, CHAIN_LSTM>(d, s);
        return chain_launch<1, 2, 0, CHAIN_LSTM>(d, s);
    }
    T2_REQUIRE(d.kind == CHAIN_SMA || d.kind == CHAIN_LSA, "chain_fwd: attention kind %d not covered", d.kind);
    if (d.dec) {
        T2_REQUIRE(d.UT == 1 && d.RT == 1 && d.XD && d.XM, "chain_fwd: decode loop needs one row tile and its exchange buffers");
        return d.kind == CHAIN_LSA ? chain_launch<1, 1, 4, CHAIN_LSA, 2>(d, s) : chain_launch<1, 1, 4, CHAIN_SMA, 2>(d, s);
    }
    if (d.kind == CHAIN_LSA) {
        if (d.UT == 1 && d.RT == 1) return chain_launch<1, 1, 4, CHAIN_LSA>(d, s);
        if (d.UT == 2 && d.RT == 1) return chain_launch<2, 1, 4, CHAIN_LSA>(d, s);
        if (d.UT == 2 && d.RT == 2) return chain_launch<2, 2, 4, CHAIN_LSA>(d, s);
    }
    if (d.UT == 1 && d.RT == 1) return chain_launch<1, 1, 4, CHAIN_SMA>(d, s);
    if (d.UT == 2 && d.RT == 1) return chain_launch<2, 1, 4, CHAIN_SMA>(d, s);
    if (d.UT == 2 && d.RT == 2) return chain_launch<2, 2, 4, CHAIN_SMA>(d, s);
    T2_REQUIRE(false, "chain_fwd: tiling UT=%d RT=%d not instantiated", d.UT, d.RT);
    return -1;
}

}  // namespace t2
