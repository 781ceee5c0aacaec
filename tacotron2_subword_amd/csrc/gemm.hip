// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, so the
// result is an ordinary k-ordered fp32 dot product — this is the path the 1e-4 parity contract
// is stated on).  C = act(alpha * A.B + bias) [dropout] + beta * C.
//
// Layout: 256 threads = 4 waves (2x2); block tile BM x BN, BK = 16; each wave owns a
// (BM/2)x(BN/2) sub-tile as (BM/64)x(BN/64) MFMA tiles of 32x32.  Both operands are staged
// k-major in LDS ([BK][BM+4]) so that a fragment read is 32 consecutive floats per half-wave
// (conflict-free ds_read_b32) whichever of the two source strides is the contiguous one.
// Global loads are 16 B per lane along the contiguous stride; the next K-chunk is prefetched
// into registers while the current one feeds the MFMAs (one barrier per chunk).
#include "kernels.h"
#include <cstdlib>
#include <algorithm>
#include <type_traits>

namespace t2 {

static int g_precision = 0;       // 0: fp32 operands (parity path), 1: bf16 operands for large GEMMs
// bf16 mode: stage fp32 operands as bf16 copies when both extents reach g_stage_min (T2_GEMM_STAGE=0 switches it off,
// T2_GEMM_STAGE_MIN overrides the extent)
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
static int g_stage = env_int("T2_GEMM_STAGE", 1);
void set_gemm_staging(int on) { g_stage = on != 0; }
static const int g_stage_min = env_int("T2_GEMM_STAGE_MIN", 256);
void set_precision(int p) { g_precision = p; }
int get_precision() { return g_precision; }

namespace {

constexpr int BK = 16;

struct GemmK {
    GemmDesc d;
    int kchunks;    // K-chunks (of BK) per split
    int avec, bvec;  // 16-byte loads legal for A / B
    int xcd_swizzle; // bf16-source kernel: XCD-aware tile order (tile count a multiple of 8)
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ void epilogue_store(const GemmDesc& d, float* C, int m, int n, float acc, RngKey key) {
    float v = d.alpha * acc;
    if (d.bias1) v += d.bias1[n];
    if (d.bias2) v += d.bias2[n];
    v = apply_act(v, d.act);
    if (d.drop_p > 0.f) {
        uint32_t idx = d.drop_base + (uint32_t)m * d.drop_mstride + (uint32_t)n;
        v = rng_keep(key, idx, d.drop_p) ? v * (1.0f / (1.0f - d.drop_p)) : 0.f;
    }
    const long mo = d.crow_mod ? (long)(m % d.crow_mod) * d.crow_mul + m / d.crow_mod : (long)m;
    float* p = C + mo * d.ldc + n;
    if (d.beta != 0.f) v += d.beta * (*p);
    *p = v;
}

// Implicit im2col for a k-tap 1-D convolution over channels-last frames X[B*T, C]: the operand is
// the virtual matrix V[m, dk*C + ci] = X[m + dk - pad, ci] if the shifted frame stays inside the
// same utterance (0 <= m%T + dk - pad < T), else 0.  T == 0: ordinary strided operand.
struct ConvAddr { int T, C, pad; };

// Load one operand tile (R rows x BK) into registers.  KC: k is the contiguous stride.
template <int R, bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, long srow, long sk, int row0, int nrows,
                                          int k0, int kend, int vec, ConvAddr cv, f32x4 (&regs)[R * 4 / 256]) {
    constexpr int NQ = R * 4 / 256;
    if (!cv.T && vec && row0 + R <= nrows && k0 + BK <= kend) {          // interior tile: unconditional 16-byte loads
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = threadIdx.x + i * 256;
            const float* p = KC ? base + (long)(row0 + (q >> 2)) * srow + k0 + (q & 3) * 4
                                : base + (long)(k0 + q / (R / 4)) * sk + row0 + (q % (R / 4)) * 4;
            regs[i] = *reinterpret_cast<const f32x4*>(p);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int q = threadIdx.x + i * 256;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            int row = row0 + (q >> 2), k = k0 + (q & 3) * 4;
            if (row < nrows) {
                if (cv.T) {                                  // rows = frames m, k = dk*C + ci  (C % 4 == 0)
                    if (k < kend) {
                        const int dk = k / cv.C, ci = k - dk * cv.C, t = row % cv.T + dk - cv.pad;
                        if (t >= 0 && t < cv.T) v = *reinterpret_cast<const f32x4*>(base + (long)(row + dk - cv.pad) * cv.C + ci);
                    }
                } else {
                    const float* p = base + (long)row * srow + k;
                    if (vec && k + 3 < kend) {
                        v = *reinterpret_cast<const f32x4*>(p);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (k + j < kend) v[j] = p[j];
                    }
                }
            }
        } else {
            int k = k0 + q / (R / 4), row = row0 + (q % (R / 4)) * 4;
            if (k < kend) {
                if (cv.T) {                                  // k = frames m, rows = dk*C + ci
                    if (row < nrows) {
                        const int dk = row / cv.C, ci = row - dk * cv.C, t = k % cv.T + dk - cv.pad;
                        if (t >= 0 && t < cv.T) v = *reinterpret_cast<const f32x4*>(base + (long)(k + dk - cv.pad) * cv.C + ci);
                    }
                } else {
                    const float* p = base + (long)k * sk + row;
                    if (vec && row + 3 < nrows) {
                        v = *reinterpret_cast<const f32x4*>(p);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (row + j < nrows) v[j] = p[j];
                    }
                }
            }
        }
        regs[i] = v;
    }
}

template <int R, bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const f32x4 (&regs)[R * 4 / 256]) {
    constexpr int NQ = R * 4 / 256;
    constexpr int P = R + 4;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int q = threadIdx.x + i * 256;
        if (KC) {
            int row = q >> 2, k = (q & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) lds[(k + j) * P + row] = regs[i][j];
        } else {
            int k = q / (R / 4), row = (q % (R / 4)) * 4;
            *reinterpret_cast<f32x4*>(&lds[k * P + row]) = regs[i];
        }
    }
}

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmK g) {
    const GemmDesc& d = g.d;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PA = BM + 4, PB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[2][BK * PA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * PB];

    const int z = blockIdx.z;
    const int split = z % d.splitk, bz = z / d.splitk;
    const float* A = d.A + (long)bz * d.bsA;
    const float* B = d.B + (long)bz * d.bsB;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = split * g.kchunks * BK;
    const int kend = min(d.K, kbeg + g.kchunks * BK);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hk = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const ConvAddr cva{d.conv_a ? d.conv_T : 0, d.conv_C, d.conv_pad}, cvb{d.conv_b ? d.conv_T : 0, d.conv_C, d.conv_pad};
    f32x4 ra[BM * 4 / 256], rb[BN * 4 / 256];
    if (kbeg < kend) {
        load_tile<BM, A_KC>(A, d.sam, d.sak, m0, d.M, kbeg, kend, g.avec, cva, ra);
        load_tile<BN, B_KC>(B, d.sbn, d.sbk, n0, d.N, kbeg, kend, g.bvec, cvb, rb);
        store_tile<BM, A_KC>(As[0], ra);
        store_tile<BN, B_KC>(Bs[0], rb);
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {
            load_tile<BM, A_KC>(A, d.sam, d.sak, m0, d.M, k0 + BK, kend, g.avec, cva, ra);
            load_tile<BN, B_KC>(B, d.sbn, d.sbk, n0, d.N, k0 + BK, kend, g.bvec, cvb, rb);
        }
        const float* as = As[cur] + wm * (BM / 2) + r;
        const float* bs = Bs[cur] + wn * (BN / 2) + r;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[(2 * kk + hk) * PA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[(2 * kk + hk) * PB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            store_tile<BM, A_KC>(As[cur ^ 1], ra);
            store_tile<BN, B_KC>(Bs[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: lane holds column n = r, rows (e&3) + 8*(e>>2) + 4*hk of each 32x32 tile
    const RngKey key = rng_key(d.seed, d.site);
    float* C = d.C + (long)bz * d.bsC;
    float* ws = d.splitk > 1 ? d.ws + ((long)split * d.batch + bz) * (long)d.M * d.N : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + r;
            if (n >= d.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk;
                if (m >= d.M) continue;
                if (ws) ws[(long)m * d.N + n] = acc[i][j][e];
                else epilogue_store(d, C, m, n, acc[i][j][e], key);
            }
        }
}

__global__ void splitk_reduce_kernel(GemmDesc d) {
    const long total = (long)d.batch * d.M * d.N;
    const RngKey key = rng_key(d.seed, d.site);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float acc = 0.f;                                   // partials requested 8 at a time, added in split order
        int s = 0;
        for (; s + 8 <= d.splitk; s += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = d.ws[(long)(s + j) * total + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        if (s + 4 <= d.splitk) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = d.ws[(long)(s + j) * total + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[j];
            s += 4;
        }
        for (; s < d.splitk; ++s) acc += d.ws[(long)s * total + i];
        const int bz = (int)(i / ((long)d.M * d.N));
        const long rem = i - (long)bz * d.M * d.N;
        const int m = (int)(rem / d.N), n = (int)(rem % d.N);
        epilogue_store(d, d.C + (long)bz * d.bsC, m, n, acc, key);
    }
}


// ---------------------------------------------------------------------------------------------
// bf16-operand variant (fp32 in HBM, converted while staging; fp32 accumulate):
// v_mfma_f32_32x32x16_bf16, tile 128x128x64, LDS tiles row-major [row][k] bf16 with an XOR
// swizzle of the 16-byte slots (swz16), double buffered, next chunk prefetched into registers.  16x the matrix rate of the fp32 path, so
// this kernel is bound by operand delivery (each thread moves 64 B of fp32 per MFMA).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BK16 = 64, PK16 = BK16;
// 16-byte slot s (8 bf16) of row R lives at slot s ^ ((R ^ (R >> 2)) & 7): with unpadded 128-byte rows this
// makes the K-contiguous stores, the row-contiguous (transposing) stores and the ds_read_b128 fragment
// reads all bank-conflict-free (checked exhaustively over the lane groups of MI355X_MICROARCH.md §LDS).
__device__ __forceinline__ int swz16(int row, int slot) { return row * PK16 + ((slot ^ ((row ^ (row >> 2)) & 7)) << 3); }

__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (__bf16)lo[j]; v[4 + j] = (__bf16)hi[j]; }
    return v;
}

// 128 rows x 64 k, k contiguous in the source: 4 tasks (row, 8 k) per thread.
__device__ __forceinline__ void load16_kc(const float* __restrict__ base, long srow, int row0, int nrows, int k0, int kend, int vec,
                                          ConvAddr cv, bf16x8 (&regs)[4]) {
    if (cv.T && row0 + 128 <= nrows && k0 + BK16 <= kend) {              // implicit-conv interior tile: select, no branches
        f32x4 lo[4], hi[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = threadIdx.x + i * 256, row = row0 + (q >> 3), k = k0 + (q & 7) * 8;
            const int dk = k / cv.C, ci = k - dk * cv.C, t = row % cv.T + dk - cv.pad;
            ok[i] = t >= 0 && t < cv.T;
            const float* p = base + (ok[i] ? (long)(row + dk - cv.pad) * cv.C + ci : 0l);
            lo[i] = *reinterpret_cast<const f32x4*>(p); hi[i] = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
            regs[i] = pack8(ok[i] ? lo[i] : zz, ok[i] ? hi[i] : zz);
        }
        return;
    }
    if (!cv.T && vec && row0 + 128 <= nrows && k0 + BK16 <= kend) {     // interior tile: unconditional 16-byte loads
        f32x4 lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = threadIdx.x + i * 256;
            const float* p = base + (long)(row0 + (q >> 3)) * srow + k0 + (q & 7) * 8;
            lo[i] = *reinterpret_cast<const f32x4*>(p); hi[i] = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) regs[i] = pack8(lo[i], hi[i]);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + i * 256, row = row0 + (q >> 3), k = k0 + (q & 7) * 8;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (row < nrows && k < kend) {
            if (cv.T) {                                      // C % 8 == 0: the 8 k's share one tap
                const int dk = k / cv.C, ci = k - dk * cv.C, t = row % cv.T + dk - cv.pad;
                if (t >= 0 && t < cv.T) {
                    const float* p = base + (long)(row + dk - cv.pad) * cv.C + ci;
                    lo = *reinterpret_cast<const f32x4*>(p); hi = *reinterpret_cast<const f32x4*>(p + 4);
                }
            } else {
                const float* p = base + (long)row * srow + k;
                if (vec && k + 7 < kend) {
                    lo = *reinterpret_cast<const f32x4*>(p); hi = *reinterpret_cast<const f32x4*>(p + 4);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { if (k + j < kend) lo[j] = p[j]; if (k + 4 + j < kend) hi[j] = p[4 + j]; }
                }
            }
        }
        regs[i] = pack8(lo, hi);
    }
}
__device__ __forceinline__ void store16_kc(__bf16* __restrict__ lds, const bf16x8 (&regs)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + i * 256;
        *reinterpret_cast<bf16x8*>(lds + swz16(q >> 3, q & 7)) = regs[i];
    }
}
// 64 k-rows x 128 "rows" (m or n), rows contiguous in the source: one task (4 rows, 8 k) per thread.
__device__ __forceinline__ void load16_mc(const float* __restrict__ base, long sk, int row0, int nrows, int k0, int kend, int vec,
                                          ConvAddr cv, bf16x8 (&regs)[4]) {
    const int row = row0 + (threadIdx.x & 31) * 4, kb = k0 + (threadIdx.x >> 5) * 8;
    f32x4 v[8];
    if (!cv.T && vec && row0 + 128 <= nrows && k0 + BK16 <= kend) {     // interior tile: 8 unconditional 16-byte loads
        const float* p = base + (long)kb * sk + row;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) v[kk] = *reinterpret_cast<const f32x4*>(p + (long)kk * sk);
    } else if (cv.T && row0 + 128 <= nrows && k0 + BK16 <= kend) {       // implicit-conv interior tile: select, no branches
        const int dk = row / cv.C, ci = row - dk * cv.C;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = kb + kk, t = k % cv.T + dk - cv.pad;
            const bool ok = t >= 0 && t < cv.T;
            const f32x4 x = *reinterpret_cast<const f32x4*>(base + (ok ? (long)(k + dk - cv.pad) * cv.C + ci : 0l));
            v[kk] = ok ? x : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = kb + kk;
        v[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < kend && row < nrows) {
            if (cv.T) {
                const int dk = row / cv.C, ci = row - dk * cv.C, t = k % cv.T + dk - cv.pad;
                if (t >= 0 && t < cv.T) v[kk] = *reinterpret_cast<const f32x4*>(base + (long)(k + dk - cv.pad) * cv.C + ci);
            } else {
                const float* p = base + (long)k * sk + row;
                if (vec && row + 3 < nrows) v[kk] = *reinterpret_cast<const f32x4*>(p);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (row + j < nrows) v[kk][j] = p[j];
                }
            }
        }
    }
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
        bf16x8 o;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) o[kk] = (__bf16)v[kk][mm];
        regs[mm] = o;
    }
}
__device__ __forceinline__ void store16_mc(__bf16* __restrict__ lds, const bf16x8 (&regs)[4]) {
    const int row = (threadIdx.x & 31) * 4, ks = threadIdx.x >> 5;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) *reinterpret_cast<bf16x8*>(lds + swz16(row + mm, ks)) = regs[mm];
}

// ---- two-phase loaders of an INTERIOR tile (all 128 rows and all 64 k in range, 16-byte aligned): `issue` only requests
// the fp32 data into registers, `finish` converts and stores to LDS.  Keeping the raw registers alive lets the main loop
// hold TWO k-steps in flight per wave: measured on the old loop the wait + convert sat directly behind the loads, in
// front of the MFMAs (one k-step of 64 KB in flight per workgroup = the ~50 GB/s per CU that bounded the kernel).
struct RawKC { f32x4 lo[4], hi[4]; unsigned ok; };
struct RawMC { f32x4 v[8]; unsigned ok; };
__device__ __forceinline__ void issue_kc(const float* __restrict__ base, long srow, int row0, int k0, ConvAddr cv, RawKC& r) {
    r.ok = 0xFu;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + i * 256, row = row0 + (q >> 3), k = k0 + (q & 7) * 8;
        const float* p;
        if (cv.T) {
            const int dk = k / cv.C, ci = k - dk * cv.C, t = row % cv.T + dk - cv.pad;
            const bool ok = t >= 0 && t < cv.T;
            if (!ok) r.ok &= ~(1u << i);
            p = base + (ok ? (long)(row + dk - cv.pad) * cv.C + ci : 0l);
        } else {
            p = base + (long)row * srow + k;
        }
        r.lo[i] = *reinterpret_cast<const f32x4*>(p); r.hi[i] = *reinterpret_cast<const f32x4*>(p + 4);
    }
}
__device__ __forceinline__ void finish_kc(__bf16* __restrict__ lds, const RawKC& r) {
    const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + i * 256;
        const bool ok = (r.ok >> i) & 1u;
        *reinterpret_cast<bf16x8*>(lds + swz16(q >> 3, q & 7)) = pack8(ok ? r.lo[i] : zz, ok ? r.hi[i] : zz);
    }
}
__device__ __forceinline__ void issue_mc(const float* __restrict__ base, long sk, int row0, int k0, ConvAddr cv, RawMC& r) {
    const int row = row0 + (threadIdx.x & 31) * 4, kb = k0 + (threadIdx.x >> 5) * 8;
    r.ok = 0xFFu;
    if (cv.T) {
        const int dk = row / cv.C, ci = row - dk * cv.C;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = kb + kk, t = k % cv.T + dk - cv.pad;
            const bool ok = t >= 0 && t < cv.T;
            if (!ok) r.ok &= ~(1u << kk);
            r.v[kk] = *reinterpret_cast<const f32x4*>(base + (ok ? (long)(k + dk - cv.pad) * cv.C + ci : 0l));
        }
    } else {
        const float* p = base + (long)kb * sk + row;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) r.v[kk] = *reinterpret_cast<const f32x4*>(p + (long)kk * sk);
    }
}
__device__ __forceinline__ void finish_mc(__bf16* __restrict__ lds, const RawMC& r) {
    const int row = (threadIdx.x & 31) * 4, ks = threadIdx.x >> 5;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
        bf16x8 o;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) o[kk] = (__bf16)(((r.ok >> kk) & 1u) ? r.v[kk][mm] : 0.f);
        *reinterpret_cast<bf16x8*>(lds + swz16(row + mm, ks)) = o;
    }
}
template <bool KC> struct RawOf { using type = RawMC; };
template <> struct RawOf<true> { using type = RawKC; };

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmK g) {
    const GemmDesc& d = g.d;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    __bf16* const lds = reinterpret_cast<__bf16*>(smem16);
    auto As = [&](int b) { return lds + b * (128 * PK16); };
    auto Bs = [&](int b) { return lds + (2 + b) * (128 * PK16); };

    const int z = blockIdx.z;
    const int split = z % d.splitk, bz = z / d.splitk;
    const float* A = d.A + (long)bz * d.bsA;
    const float* B = d.B + (long)bz * d.bsB;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int kbeg = split * g.kchunks * BK, kend = min(d.K, kbeg + g.kchunks * BK);   // kchunks counts 16-wide chunks

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const ConvAddr cva{d.conv_a ? d.conv_T : 0, d.conv_C, d.conv_pad}, cvb{d.conv_b ? d.conv_T : 0, d.conv_C, d.conv_pad};
    // interior workgroup: every tile row/column in range, whole 64-wide k-steps, aligned operands -> pipelined loop
    const bool interior = m0 + 128 <= d.M && n0 + 128 <= d.N && kbeg < kend && (kend - kbeg) % BK16 == 0 && g.avec && g.bvec;
    if (interior) {
        using RA = typename RawOf<A_KC>::type;
        using RB = typename RawOf<B_KC>::type;
        RA a0; RB b0;
        auto issue = [&](int k0, RA& ra_, RB& rb_) {
            if constexpr (A_KC) issue_kc(A, d.sam, m0, k0, cva, ra_); else issue_mc(A, d.sak, m0, k0, cva, ra_);
            if constexpr (B_KC) issue_kc(B, d.sbn, n0, k0, cvb, rb_); else issue_mc(B, d.sbk, n0, k0, cvb, rb_);
        };
        auto finish = [&](int buf, const RA& ra_, const RB& rb_) {
            if constexpr (A_KC) finish_kc(As(buf), ra_); else finish_mc(As(buf), ra_);
            if constexpr (B_KC) finish_kc(Bs(buf), rb_); else finish_mc(Bs(buf), rb_);
        };
        auto mma = [&](int buf) {
            const __bf16* as = As(buf);
            const __bf16* bs = Bs(buf);
#pragma unroll
            for (int ks = 0; ks < BK16 / 16; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(as + swz16(wm * 64 + i * 32 + r, 2 * ks + h));
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bs + swz16(wn * 64 + j * 32 + r, 2 * ks + h));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        };
        const int nks = (kend - kbeg) / BK16;
        issue(kbeg, a0, b0);
        finish(0, a0, b0);
        __syncthreads();
        int cur = 0;
        for (int ks = 0; ks < nks; ++ks) {
            const bool more = ks + 1 < nks;
            if (more) issue(kbeg + (ks + 1) * BK16, a0, b0);          // in flight during the MFMAs below
            __builtin_amdgcn_sched_barrier(0);
            mma(cur);
            __builtin_amdgcn_sched_barrier(0);
            if (more) finish(cur ^ 1, a0, b0);
            __syncthreads();
            cur ^= 1;
        }
    } else {
    bf16x8 ra[4], rb[4];
    auto load = [&](int k0) {
        if (A_KC) load16_kc(A, d.sam, m0, d.M, k0, kend, g.avec, cva, ra); else load16_mc(A, d.sak, m0, d.M, k0, kend, g.avec, cva, ra);
        if (B_KC) load16_kc(B, d.sbn, n0, d.N, k0, kend, g.bvec, cvb, rb); else load16_mc(B, d.sbk, n0, d.N, k0, kend, g.bvec, cvb, rb);
    };
    auto store = [&](int b) {
        if (A_KC) store16_kc(As(b), ra); else store16_mc(As(b), ra);
        if (B_KC) store16_kc(Bs(b), rb); else store16_mc(Bs(b), rb);
    };
    if (kbeg < kend) { load(kbeg); store(0); }
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK16) {
        const bool more = k0 + BK16 < kend;
        if (more) load(k0 + BK16);
        const __bf16* as = As(cur);
        const __bf16* bs = Bs(cur);
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(as + swz16(wm * 64 + i * 32 + r, 2 * ks + h));
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bs + swz16(wn * 64 + j * 32 + r, 2 * ks + h));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    }

    const RngKey key = rng_key(d.seed, d.site);
    float* C = d.C + (long)bz * d.bsC;
    float* ws = d.splitk > 1 ? d.ws + ((long)split * d.batch + bz) * (long)d.M * d.N : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r;
            if (n >= d.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= d.M) continue;
                if (ws) ws[(long)m * d.N + n] = acc[i][j][e];
                else epilogue_store(d, C, m, n, acc[i][j][e], key);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// bf16-SOURCE variant: both operands already bf16 and K-contiguous in memory (A16 [M][lda], B16 [N][ldb]), staged by
// the cast kernels below (or handed over by the caller).  Whole 128x128 tiles and whole 64-wide k-steps only.  Half the
// operand bytes per MFMA of the converting kernel above and no conversion in the loop; two k-steps of 16-byte loads
// (8 VGPRs each per operand) are kept in flight per wave.
// ---------------------------------------------------------------------------------------------
// Block tile (32*TM*WM) x (32*TN*WN): WM x WN waves, each a TM x TN grid of 32x32 MFMA tiles; PF k-steps in flight.
// Used as 128x128 (2x2 waves of 64x64), PF = 2: see the launch site for the other shapes that were measured.
template <int NA, int NB> struct Raw16 { bf16x8 a[NA], b[NB]; unsigned ok; };
// CONV_A: A16 is the bf16 copy of the frames X[M][C] and the operand is its implicit im2col (ConvAddr); C % 64 == 0, so
// one 64-wide k-step lies inside one tap: the tile is the frame tile shifted by (tap - pad) rows, rows that leave
// their utterance read as zero.  tpos[i] = (m0 + row_i) % T.
template <bool CONV_A, int NT, int NA, int NB>
__device__ __forceinline__ void issue16(const __bf16* __restrict__ A, long lda, const __bf16* __restrict__ B, long ldb, int k0,
                                        ConvAddr cv, const int (&tpos)[NA], Raw16<NA, NB>& r) {
    r.ok = ~0u;
    int shift = 0, ka = k0;
    if constexpr (CONV_A) { const int dk = k0 / cv.C; ka = k0 - dk * cv.C; shift = dk - cv.pad; }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = threadIdx.x + i * NT;
        long row = q >> 3;
        if constexpr (CONV_A) {
            const bool ok = (unsigned)(tpos[i] + shift) < (unsigned)cv.T;
            if (ok) row += shift; else r.ok &= ~(1u << i);
        }
        r.a[i] = *reinterpret_cast<const bf16x8*>(A + row * lda + ka + (q & 7) * 8);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int q = threadIdx.x + i * NT;
        r.b[i] = *reinterpret_cast<const bf16x8*>(B + (long)(q >> 3) * ldb + k0 + (q & 7) * 8);
    }
}
template <bool CONV_A, int NT, int NA, int NB>
__device__ __forceinline__ void finish16(__bf16* __restrict__ as, __bf16* __restrict__ bs, const Raw16<NA, NB>& r) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = threadIdx.x + i * NT;
        bf16x8 a = r.a[i];
        if constexpr (CONV_A) {
            if (!((r.ok >> i) & 1u)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (__bf16)0.f;
            }
        }
        *reinterpret_cast<bf16x8*>(as + swz16(q >> 3, q & 7)) = a;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int q = threadIdx.x + i * NT;
        *reinterpret_cast<bf16x8*>(bs + swz16(q >> 3, q & 7)) = r.b[i];
    }
}

template <bool CONV_A, int WM, int WN, int TM, int TN, int PF>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16src_kernel(GemmK g, const __bf16* __restrict__ A16, long lda, const __bf16* __restrict__ B16, long ldb) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN, NA = BM * 8 / NT, NB = BN * 8 / NT;
    const ConvAddr cva{g.d.conv_T, g.d.conv_C, g.d.conv_pad};
    int tpos[NA];
    const GemmDesc& d = g.d;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    __bf16* const lds = reinterpret_cast<__bf16*>(smem16);
    auto As = [&](int b) { return lds + b * (BM * PK16); };
    auto Bs = [&](int b) { return lds + 2 * (BM * PK16) + b * (BN * PK16); };

    const int split = blockIdx.z;
    // Tile of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (linear id % 8) and every XCD has its own
    // 4 MB L2: with the plain (x, y) -> (n, m) map each XCD sees every A row panel and one eighth of the B panels and
    // re-fetches operands 5-6x (profiles/r01_pmc_gemm_traffic.json).  Remap so that an XCD works through a CONTIGUOUS
    // run of tiles, ordered in groups of 8 tile rows (a group's tiles share 8 A panels and walk the B panels once).
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.xcd_swizzle) {
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy;
        const int lin = by * gx + bx;
        const int pid = (lin & 7) * (total >> 3) + (lin >> 3);
        constexpr int GM = 8;
        const int per_group = GM * gx, group = pid / per_group, first_m = group * GM;
        const int gsz = min(gy - first_m, GM);
        by = first_m + (pid % per_group) % gsz;
        bx = (pid % per_group) / gsz;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = split * g.kchunks * BK, kend = min(d.K, kbeg + g.kchunks * BK);
    const __bf16* A = A16 + (long)m0 * lda;
    const __bf16* B = B16 + (long)n0 * ldb;
#pragma unroll
    for (int i = 0; i < NA; ++i) tpos[i] = CONV_A ? (m0 + ((int)(threadIdx.x + i * NT) >> 3)) % cva.T : 0;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mma = [&](int buf) {
        const __bf16* as = As(buf);
        const __bf16* bs = Bs(buf);
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(as + swz16(wm * (32 * TM) + i * 32 + r, 2 * ks + h));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bs + swz16(wn * (32 * TN) + j * 32 + r, 2 * ks + h));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    const int nks = (kend - kbeg) / BK16;
    if constexpr (PF == 1) {                    // one k-step in flight (the 256x256 tile has no registers for two)
        if (nks > 0) {
            Raw16<NA, NB> r0;
            issue16<CONV_A, NT>(A, lda, B, ldb, kbeg, cva, tpos, r0);
            finish16<CONV_A, NT>(As(0), Bs(0), r0);
            __syncthreads();
            for (int ks = 0; ks < nks; ++ks) {
                if (ks + 1 < nks) issue16<CONV_A, NT>(A, lda, B, ldb, kbeg + (ks + 1) * BK16, cva, tpos, r0);
                __builtin_amdgcn_sched_barrier(0);
                mma(ks & 1);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < nks) finish16<CONV_A, NT>(As((ks & 1) ^ 1), Bs((ks & 1) ^ 1), r0);
                __syncthreads();
            }
        }
    } else if (nks > 0) {
        Raw16<NA, NB> r0, r1;
        issue16<CONV_A, NT>(A, lda, B, ldb, kbeg, cva, tpos, r0);
        if (nks > 1) issue16<CONV_A, NT>(A, lda, B, ldb, kbeg + BK16, cva, tpos, r1);
        finish16<CONV_A, NT>(As(0), Bs(0), r0);
        __syncthreads();
        // invariant at the top of step ks: LDS buffer ks&1 holds k-step ks, `nxt` holds k-step ks+1 (in flight),
        // `free` is empty and receives k-step ks+2 before the MFMAs
        auto step = [&](int ks, Raw16<NA, NB>& nxt, Raw16<NA, NB>& free_) {
            if (ks + 2 < nks) issue16<CONV_A, NT>(A, lda, B, ldb, kbeg + (ks + 2) * BK16, cva, tpos, free_);
            __builtin_amdgcn_sched_barrier(0);
            mma(ks & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < nks) finish16<CONV_A, NT>(As((ks & 1) ^ 1), Bs((ks & 1) ^ 1), nxt);
            __syncthreads();
        };
        int ks = 0;
        for (; ks + 1 < nks; ks += 2) { step(ks, r1, r0); step(ks + 1, r0, r1); }
        if (ks < nks) step(ks, r1, r0);
    }

    const RngKey key = rng_key(d.seed, d.site);
    float* C = d.C;
    float* ws = d.splitk > 1 ? d.ws + (long)split * (long)d.M * d.N : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (32 * TN) + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (ws) ws[(long)m * d.N + n] = acc[i][j][e];
                else epilogue_store(d, C, m, n, acc[i][j][e], key);
            }
        }
}

template <bool CONV_A, int WM, int WN, int TM, int TN, int PF>
int launch_bf16src(const GemmK& g, const __bf16* pa, long lda, const __bf16* pb, long ldb, int splitk, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const size_t smem = (size_t)2 * (BM + BN) * PK16 * sizeof(__bf16);
    T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16src_kernel<CONV_A, WM, WN, TM, TN, PF>), smem));   // (per device and kernel)
    GemmK gk = g;
    static const int swz = getenv("T2_GEMM_XCD") ? atoi(getenv("T2_GEMM_XCD")) : 1;
    gk.xcd_swizzle = swz && ((g.d.N / BN) * (g.d.M / BM)) % 8 == 0 && (g.d.M / BM) >= 8;
    hipLaunchKernelGGL((gemm_bf16src_kernel<CONV_A, WM, WN, TM, TN, PF>), dim3(g.d.N / BN, g.d.M / BM, splitk), dim3(64 * WM * WN), smem, s, gk, pa, lda, pb, ldb);
    T2_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 block tile, BK = 64, 8 waves as 2 (M) x 4 (N), each owning 128 x 64 of the tile as 4 x 2 MFMA tiles of
// 32 x 32 (128 accumulator registers).  One workgroup per CU; 128 KB of LDS = 2 K-tiles x 4 half-tiles of 16 KB:
//     A0 / A1: the first / second 64 rows of BOTH wave rows        B0 / B1: the first / second 32 columns of the 4 wave columns
// so a half-tile holds what every wave needs for one operand of one quadrant of its output.  Operands go global -> LDS
// with LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers); an instruction writes 1 KB = 8 rows of 128 B in lane
// order, and the bank swizzle (16-byte chunk ^= (row >> 1) & 7: conflict-free ds_read_b128 of 32 rows x 2 chunks) is put on
// the SOURCE address of each lane.
//
// K loop: 4 phases per K-tile, one output quadrant x K = 64 (8 MFMAs) each:
//     phase   quadrant   fragment reads      LDS-DMA issued (one half-tile = 2 instructions per thread)
//       0     (a0, b0)   A0, B0 of kt        B1 of kt+1
//       1     (a0, b1)   B1 of kt            A1 of kt+1
//       2     (a1, b1)   A1 of kt            A0 of kt+2
//       3     (a1, b0)   -                   B0 of kt+2
//   phase = { reads ; issue ; s_waitcnt vmcnt(8) ; s_barrier ; MFMAs ; s_barrier }: raw barriers and counted waits, so
//   four half-tiles stay in flight across every barrier (a half-tile is read 5-6 phases after it was issued).  A staged
//   buffer is read one phase after the wait + barrier that retires it and restaged at least two phases after its last
//   read.  The two wave rows run one barrier apart (wave row 1 enters through an extra barrier), so on each SIMD one wave
//   reads fragments while the other one issues MFMAs.  The last two K-tiles drain with vmcnt 4 / 2 / 0.
// ---------------------------------------------------------------------------------------------
#ifndef T2_G256_PRIO
#define T2_G256_PRIO 1
#endif
template <bool CONV_A>
__global__ __launch_bounds__(512) void gemm_bf16src256_kernel(GemmK g, const __bf16* __restrict__ A16, long lda, const __bf16* __restrict__ B16, long ldb) {
    const GemmDesc& d = g.d;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.xcd_swizzle) {                                   // as gemm_bf16src_kernel: an XCD walks a contiguous run of tiles
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy;
        const int lin = by * gx + bx;
        const int pid = (lin & 7) * (total >> 3) + (lin >> 3);
        constexpr int GM = 4;
        const int per_group = GM * gx, group = pid / per_group, first_m = group * GM;
        const int gsz = min(gy - first_m, GM);
        by = first_m + (pid % per_group) % gsz;
        bx = (pid % per_group) / gsz;
    }
    const int m0 = by * 256, n0 = bx * 256, split = blockIdx.z;
    const int kbeg = split * g.kchunks * BK, kend = min(d.K, kbeg + g.kchunks * BK);
    const int nkt = (kend - kbeg) / 64;                    // >= 2 (launch site)
    const int nkt2 = (nkt + 1) & ~1;                       // the loop runs whole pairs: an odd count gets one K-tile of zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: LDS-DMA bases live in SGPRs)
    const int wr = wave >> 2, wc = wave & 3, r = lane & 31, hk = lane >> 5;

    // per-thread source offsets (bytes) of the two LDS-DMA instructions of each half-tile, swizzled
    unsigned offA[2][2], offB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = i * 512 + tid, hr = s >> 3, gc = (s & 7) ^ ((hr >> 1) & 7);
            offA[h][i] = (unsigned)((((hr >> 6) * 128 + h * 64 + (hr & 63)) * lda + gc * 8) * 2);
            offB[h][i] = (unsigned)((((hr >> 5) * 64 + h * 32 + (hr & 31)) * ldb + gc * 8) * 2);
        }
    // CONV_A: A16 holds the frames X[M][C] (lda = C) and the operand is their implicit im2col: K-tile k0 lies inside tap
    // dk = k0 / C (C % 64 == 0), so its rows are the frame rows shifted by dk - pad; a row that leaves its utterance gets
    // an offset past the buffer's range, which the LDS-DMA fills with zeros.
    const ConvAddr cva{d.conv_T, d.conv_C, d.conv_pad};
    int tposA[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hr = (i * 512 + tid) >> 3;
            tposA[h][i] = CONV_A ? (m0 + (hr >> 6) * 128 + h * 64 + (hr & 63)) % cva.T : 0;
        }
    int cdk[2], cka[2];                                    // tap and channel offset of the next K-tile of A half 0 / 1
    cdk[0] = cdk[1] = CONV_A ? kbeg / cva.C : 0;
    cka[0] = cka[1] = CONV_A ? kbeg % cva.C : 0;
    auto rsA = CONV_A ? __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(A16), 0, (int)((long)d.M * lda * 2), 0x00020000)
                      : __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(A16 + (long)m0 * lda + kbeg), 0, (int)(256 * lda * 2), 0x00020000);
    auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(B16 + (long)n0 * ldb + kbeg), 0, (int)(256 * ldb * 2), 0x00020000);
    // half-tile ht (0: A0, 1: A1, 2: B0, 3: B1) of K-tile kt -> LDS slot (kt & 1, ht); the calls of one half come in kt order
    auto stage = [&](int ht, int kt) {
        unsigned char* dst = smem16 + (((kt & 1) * 4 + ht) << 14) + wave * 1024;
        const int ko = kt * 128;
        const bool live = kt < nkt;                        // (the padding K-tile of an odd count: offsets past the range read zeros)
        if (ht < 2) {
            if constexpr (CONV_A) {
                const int shift = cdk[ht] - cva.pad;
                const unsigned rel = (unsigned)(((long)(m0 + shift) * lda + cka[ht]) * 2);      // (wraps for m0 + shift < 0: those rows are masked)
                const unsigned o0 = live && (unsigned)(tposA[ht][0] + shift) < (unsigned)cva.T ? offA[ht][0] + rel : 0x80000000u;
                const unsigned o1 = live && (unsigned)(tposA[ht][1] + shift) < (unsigned)cva.T ? offA[ht][1] + rel : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)dst, 16, o0, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dst + 8192), 16, o1, 0, 0, 0);
                cka[ht] += 64;
                if (cka[ht] == cva.C) { cka[ht] = 0; ++cdk[ht]; }
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)dst, 16, live ? offA[ht][0] : 0x80000000u, ko, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dst + 8192), 16, live ? offA[ht][1] : 0x80000000u, ko, 0, 0);
            }
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)dst, 16, live ? offB[ht - 2][0] : 0x80000000u, ko, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(dst + 8192), 16, live ? offB[ht - 2][1] : 0x80000000u, ko, 0, 0);
        }
    };
    // fragment addresses inside a half-tile: row hr, 16-byte chunk c at hr * 128 + ((c ^ ((hr >> 1) & 7)) << 4)
    const int sw = (r >> 1) & 7;
    unsigned cofs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) cofs[ks] = (unsigned)(((2 * ks + hk) ^ sw) << 4);
    const unsigned rowA = (unsigned)((wr * 64 + r) * 128), rowB = (unsigned)((wc * 32 + r) * 128);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 Af[2][4], Bf[2][4];

    auto read_a = [&](int buf, int h) {
        const unsigned char* base = smem16 + ((buf * 4 + h) << 14) + rowA;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Af[i][ks] = *reinterpret_cast<const bf16x8*>(base + i * 4096 + cofs[ks]);
    };
    auto read_b = [&](int buf, int h) {
        const unsigned char* base = smem16 + ((buf * 4 + 2 + h) << 14) + rowB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) Bf[h][ks] = *reinterpret_cast<const bf16x8*>(base + cofs[ks]);
    };
    auto mma = [&](int a, int b) {
#if T2_G256_PRIO
        __builtin_amdgcn_s_setprio(1);
#else
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[a * 2 + i][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[i][ks], Bf[b][ks], acc[a * 2 + i][b], 0, 0, 0);
#if T2_G256_PRIO
        __builtin_amdgcn_s_setprio(0);
#else
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
#define T2_G256_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    // one K-tile (buffer BUF): TAIL 0 = steady state, 1 = K-tile nkt-2, 2 = K-tile nkt-1
    auto ktile = [&](auto bufc, auto tailc, int kt) {
        constexpr int BUF = decltype(bufc)::value, TAIL = decltype(tailc)::value;
        // phase 0
        read_b(BUF, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(BUF, 0);
        if (TAIL < 2) { stage(3, kt + 1); T2_G256_WAIT(8); } else T2_G256_WAIT(2);
        __builtin_amdgcn_s_barrier();
        mma(0, 0);
        __builtin_amdgcn_s_barrier();
        // phase 1
        read_b(BUF, 1);
        if (TAIL < 2) { stage(1, kt + 1); T2_G256_WAIT(8); } else T2_G256_WAIT(0);
        __builtin_amdgcn_s_barrier();
        mma(0, 1);
        __builtin_amdgcn_s_barrier();
        // phase 2
        read_a(BUF, 1);
        if (TAIL == 0) { stage(0, kt + 2); T2_G256_WAIT(8); }
        __builtin_amdgcn_s_barrier();
        mma(1, 1);
        __builtin_amdgcn_s_barrier();
        // phase 3
        if (TAIL == 0) { stage(2, kt + 2); T2_G256_WAIT(8); } else if (TAIL == 1) T2_G256_WAIT(4);
        __builtin_amdgcn_s_barrier();
        mma(1, 0);
        __builtin_amdgcn_s_barrier();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    // prologue: K-tile 0 and A0, B0 of K-tile 1
    stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0); stage(0, 1); stage(2, 1);
    T2_G256_WAIT(8);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    int kt = 0;
    for (; kt + 2 < nkt2; kt += 2) { ktile(I0{}, I0{}, kt); ktile(I1{}, I0{}, kt + 1); }
    ktile(I0{}, I1{}, kt);
    ktile(I1{}, I2{}, kt + 1);
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef T2_G256_WAIT

    // Epilogue through LDS (free now): each wave lays 64 rows x 64 columns of its sub-tile out row-major in its own
    // 16 KB and walks it with 16-byte reads, so a row leaves as 256 contiguous bytes and the per-row work (row map,
    // dropout index base) is done once per 4 elements.
    const RngKey key = rng_key(d.seed, d.site);
    float* ws = d.splitk > 1 ? d.ws + (long)split * (long)d.M * d.N : nullptr;
    float* stg = reinterpret_cast<float*>(smem16) + wave * 4096;
    const bool vec_ok = ws || ((reinterpret_cast<uintptr_t>(d.C) & 15) == 0 && d.ldc % 4 == 0);
    const float inv_keep = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    const int c4 = (lane & 15) * 4, n = n0 + wc * 64 + c4;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (!ws) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bias[c] = (d.bias1 ? d.bias1[n + c] : 0.f) + (d.bias2 ? d.bias2[n + c] : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) stg[(ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * hk) * 64 + j * 32 + r] = acc[2 * p + ii][j][e];
        __syncthreads();
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + (lane >> 4);
            const int m = m0 + wr * 128 + p * 64 + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 64 + c4);
            if (ws) { *reinterpret_cast<f32x4*>(ws + (long)m * d.N + n) = v; continue; }
            const uint32_t ibase = d.drop_base + (uint32_t)m * d.drop_mstride + (uint32_t)n;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = apply_act(d.alpha * v[c] + bias[c], d.act);
                if (d.drop_p > 0.f) x = rng_keep(key, ibase + c, d.drop_p) ? x * inv_keep : 0.f;
                v[c] = x;
            }
            const long mo = d.crow_mod ? (long)(m % d.crow_mod) * d.crow_mul + m / d.crow_mod : (long)m;
            float* pc = d.C + mo * d.ldc + n;
            if (vec_ok) {
                if (d.beta != 0.f) v += d.beta * *reinterpret_cast<const f32x4*>(pc);
                *reinterpret_cast<f32x4*>(pc) = v;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) pc[c] = d.beta != 0.f ? v[c] + d.beta * pc[c] : v[c];
            }
        }
        if (p == 0) __syncthreads();
    }
}

template <bool CONV_A>
int launch_bf16src256(const GemmK& g, const __bf16* pa, long lda, const __bf16* pb, long ldb, int splitk, hipStream_t s) {
    constexpr size_t smem = 128 * 1024;
    T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16src256_kernel<CONV_A>), smem));
    GemmK gk = g;
    const int gx = g.d.N / 256, gy = g.d.M / 256;
    gk.xcd_swizzle = (gx * gy) % 8 == 0 && gy >= 4;
    hipLaunchKernelGGL((gemm_bf16src256_kernel<CONV_A>), dim3(gx, gy, splitk), dim3(512), smem, s, gk, pa, lda, pb, ldb);
    T2_LAUNCH_CHECK();
    return 0;
}

// staging casts: fp32 operand -> bf16 [rows][K] (K contiguous, leading dimension K)
// k contiguous in the source: 8 elements per task
__global__ void stage_kc_kernel(const float* __restrict__ src, long ld, __bf16* __restrict__ dst, int rows, int K) {
    const int k8 = K >> 3;
    const long total = (long)rows * k8;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long row = t / k8; const int s = (int)(t - row * k8);
        const float* p = src + row * ld + s * 8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
        *reinterpret_cast<bf16x8*>(dst + row * (long)K + s * 8) = pack8(lo, hi);
    }
}
// rows contiguous in the source (src[k*ld + row]): 64 k x 64 rows per workgroup through LDS
__global__ __launch_bounds__(256) void stage_mc_kernel(const float* __restrict__ src, long ld, __bf16* __restrict__ dst, int rows, int K) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (threadIdx.x >> 4) + 16 * i, r4 = (threadIdx.x & 15) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)(k0 + k) * ld + r0 + r4);
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[k][r4 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kg = threadIdx.x & 7, row = (threadIdx.x >> 3) + 32 * i;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)tile[kg * 8 + j][row];
        *reinterpret_cast<bf16x8*>(dst + (long)(r0 + row) * K + k0 + kg * 8) = o;
    }
}

// implicit-im2col B operand (k = frame m, row n = tap*C + ci): dst[n][m] = X[m + tap - pad][ci] inside the utterance, else 0
__global__ __launch_bounds__(256) void stage_conv_mc_kernel(const float* __restrict__ X, ConvAddr cv, __bf16* __restrict__ dst, int K) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int dk = r0 / cv.C, ci0 = r0 - dk * cv.C, shift = dk - cv.pad;          // C % 64 == 0: one tap per row tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (threadIdx.x >> 4) + 16 * i, r4 = (threadIdx.x & 15) * 4;
        const int m = k0 + k;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)(m % cv.T + shift) < (unsigned)cv.T) v = *reinterpret_cast<const f32x4*>(X + (long)(m + shift) * cv.C + ci0 + r4);
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[k][r4 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kg = threadIdx.x & 7, row = (threadIdx.x >> 3) + 32 * i;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)tile[kg * 8 + j][row];
        *reinterpret_cast<bf16x8*>(dst + (long)(r0 + row) * K + k0 + kg * 8) = o;
    }
}

int stage_operand(const float* src, bool kc, long ld, __bf16* dst, int rows, int K, hipStream_t s) {
    if (kc) {
        long blocks = ((long)rows * (K >> 3) + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(stage_kc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, ld, dst, rows, K);
    } else {
        hipLaunchKernelGGL(stage_mc_kernel, dim3(rows / 64, K / 64), dim3(256), 0, s, src, ld, dst, rows, K);
    }
    T2_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN>
void launch_cfg(const GemmK& g, bool akc, bool bkc, dim3 grid, hipStream_t s) {
    if (akc && bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true>), grid, dim3(256), 0, s, g);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false>), grid, dim3(256), 0, s, g);
    else if (!akc && bkc) hipLaunchKernelGGL((gemm_kernel<BM, BN, false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false>), grid, dim3(256), 0, s, g);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int stage_bf16(const float* src, bool kc, long ld, __bf16* dst, int rows, int K, hipStream_t s) {
    T2_REQUIRE(rows > 0 && K > 0 && rows % 64 == 0 && K % 64 == 0 && ld % 4 == 0 && aligned16(src) && aligned16(dst),
               "stage_bf16: rows=%d K=%d ld=%ld must be whole 64-tiles of 16-byte aligned rows", rows, K, ld);
    return stage_operand(src, kc, ld, dst, rows, K, s);
}

int gemm(const GemmDesc& din, hipStream_t s) {
    GemmK g{};
    g.d = din;
    GemmDesc& d = g.d;
    T2_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0, "gemm: bad shape M=%d N=%d K=%d batch=%d", d.M, d.N, d.K, d.batch);
    if (d.conv_a || d.conv_b) {
        T2_REQUIRE(d.conv_T > 0 && d.conv_C > 0 && d.conv_C % 4 == 0 && !(d.conv_a && d.conv_b), "gemm: bad implicit-conv operand (T=%d C=%d)", d.conv_T, d.conv_C);
        T2_REQUIRE(d.batch == 1, "gemm: implicit-conv operands need batch == 1");
        if (d.conv_a) { d.sam = 0; d.sak = 1; T2_REQUIRE(d.K % d.conv_C == 0 && aligned16(d.A), "gemm: conv A operand: K=%d C=%d", d.K, d.conv_C); }
        if (d.conv_b) { d.sbk = 0; d.sbn = 1; T2_REQUIRE(d.N % d.conv_C == 0 && aligned16(d.B), "gemm: conv B operand: N=%d C=%d", d.N, d.conv_C); }
    }
    T2_REQUIRE(d.sam == 1 || d.sak == 1, "gemm: A needs one unit stride (sam=%ld sak=%ld)", d.sam, d.sak);
    T2_REQUIRE(d.sbn == 1 || d.sbk == 1, "gemm: B needs one unit stride (sbn=%ld sbk=%ld)", d.sbn, d.sbk);
    T2_REQUIRE(d.drop_p == 0.f || (d.batch == 1 && (long)d.M * d.N < (1l << 32)), "gemm: dropout epilogue needs batch==1 and M*N < 2^32");
    const bool akc = d.sak == 1, bkc = d.sbk == 1;
    if (d.drop_mstride == 0) d.drop_mstride = (uint32_t)d.N;
    g.avec = aligned16(d.A) && (d.bsA % 4 == 0) && ((akc ? d.sam : d.sak) % 4 == 0);
    g.bvec = aligned16(d.B) && (d.bsB % 4 == 0) && ((bkc ? d.sbn : d.sbk) % 4 == 0);

    // 128x128 tiles whenever both extents allow it (4 MFMAs per 4 LDS fragment reads); a grid that
    // would not fill the chip is completed by split-K when scratch is available, else by 64x64 tiles.
    // bf16-operand mode: large GEMMs only (both extents >= 64), conv operands need C % 8 == 0
    const bool use_bf16 = g_precision == 1 && !d.fp32_only && d.M >= 64 && d.N >= 64 && d.K >= 64 &&
                          (!(d.conv_a || d.conv_b) || d.conv_C % 8 == 0);
    // bf16-source path: both operands staged as bf16 [rows][K] in front of the split-K scratch (or handed over by the
    // caller); pays when each staged element is reused by many tiles, i.e. when both extents are large
    bool staged = false;
    __bf16* a16 = nullptr; __bf16* b16 = nullptr;
    const bool conv_any = d.conv_a || d.conv_b;
    if (use_bf16 && g_stage && (!conv_any || (d.conv_C % 64 == 0 && !d.A16 && !d.B16)) && d.batch == 1 && d.M % 128 == 0 && d.N % 128 == 0 && d.K % 64 == 0) {
        // implicit-conv A: only the frames [M][C] are staged (the kernel shifts rows per tap); implicit-conv B: the whole
        // im2col transpose [N][K] is written out (N = taps*C rows of K frames)
        const size_t a_elems = d.conv_a ? (size_t)d.M * d.conv_C : (size_t)d.M * d.K;
        const size_t need_a = d.A16 ? 0 : ((a_elems * sizeof(__bf16) + 255) & ~(size_t)255);
        const size_t need_b = d.B16 ? 0 : (((size_t)d.N * d.K * sizeof(__bf16) + 255) & ~(size_t)255);
        const bool big = (d.A16 || d.N >= g_stage_min) && (d.B16 || d.M >= g_stage_min);
        const bool ok_src = (d.A16 || g.avec) && (d.B16 || g.bvec) &&
                            (!d.A16 || (aligned16(d.A16) && d.lda16 % 8 == 0)) && (!d.B16 || (aligned16(d.B16) && d.ldb16 % 8 == 0));
        if (big && ok_src && (need_a + need_b == 0 || (d.ws && aligned16(d.ws) && d.ws_bytes >= need_a + need_b))) {
            staged = true;
            unsigned char* base = reinterpret_cast<unsigned char*>(d.ws);
            if (need_a) a16 = reinterpret_cast<__bf16*>(base);
            if (need_b) b16 = reinterpret_cast<__bf16*>(base + need_a);
            if (need_a + need_b) { d.ws = reinterpret_cast<float*>(base + need_a + need_b); d.ws_bytes -= need_a + need_b; }
        }
    }
    // 256 x 256 tiles (gemm_bf16src256_kernel) when the staged operands are whole 256-tiles and K splits into an even
    // number of 64-wide K-tiles; its split-K factor fills whole rounds of one workgroup per CU
    static const int g_t256 = env_int("T2_GEMM_256", 1);
    const bool use256 = staged && g_t256 && d.M % 256 == 0 && d.N % 256 == 0 && d.K >= 128 && (!d.conv_a || (long)d.M * d.conv_C * 2 < (1l << 31));
    int split256 = 1;
    if (use256) {
        const long tiles = (long)(d.M / 256) * (d.N / 256);
        const int nkt = d.K / 64;
        if (d.splitk > 0) split256 = std::min(d.splitk, std::max(1, nkt / 2));
        else if (d.ws && d.beta == 0.f && tiles < 512) {
            double best = 0.0;
            for (int sp = 1; sp <= 16; ++sp) {
                if (sp > 1 && nkt / sp < 16) break;
                const long wgs = tiles * sp, rounds = (wgs + 255) / 256;
                const double eff = (double)wgs / (256.0 * rounds);
                if (eff > best * 1.05) { best = eff; split256 = sp; }
            }
        }
        if (split256 > 1 && !(d.ws && d.beta == 0.f)) split256 = 1;
        if (split256 > 1) {
            const size_t per = (size_t)d.M * d.N * sizeof(float);
            if ((size_t)split256 * per > d.ws_bytes) split256 = (int)(d.ws_bytes / per);
            if (split256 < 1) split256 = 1;
        }
    }
    const int kch = (d.K + BK - 1) / BK;
    const bool can_split = d.ws && d.beta == 0.f && kch >= 64;
    const long tiles128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch;
    const bool small = !use_bf16 && ((d.M <= 64 || d.N <= 64) || (tiles128 < 256 && !can_split));
    const int BMN = small ? 64 : 128;
    const int tm = (d.M + BMN - 1) / BMN, tn = (d.N + BMN - 1) / BMN;
    int splitk = 1;
    if (d.ws && d.beta == 0.f) {
        splitk = d.splitk;
        if (splitk <= 0) {
            const long tiles = (long)tm * tn * d.batch;
            splitk = 1;
            // fewer than two tile-waves over the 256 CUs and a long K: split so that every CU holds several workgroups
            // (a lone 128x128 workgroup per CU cannot cover its own operand latency)
            if (tiles < 512 && kch >= 64) {
                splitk = (int)((1024 + tiles - 1) / tiles);
                if (splitk > kch / 16) splitk = kch / 16;
            }
        }
        const size_t per = (size_t)d.batch * d.M * d.N * sizeof(float);
        if ((size_t)splitk * per > d.ws_bytes) splitk = (int)(d.ws_bytes / per);
        if (splitk < 1) splitk = 1;
    }
    if (use256) splitk = split256;
    g.kchunks = (kch + splitk - 1) / splitk;
    if (use_bf16) g.kchunks = (g.kchunks + 3) & ~3;   // whole 64-wide chunks per split
    splitk = (kch + g.kchunks - 1) / g.kchunks;     // drop empty splits
    if (use256 && splitk > 1 && kch - (splitk - 1) * g.kchunks < 8) {   // the 256-tile kernel needs two K-tiles in every split
        --splitk;
        g.kchunks = ((kch + splitk - 1) / splitk + 3) & ~3;
        splitk = (kch + g.kchunks - 1) / g.kchunks;
    }
    d.splitk = splitk;
    static const int g_log = env_int("T2_GEMM_LOG", 0);      // dev: one line per product (shape, kernel, split, which operands get staged)
    if (g_log)
        fprintf(stderr, "t2gemm M=%d N=%d K=%d batch=%d %s%s kernel=%s splitk=%d stageA=%d stageB=%d convA=%d convB=%d beta=%g\n", d.M, d.N, d.K, d.batch,
                akc ? "A[m][k]" : "A[k][m]", bkc ? " B[n][k]" : " B[k][n]", staged ? (use256 ? "src256" : "src128") : use_bf16 ? "bf16conv" : small ? "f32_64" : "f32_128",
                splitk, a16 != nullptr, b16 != nullptr, d.conv_a, d.conv_b, (double)d.beta);
    T2_REQUIRE((long)d.batch * splitk <= 65535, "gemm: batch*splitk too large (%d*%d)", d.batch, splitk);
    dim3 grid(tn, tm, d.batch * splitk);
    if (staged) {
        if (a16) {
            if (d.conv_a) T2_TRY_RC(stage_operand(d.A, true, d.conv_C, a16, d.M, d.conv_C, s));
            else T2_TRY_RC(stage_operand(d.A, akc, akc ? d.sam : d.sak, a16, d.M, d.K, s));
        }
        if (b16) {
            if (d.conv_b) {
                hipLaunchKernelGGL(stage_conv_mc_kernel, dim3(d.N / 64, d.K / 64), dim3(256), 0, s, d.B, ConvAddr{d.conv_T, d.conv_C, d.conv_pad}, b16, d.K);
                T2_LAUNCH_CHECK();
            } else T2_TRY_RC(stage_operand(d.B, bkc, bkc ? d.sbn : d.sbk, b16, d.N, d.K, s));
        }
        const __bf16* pa = d.A16 ? d.A16 : a16; const long lda = d.A16 ? d.lda16 : (d.conv_a ? d.conv_C : d.K);
        const __bf16* pb = d.B16 ? d.B16 : b16; const long ldb = d.B16 ? d.ldb16 : d.K;
        // 128x128 block tile, two k-steps in flight.  Measured alternatives (same template, other parameters): 256x128 with
        // 8 waves 5-12 % slower, 256x256 with 128x64 wave tiles and one k-step in flight 4.6x slower (one workgroup per CU:
        // nothing overlaps its barriers) — the kernel is bound by latency hiding, not by operand bytes per CU
        if (use256 && d.conv_a) T2_TRY_RC((launch_bf16src256<true>(g, pa, lda, pb, ldb, splitk, s)));
        else if (use256) T2_TRY_RC((launch_bf16src256<false>(g, pa, lda, pb, ldb, splitk, s)));
        else if (d.conv_a) T2_TRY_RC((launch_bf16src<true, 2, 2, 2, 2, 2>(g, pa, lda, pb, ldb, splitk, s)));
        else T2_TRY_RC((launch_bf16src<false, 2, 2, 2, 2, 2>(g, pa, lda, pb, ldb, splitk, s)));
    } else if (use_bf16) {
        const size_t smem = (size_t)4 * 128 * PK16 * sizeof(__bf16);
        T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16_kernel<true, true>), smem));
        T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16_kernel<true, false>), smem));
        T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16_kernel<false, true>), smem));
        T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16_kernel<false, false>), smem));
        if (akc && bkc) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, dim3(256), smem, s, g);
        else if (akc && !bkc) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, dim3(256), smem, s, g);
        else if (!akc && bkc) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, dim3(256), smem, s, g);
        else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, dim3(256), smem, s, g);
    } else if (small) launch_cfg<64, 64>(g, akc, bkc, grid, s);
    else launch_cfg<128, 128>(g, akc, bkc, grid, s);
    T2_LAUNCH_CHECK();
    if (splitk > 1) {
        const long total = (long)d.batch * d.M * d.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, d);
        T2_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace t2
