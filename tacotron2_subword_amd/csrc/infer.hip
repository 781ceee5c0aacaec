// Autoregressive decode: the tail of one decoder step (Decoder.decode model.py:382-388 + the stop test
// :461,480 + Prenet.forward :13-24 of the next step) in TWO launches that each fill the chip:
//
//   proj_stop_kernel     grid (B, G)      mel_t = W_proj . [dec_h | ctx | ctx_sub] + b ; gate_t likewise ;
//                                         stop_index[b] = t the first time sigmoid(gate_t) > threshold
//   prenet_step_kernel   grid (B, NS, 4)  p2(t+1) = drop(relu(W2 . drop(relu(W1 . mel_t)))) per stream
//
// They replace six skinny GEMM launches (M = batch rows only) + the stop kernel per decoded frame.
// One workgroup per batch item is not enough: a CU pulls ~64 B/clk from L2, so the 1.3 MB of weights a
// whole item needs would take ~10 us through one CU; spread over G (resp. 4) workgroups per item the rows
// stream from L2 in parallel (weights are L2/MALL-resident: 0.66 MB + 2 x 0.34 MB).  Rows are read one
// wave per output, lanes along K, 16 B per lane, several rows in flight.
#include "kernels.h"

namespace t2 {

namespace {

constexpr int NTP = 1024;    // 16 waves: these kernels are chains of dependent L2/MALL round trips (~1 us each), so
                             // every phase requests its first batch of weight rows BEFORE the barrier that
                             // publishes its input, and keeps a whole row (or 4) in flight per wave

__global__ __launch_bounds__(NTP) void proj_stop_kernel(StepTailDesc d, int G) {
    const int b = blockIdx.x, gi = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = NTP / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                       // [WO] decoder output row of this item
    const int M = d.M, WO = d.WO;
    const float* xr = d.dout + (long)b * d.lddout;
    for (int i = tid * 4; i < WO; i += NTP * 4) *reinterpret_cast<f32x4*>(xin + i) = *reinterpret_cast<const f32x4*>(xr + i);
    // outputs o = gi + G*j (o == M is the gate), one wave per output; 2048 columns of a row in flight per pass
    auto row = [&](int o) { return o < M ? d.proj_w + (long)o * WO : d.gate_w; };
    auto load8 = [&](const float* w, int k0, f32x4 (&a)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f32x4*>(w + min(k0 + u * 256 + lane * 4, WO - 4));   // clamped, masked in fma8
    };
    auto fma8 = [&](int k0, const f32x4 (&a)[8], float& acc) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * 256 + lane * 4;
            if (k < WO) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(xin + k);
                acc += a[u][0] * x[0] + a[u][1] * x[1] + a[u][2] * x[2] + a[u][3] * x[3];
            }
        }
    };
    f32x4 a[8];
    load8(row(min(gi + G * wave, M)), 0, a);         // first row of this wave requested before xin is published
    __syncthreads();
    for (int j = wave; gi + G * j <= M; j += NW) {
        const int o = gi + G * j;
        const float* w = row(o);
        float s0 = 0.f;
        for (int k0 = 0; k0 < WO; k0 += 2048) {
            if (j != wave || k0 != 0) load8(w, k0, a);
            fma8(k0, a, s0);
        }
        s0 = wave_sum(s0);
        if (lane == 0) {
            if (o < M) d.mel_out[(long)b * d.ldmel + o] = s0 + d.proj_b[o];
            else {
                const float g = s0 + d.gate_b[0];
                d.gate_out[(long)b * d.ldgate] = g;
                if (d.stop_index && d.stop_index[b] < 0 && 1.0f / (1.0f + expf(-g)) > d.thr) { d.stop_index[b] = d.t; atomicAdd(d.done, 1); }
            }
        }
    }
}

constexpr int NSLICE = 4;    // workgroups per (item, stream): each recomputes layer 1 and owns P/4 rows of layer 2
constexpr int KQ = 4;        // layer-1 K split across thread groups
constexpr int MAXK = 32;     // k per thread held in registers: n_mel <= KQ*MAXK

__global__ __launch_bounds__(NTP) void prenet_step_kernel(StepTailDesc d) {
    const int b = blockIdx.x, s = blockIdx.y, sl = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = NTP / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = d.M, P = d.P;
    float* x = smem;                         // [M]
    float* h1 = smem + ((M + 3) & ~3);       // [P]
    float* part = h1 + P;                    // [KQ][P]
    for (int m = tid; m < M; m += NTP) x[m] = d.x_in ? d.x_in[(long)b * d.ldx_in + m] : 0.f;
    const float scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    const uint32_t ibase = d.drop_base + (uint32_t)b * d.drop_mstride;

    // layer 1 (K = M) over the transposed weight [M][P]: work item (n, kq) takes k = kq, kq+KQ, ...; coalesced across n
    auto load1 = [&](int idx, float (&wv)[MAXK]) {
        const int n = idx % P, kq = idx / P;
        const float* w = d.w1t[s] + n;
#pragma unroll
        for (int u = 0; u < MAXK; ++u) wv[u] = w[(long)min(kq + u * KQ, M - 1) * P];
    };
    // layer 2: rows [n0, n1) of this slice, one wave per row, 4 rows of a wave in flight (256 columns per pass)
    const int n0 = sl * (P / NSLICE), n1 = n0 + P / NSLICE;
    auto load2 = [&](int r0, int k, f32x4 (&a)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(d.w2[s] + (long)min(r0 + u, n1 - 1) * P + min(k, P - 4));
    };
    float wv[MAXK];
    f32x4 a2[4];
    load1(min(tid, P * KQ - 1), wv);                 // both layers' first weight batches are in flight before x is published
    load2(n0 + wave * 4, lane * 4, a2);
    __syncthreads();
    for (int idx = tid; idx < P * KQ; idx += NTP) {
        if (idx != tid) load1(idx, wv);
        const int n = idx % P, kq = idx / P;
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < MAXK; ++u) {
            const int k = kq + u * KQ;
            if (k < M) sum += wv[u] * x[k];
        }
        part[kq * P + n] = sum;
    }
    __syncthreads();
    for (int n = tid; n < P; n += NTP) {
        float sum = 0.f;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) sum += part[kq * P + n];
        float v = fmaxf(sum, 0.f);
        if (d.drop_p > 0.f) v = rng_keep(rng_key(d.seed, d.site1[s]), ibase + (uint32_t)n, d.drop_p) ? v * scale : 0.f;
        h1[n] = v;
        if (sl == 0 && d.p1[s]) d.p1[s][(long)b * d.ldp + n] = v;
    }
    __syncthreads();
    for (int r0 = n0 + wave * 4; r0 < n1; r0 += NW * 4) {
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = lane * 4; k < P; k += 256) {
            if (r0 != n0 + wave * 4 || k != lane * 4) load2(r0, k, a2);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(h1 + k);
#pragma unroll
            for (int u = 0; u < 4; ++u) sum[u] += a2[u][0] * xv[0] + a2[u][1] * xv[1] + a2[u][2] * xv[2] + a2[u][3] * xv[3];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float tot = wave_sum(sum[u]);
            const int n = r0 + u;
            if (lane == 0 && n < n1) {
                float v = fmaxf(tot, 0.f);
                if (d.drop_p > 0.f) v = rng_keep(rng_key(d.seed, d.site2[s]), ibase + (uint32_t)n, d.drop_p) ? v * scale : 0.f;
                d.p2[s][(long)b * d.ldp + n] = v;
                if (d.p2_16[s]) d.p2_16[s][(long)b * d.ldp16 + n] = (__bf16)v;
            }
        }
    }
}

}  // namespace

int step_tail(const StepTailDesc& d, hipStream_t s) {
    T2_REQUIRE(d.B >= 1 && d.NS >= 1 && d.NS <= 2, "step_tail: bad B=%d NS=%d", d.B, d.NS);
    T2_REQUIRE(d.M % 4 == 0 && d.M <= KQ * MAXK && d.P % 16 == 0 && d.WO % 4 == 0, "step_tail: M=%d (x4, <= %d) P=%d (x16) WO=%d (x4) unsupported", d.M, KQ * MAXK, d.P, d.WO);
    StepTailDesc e = d;
    if (d.do_proj) {
        T2_REQUIRE((size_t)d.WO * sizeof(float) <= 64 * 1024, "step_tail: decoder row of %d floats too wide for LDS", d.WO);
        int G = 1;
        while (G < 16 && d.B * G * 2 <= 512) G *= 2;                 // ~256-512 workgroups
        hipLaunchKernelGGL(proj_stop_kernel, dim3(d.B, G), dim3(NTP), (size_t)d.WO * sizeof(float), s, d, G);
        T2_LAUNCH_CHECK();
        e.x_in = d.mel_out; e.ldx_in = d.ldmel;                      // the prenets of step t+1 read the frame just written
    }
    if (d.do_prenet) {
        const size_t smem = ((size_t)((d.M + 3) & ~3) + (size_t)(1 + KQ) * d.P) * sizeof(float);
        hipLaunchKernelGGL(prenet_step_kernel, dim3(d.B, d.NS, NSLICE), dim3(NTP), smem, s, e);
        T2_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace t2
