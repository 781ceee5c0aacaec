// Autoregressive decode: the tail of one decoder step in ONE launch, one workgroup per batch item
// (Decoder.decode model.py:382-388 + the stop test :461,480 + Prenet.forward :13-24 of the next step):
//
//   mel_t = W_proj . [dec_h | ctx | ctx_sub] + b ;  gate_t = w_gate . [...] + b_gate
//   stop_index[b] = t  the first time sigmoid(gate_t) > threshold
//   p2(t+1) = drop(relu(W2 . drop(relu(W1 . mel_t))))          for each stream's prenet
//
// Replaces six skinny GEMM launches (M = batch rows only) + the stop kernel per decoded frame.  The
// projection rows are read one wave per output (lanes along K, 16 B per lane, several rows in flight);
// the weights (0.66 MB + 2 x 0.34 MB) stay L2-resident across the B workgroups.
#include "kernels.h"

namespace t2 {

namespace {

constexpr int NTT = 1024;

__global__ __launch_bounds__(NTT) void step_tail_kernel(StepTailDesc d) {
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = NTT / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                       // [WO]  decoder output row
    float* mel = xin + d.WO;                 // [M+4] frame (prenet input); mel[M] = gate
    float* h1 = mel + ((d.M + 1 + 3) & ~3);  // [NS][P]
    const int M = d.M, P = d.P, WO = d.WO;

    if (d.do_proj) {
        const float* xr = d.dout + (long)b * d.lddout;
        for (int i = tid * 4; i < WO; i += NTT * 4) *reinterpret_cast<f32x4*>(xin + i) = *reinterpret_cast<const f32x4*>(xr + i);
        __syncthreads();
        // M + 1 outputs, one wave each, 2 rows in flight per wave
        for (int o0 = wave; o0 <= M; o0 += 2 * NW) {
            const int o1 = o0 + NW;
            const float* w0 = o0 < M ? d.proj_w + (long)o0 * WO : d.gate_w;
            const float* w1 = o1 < M ? d.proj_w + (long)o1 * WO : d.gate_w;      // o1 > M: computed, discarded
            float s0 = 0.f, s1 = 0.f;
            for (int k = lane * 4; k < WO; k += 256) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(w0 + k);
                const f32x4 c = *reinterpret_cast<const f32x4*>(w1 + k);
                const f32x4 x = *reinterpret_cast<const f32x4*>(xin + k);
                s0 += a[0] * x[0] + a[1] * x[1] + a[2] * x[2] + a[3] * x[3];
                s1 += c[0] * x[0] + c[1] * x[1] + c[2] * x[2] + c[3] * x[3];
            }
            s0 = wave_sum(s0); s1 = wave_sum(s1);
            if (lane == 0) {
                mel[o0] = s0 + (o0 < M ? d.proj_b[o0] : d.gate_b[0]);
                if (o1 <= M) mel[o1] = s1 + (o1 < M ? d.proj_b[o1] : d.gate_b[0]);
            }
        }
        __syncthreads();
        for (int m = tid; m < M; m += NTT) d.mel_out[(long)b * d.ldmel + m] = mel[m];
        if (tid == 0) {
            const float g = mel[M];
            d.gate_out[(long)b * d.ldgate] = g;
            if (d.stop_index && d.stop_index[b] < 0 && 1.0f / (1.0f + expf(-g)) > d.thr) { d.stop_index[b] = d.t; atomicAdd(d.done, 1); }
        }
    } else {
        for (int m = tid; m < M; m += NTT) mel[m] = d.x_in ? d.x_in[(long)b * d.ldx_in + m] : 0.f;
        __syncthreads();
    }
    if (!d.do_prenet) return;

    const float scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    // layer 1: NS*P outputs, K = M: one thread per output (rows are short)
    for (int o = tid; o < d.NS * P; o += NTT) {
        const int s = o / P, n = o % P;
        const float* w = d.w1[s] + (long)n * M;
        float sum = 0.f;
        for (int k = 0; k < M; k += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(w + k);
            sum += a[0] * mel[k] + a[1] * mel[k + 1] + a[2] * mel[k + 2] + a[3] * mel[k + 3];
        }
        float v = fmaxf(sum, 0.f);
        if (d.drop_p > 0.f) v = rng_keep(rng_key(d.seed, d.site1[s]), d.drop_base + (uint32_t)b * d.drop_mstride + (uint32_t)n, d.drop_p) ? v * scale : 0.f;
        h1[o] = v;
        if (d.p1[s]) d.p1[s][(long)b * d.ldp + n] = v;
    }
    __syncthreads();
    // layer 2: NS*P outputs, K = P: one wave per output, 4 rows in flight
    for (int o0 = wave * 4; o0 < d.NS * P; o0 += NW * 4) {
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        const int s = o0 / P;                                   // P % 4 == 0: the 4 rows share a stream
        const float* hs = h1 + s * P;
        for (int k = lane * 4; k < P; k += 256) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(hs + k);
            f32x4 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(d.w2[s] + (long)(o0 % P + u) * P + k);
#pragma unroll
            for (int u = 0; u < 4; ++u) sum[u] += a[u][0] * x[0] + a[u][1] * x[1] + a[u][2] * x[2] + a[u][3] * x[3];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float tot = wave_sum(sum[u]);
            if (lane == 0) {
                const int n = o0 % P + u;
                float v = fmaxf(tot, 0.f);
                if (d.drop_p > 0.f) v = rng_keep(rng_key(d.seed, d.site2[s]), d.drop_base + (uint32_t)b * d.drop_mstride + (uint32_t)n, d.drop_p) ? v * scale : 0.f;
                d.p2[s][(long)b * d.ldp + n] = v;
                if (d.p2_16[s]) d.p2_16[s][(long)b * d.ldp16 + n] = (__bf16)v;
            }
        }
    }
}

}  // namespace

int step_tail(const StepTailDesc& d, hipStream_t s) {
    T2_REQUIRE(d.B >= 1 && d.NS >= 1 && d.NS <= 2, "step_tail: bad B=%d NS=%d", d.B, d.NS);
    T2_REQUIRE(d.M % 4 == 0 && d.P % 4 == 0 && d.WO % 4 == 0, "step_tail: M=%d P=%d WO=%d must be multiples of 4", d.M, d.P, d.WO);
    const size_t smem = ((size_t)d.WO + ((d.M + 1 + 3) & ~3) + (size_t)d.NS * d.P) * sizeof(float);
    T2_REQUIRE(smem <= 64 * 1024, "step_tail: row too wide for LDS (%zu bytes)", smem);
    hipLaunchKernelGGL(step_tail_kernel, dim3(d.B), dim3(NTT), smem, s, d);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
