// Gradient-norm clipping + Adam in two passes over all parameter tensors (train.py:322-330 of the reference:
// torch.nn.utils.clip_grad_norm_ followed by torch.optim.Adam.step, weight decay added to the gradient).
//   pass 1  sum of squares of every gradient, fixed-order two-stage reduction -> total norm, clip coefficient on the device
//   pass 2  g' = clip * g + wd * p ; m = lerp(m, g', 1-b1) ; v = b2 v + (1-b2) g'^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// One "multi-tensor" launch per pass: a table of (p, g, m, v, n, first chunk) rows in device memory, workgroup -> chunk.
// The torch foreach path makes ~10 passes over the 52 M parameters (1.4 ms + 0.5 ms for the norm); this is HBM-bound
// streaming: 28 B per parameter.
#include "kernels.h"

namespace t2 {

namespace {

constexpr int kChunk = 8192;          // elements per workgroup

__device__ __forceinline__ int find_tensor(const AdamTensor* __restrict__ tab, int n, int chunk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                  // last row with first_chunk <= chunk
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const AdamTensor* __restrict__ tab, int n, float* __restrict__ partial) {
    const int chunk = blockIdx.x;
    const AdamTensor t = tab[find_tensor(tab, n, chunk)];
    const long off = (long)(chunk - t.first_chunk) * kChunk;
    const long cnt = min((long)kChunk, t.numel - off);
    const float* g = t.g + off;
    float s = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        for (long i = threadIdx.x * 4; i + 3 < cnt; i += 256 * 4) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(g + i);
            s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        }
        for (long i = (cnt & ~3l) + threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
    } else {
        for (long i = threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total norm, out[1] = clip coefficient = min(1, max_norm / (norm + 1e-6))   (max_norm <= 0: no clipping),
// out[2] = 1 when the update must be skipped
__global__ __launch_bounds__(1024) void norm_finish_kernel(const float* __restrict__ partial, int n, float max_norm, float* __restrict__ out,
                                                           const unsigned* __restrict__ sticky) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 16; ++i) tot += red[i];
        const float norm = (float)sqrt(tot);
        out[0] = norm;
        out[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
        // an aborted persistent kernel (chain_common.h) left invalid gradients behind: no parameter may move
        out[2] = (sticky && __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) ? 1.0f : 0.0f;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamTensor* __restrict__ tab, int n, const float* __restrict__ coef,
                                                   float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2, float eps, float wd) {
    const int chunk = blockIdx.x + tab[0].first_chunk;       // a sub-table (rows of one step count) starts at its own first chunk
    const AdamTensor t = tab[find_tensor(tab, n, chunk)];
    const long off = (long)(chunk - t.first_chunk) * kChunk;
    const long cnt = min((long)kChunk, t.numel - off);
    const float clip = coef ? coef[1] : 1.0f;
    if (coef && coef[2] != 0.0f) return;                      // gradients invalid (norm_finish_kernel): leave p, m, v untouched
    float* p = t.p + off; const float* g = t.g + off; float* m = t.m + off; float* v = t.v + off;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg = gg * clip + wd * pp;
        mm = mm + (1.0f - b1) * (gg - mm);                     // lerp, as torch's foreach Adam
        vv = vv * b2 + (1.0f - b2) * gg * gg;
        pp -= lr_over_bc1 * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
    };
    const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    long done = 0;
    if (al) {
        for (long i = threadIdx.x * 4; i + 3 < cnt; i += 256 * 4) {
            f32x4 pp = *reinterpret_cast<f32x4*>(p + i), mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float ps = pp[c], ms = mm[c], vs = vv[c];
                upd(ps, gg[c], ms, vs);
                pp[c] = ps; mm[c] = ms; vv[c] = vs;
            }
            *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(m + i) = mm; *reinterpret_cast<f32x4*>(v + i) = vv;
        }
        done = cnt & ~3l;
    }
    for (long i = done + threadIdx.x; i < cnt; i += 256) upd(p[i], g[i], m[i], v[i]);
}

}  // namespace

int adam_chunks(long numel) { return (int)((numel + kChunk - 1) / kChunk); }

int adam_norm(const AdamTensor* table_dev, int n_tensors, int n_chunks, float* partial, float* norm_out, float max_norm, hipStream_t s) {
    T2_REQUIRE(table_dev && n_tensors >= 1 && n_chunks >= 1 && partial && norm_out, "adam_norm: bad arguments");
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(256), 0, s, table_dev, n_tensors, partial);
    hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(1024), 0, s, partial, n_chunks, max_norm, norm_out, chain_sticky_words());
    T2_LAUNCH_CHECK();
    return 0;
}

// max_norm >= 0: norm + clip coefficient over this table, then the update.  max_norm < 0: update only, with the
// coefficient adam_norm left in norm_out[1]; the table may then be a run of rows of a larger table (n_chunks = its chunks).
int adam_step(const AdamTensor* table_dev, int n_tensors, int n_chunks, float* partial, float* norm_out, float max_norm,
              float lr, float b1, float b2, float eps, float wd, int step, hipStream_t s) {
    T2_REQUIRE(table_dev && n_tensors >= 1 && n_chunks >= 1 && partial && norm_out && step >= 1, "adam_step: bad arguments");
    if (max_norm >= 0.f) T2_TRY_RC(adam_norm(table_dev, n_tensors, n_chunks, partial, norm_out, max_norm, s));
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, s, table_dev, n_tensors, norm_out,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), b1, b2, eps, wd);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
