// Small bandwidth-bound helpers: RNG export (parity tests replay exactly the bits the fused
// kernels draw), layout shuffles at the model boundary, padding masks.
#include "kernels.h"

namespace t2 {

namespace {

inline int grid_for(size_t n, int block = 256, int cap = 4096) {
    size_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

__global__ void rng_keep_mask_kernel(RngKey k, uint32_t n, float p, uint8_t* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = rng_keep(k, i, p) ? 1 : 0;
}
__global__ void rng_normal_kernel(RngKey k, uint32_t n, float* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = rng_normal(k, i);
}

// X[b,t,m] = t == 0 ? 0 : mel[b,m,t-1]      (Decoder.forward, model.py:407-411)
__global__ void teacher_inputs_kernel(const float* __restrict__ mel, float* __restrict__ X, int B, int M, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int m = m0 + i, ts = t0 + tx - 1;                  // source frame for output t = t0+tx
        float v = 0.f;
        if (m < M && ts >= 0 && ts < T) v = mel[((long)b * M + m) * T + ts];
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, m = m0 + tx;
        if (t < T && m < M) X[((long)b * T + t) * M + m] = tile[tx][i];
    }
}

// out[b,c,t] = t < len[b] ? in[b,t,c] : fill
__global__ void transpose_btc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int T, int C,
                                     const int* __restrict__ lengths, float fill) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (t < T && c < C) ? in[((long)b * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    const int len = lengths ? lengths[b] : T;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < T) out[((long)b * C + c) * T + t] = t < len ? tile[tx][i] : fill;
    }
}

__global__ void mask_bt_kernel(float* x, int B, int T, const int* __restrict__ lengths, float fill) {
    const long n = (long)B * T;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / T), t = (int)(i % T);
        if (t >= lengths[b]) x[i] = fill;
    }
}

__global__ void fill_kernel(float* p, float v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace

int rng_keep_mask(uint64_t seed, uint32_t site, uint32_t n, float p, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(rng_keep_mask_kernel, dim3(grid_for(n)), dim3(256), 0, s, rng_key(seed, site), n, p, out);
    T2_LAUNCH_CHECK();
    return 0;
}
int rng_normal(uint64_t seed, uint32_t site, uint32_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(rng_normal_kernel, dim3(grid_for(n)), dim3(256), 0, s, rng_key(seed, site), n, out);
    T2_LAUNCH_CHECK();
    return 0;
}
int teacher_inputs(const float* mel, float* X, int B, int M, int T, hipStream_t s) {
    dim3 grid((T + 31) / 32, (M + 31) / 32, B);
    hipLaunchKernelGGL(teacher_inputs_kernel, grid, dim3(256), 0, s, mel, X, B, M, T);
    T2_LAUNCH_CHECK();
    return 0;
}
int transpose_btc_to_bct(const float* in, float* out, int B, int T, int C, const int* lengths, float fill, hipStream_t s) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(transpose_btc_kernel, grid, dim3(256), 0, s, in, out, B, T, C, lengths, fill);
    T2_LAUNCH_CHECK();
    return 0;
}
int mask_bt(float* x, int B, int T, const int* lengths, float fill, hipStream_t s) {
    hipLaunchKernelGGL(mask_bt_kernel, dim3(grid_for((size_t)B * T)), dim3(256), 0, s, x, B, T, lengths, fill);
    T2_LAUNCH_CHECK();
    return 0;
}
int fill_f32(float* p, float v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, v, n);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
