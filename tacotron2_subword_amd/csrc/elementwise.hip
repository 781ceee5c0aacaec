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

// X[t,b,m] = t == 0 ? 0 : mel[b,m,t-1]      (Decoder.forward, model.py:407-411), time-major output
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
        if (t < T && m < M) X[((long)t * B + b) * M + m] = tile[tx][i];
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

__global__ void mask_btc_kernel(float* x, int B, int T, int C, const int* __restrict__ lengths, float fill) {
    const size_t n = (size_t)B * T * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C;
        if ((int)(r % T) >= lengths[r / T]) x[i] = fill;
    }
}

__global__ void fill_kernel(float* p, float v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void permute_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int R1, int R2, int W) {
    const size_t n = (size_t)R1 * R2 * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const size_t r = i / W;
        const int r1 = (int)(r % R1), r2 = (int)(r / R1);          // output row index = r2*R1 + r1
        out[i] = in[((size_t)r1 * R2 + r2) * W + w];
    }
}

__global__ void relu_drop_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* dz, float scale, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dz[i] = y[i] > 0.f ? dy[i] * scale : 0.f;
}

// stage 1: block (x = column tile of 64, y = row slab) sums its slab; stage 2 sums the slabs in order
__global__ void colsum_stage1_kernel(const float* __restrict__ X, long ld, int M, int N, int slabs, float* scratch) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;                                  // 4 row phases per block
    const int rows = (M + slabs - 1) / slabs;
    const int m0 = blockIdx.y * rows, m1 = min(M, m0 + rows);
    __shared__ float part[4][64];
    float acc = 0.f;
    if (n < N) for (int m = m0 + sub; m < m1; m += 4) acc += X[(long)m * ld + n];
    part[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sub == 0 && n < N) scratch[(long)blockIdx.y * N + n] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
// sum_{s < count} p[s * stride], terms requested 16 at a time and added in index order (a dependent load per term
// costs ~0.3 us each: 20 us for 64 slabs)
__device__ __forceinline__ float ordered_sum16(const float* __restrict__ p, int count, long stride) {
    float acc = 0.f;
    int s = 0;
    for (; s + 16 <= count; s += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[(long)(s + j) * stride];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j];
    }
    for (; s < count; ++s) acc += p[(long)s * stride];
    return acc;
}
__global__ void colsum_stage2_kernel(const float* __restrict__ scratch, int N, int slabs, float* out, float* out2) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float acc = ordered_sum16(scratch + n, slabs, N);
    out[n] = acc;
    if (out2) out2[n] = acc;
}
__global__ void fold_halves_kernel(float* __restrict__ X, size_t rows, int A) {
    const size_t n = rows * A;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float* r = X + (i / A) * 2 * A + (i % A);
        r[0] += r[A];
    }
}
__global__ void batch_sum_kernel(const float* __restrict__ X, int B, int n, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = ordered_sum16(X + i, B, n);
}

}  // namespace

int permute_rows(const float* in, float* out, int R1, int R2, int W, hipStream_t s) {
    const size_t n = (size_t)R1 * R2 * W;
    hipLaunchKernelGGL(permute_rows_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, R1, R2, W);
    T2_LAUNCH_CHECK();
    return 0;
}
int relu_drop_bwd(const float* dy, const float* y, float* dz, float scale, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, dy, y, dz, scale, n);
    T2_LAUNCH_CHECK();
    return 0;
}
int colsum(const float* X, long ld, int M, int N, float* out, float* out2, float* scratch, hipStream_t s) {
    const int slabs = M >= 64 * 64 ? 64 : (M >= 64 ? M / 64 : 1);
    hipLaunchKernelGGL(colsum_stage1_kernel, dim3((N + 63) / 64, slabs), dim3(256), 0, s, X, ld, M, N, slabs, scratch);
    T2_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((N + 255) / 256), dim3(256), 0, s, scratch, N, slabs, out, out2);
    T2_LAUNCH_CHECK();
    return 0;
}
int fold_halves(float* X, size_t rows, int A, hipStream_t s) {
    hipLaunchKernelGGL(fold_halves_kernel, dim3(grid_for(rows * A)), dim3(256), 0, s, X, rows, A);
    T2_LAUNCH_CHECK();
    return 0;
}
int batch_sum(const float* X, int B, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(batch_sum_kernel, dim3((n + 255) / 256), dim3(256), 0, s, X, B, n, out);
    T2_LAUNCH_CHECK();
    return 0;
}

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
int mask_btc(float* x, int B, int T, int C, const int* lengths, float fill, hipStream_t s) {
    hipLaunchKernelGGL(mask_btc_kernel, dim3(grid_for((size_t)B * T * C)), dim3(256), 0, s, x, B, T, C, lengths, fill);
    T2_LAUNCH_CHECK();
    return 0;
}
int fill_f32(float* p, float v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, v, n);
    T2_LAUNCH_CHECK();
    return 0;
}

}  // namespace t2
