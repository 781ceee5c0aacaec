// Dev tool: per-launch floors on MI355X for dependent short kernels and partial-line stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1; }
// each block of 256 threads writes B=64 rows x 8 floats (32 B segments), 4 gate planes, like the LSTM epilogue
__global__ void k_partial(float* out, int H, long ld) {
    const int u0 = blockIdx.x * 8, t = threadIdx.x;
    for (int m = 0; m < 2; ++m) {
        const int b = m * 32 + (t >> 3), u = u0 + (t & 7);
        float* g = out + (long)b * ld + u;
        g[0] = t; g[H] = t; g[2 * H] = t; g[3 * H] = t;
    }
}
// same bytes, full 128-B lines: block writes 64 rows x 32 contiguous floats
__global__ void k_full(float* out, int H, long ld) {
    const int c0 = blockIdx.x * 32, t = threadIdx.x;
    for (int i = t; i < 64 * 32; i += 256) {
        const int b = i >> 5, c = i & 31;
        out[(long)b * ld + c0 + c] = t;
    }
}
// dependent cold read: every thread reads one float written by the previous launch on another CU, then writes
__global__ void k_readwrite(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[(i * 97) % n] + 1.f;
}
__global__ void k_transc(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float x = in[i]; float s = 1.f / (1.f + expf(-x)); out[i] = s * tanhf(x) + tanhf(s); }
}

template <typename F> float run(F f, int iters, hipStream_t s) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f(i);
    hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) f(i);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int H = 1024, B = 64; const long ld = 4 * H;
    float *g, *x, *y;
    CK(hipMalloc(&g, sizeof(float) * B * ld * 512));   // 512 steps worth, so each launch touches fresh lines
    CK(hipMalloc(&x, sizeof(float) * (1 << 24))); CK(hipMalloc(&y, sizeof(float) * (1 << 24)));
    CK(hipMemset(x, 0, sizeof(float) * (1 << 24)));
    const int it = 400;
    printf("empty            %.2f us/launch\n", run([&](int i) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, g); }, it, s));
    printf("partial 32B fresh %.2f us/launch (1 MB as 32-B segments)\n", run([&](int i) { hipLaunchKernelGGL(k_partial, dim3(H / 8), dim3(256), 0, s, g + (long)(i % 512) * B * ld, H, ld); }, it, s));
    printf("partial 32B same  %.2f us/launch\n", run([&](int i) { hipLaunchKernelGGL(k_partial, dim3(H / 8), dim3(256), 0, s, g, H, ld); }, it, s));
    printf("full lines fresh  %.2f us/launch (1 MB as 128-B lines)\n", run([&](int i) { hipLaunchKernelGGL(k_full, dim3(4 * H / 32), dim3(256), 0, s, g + (long)(i % 512) * B * ld, H, ld); }, it, s));
    printf("full lines same   %.2f us/launch\n", run([&](int i) { hipLaunchKernelGGL(k_full, dim3(4 * H / 32), dim3(256), 0, s, g, H, ld); }, it, s));
    const int n = 65536;
    printf("dependent read+write 64k elems (ping-pong) %.2f us/launch\n", run([&](int i) { hipLaunchKernelGGL(k_readwrite, dim3(n / 256), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, n); }, it, s));
    printf("transcendental 64k elems %.2f us/launch\n", run([&](int i) { hipLaunchKernelGGL(k_transc, dim3(n / 256), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, n); }, it, s));
    printf("empty 512-thread blocks with 100KB LDS: ");
    return 0;
}
