// Shared device helpers for the gfx950 (MI355X / CDNA4) Tacotron2 kernels.
// Wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace t2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------------
// Counter-based RNG (dropout keep-masks, SMA pre-sigmoid noise).  A value depends only on
// (seed, site, index) so the backward pass regenerates a mask instead of storing it, and the
// parity tests can export exactly the bits the kernels use (t2_rng_keep_mask / t2_rng_normal).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct RngKey { uint32_t k0, k1; };
__host__ __device__ __forceinline__ RngKey rng_key(uint64_t seed, uint32_t site) {
    RngKey k;
    k.k0 = mix32((uint32_t)seed ^ mix32(site * 0x9E3779B9u + 0x85ebca6bu));
    k.k1 = mix32((uint32_t)(seed >> 32) + site * 0xc2b2ae35u + 0x27d4eb2fu);
    return k;
}
__host__ __device__ __forceinline__ uint32_t rng_u32(RngKey k, uint32_t idx) {
    return mix32((idx ^ k.k0) * 0x9E3779B1u + k.k1);
}
// uniform in [0,1) with 24 bits
__host__ __device__ __forceinline__ float rng_uniform(RngKey k, uint32_t idx) {
    return (float)(rng_u32(k, idx) >> 8) * (1.0f / 16777216.0f);
}
__host__ __device__ __forceinline__ bool rng_keep(RngKey k, uint32_t idx, float p) {
    return rng_uniform(k, idx) >= p;
}
// standard normal via Box-Muller on two decorrelated draws of the same index
__device__ __forceinline__ float rng_normal(RngKey k, uint32_t idx) {
    uint32_t a = rng_u32(k, idx);
    uint32_t b = mix32(a ^ 0x68bc21ebu) + k.k0;
    float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
    float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// ---------------------------------------------------------------------------------------------
// activations (accurate forms: the 1e-4 parity contract is on 400 recurrent steps)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace t2

// ---------------------------------------------------------------------------------------------
// host-side error plumbing for the C ABI
// ---------------------------------------------------------------------------------------------
void t2_set_error(const char* fmt, ...);
#define T2_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            t2_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return -2;                                                                       \
        }                                                                                    \
    } while (0)
#define T2_REQUIRE(cond, ...)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            t2_set_error(__VA_ARGS__);                                                       \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)
#define T2_LAUNCH_CHECK() T2_CHECK_HIP(hipGetLastError())
#define T2_TRY_RC(expr)             \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != 0) return rc__; \
    } while (0)

// Dynamic LDS above 64 KB needs hipFuncSetAttribute once per (device, kernel) (and again only for a larger size): the call
// costs the host tens of microseconds, and the per-step kernels are launched hundreds of times per pass.
#include <mutex>
#include <tuple>
#include <vector>
inline int t2_allow_dynamic_lds(const void* fn, size_t smem) {
    if (smem <= 64 * 1024) return 0;
    int dev = 0;
    T2_CHECK_HIP(hipGetDevice(&dev));
    static std::mutex mu;
    static std::vector<std::tuple<int, const void*, size_t>> done;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : done)
        if (std::get<0>(e) == dev && std::get<1>(e) == fn) {
            if (std::get<2>(e) >= smem) return 0;
            T2_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            std::get<2>(e) = smem;
            return 0;
        }
    T2_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    done.emplace_back(dev, fn, smem);
    return 0;
}
