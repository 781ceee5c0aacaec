// Shared device helpers of the persistent chain kernels (chain.hip, chain_bwd.hip): the hand-off protocol of
// MI355X_MICROARCH.md "Valid forms" (write-through payload stores drained by every storing wave, workgroup barrier,
// one lane signals with an agent-scope atomic; one wave polls with relaxed agent-scope loads, the other waves load
// behind the workgroup barrier, every payload load sc1) and the hardware-exp activations.
#pragma once
#include "kernels.h"

namespace t2 {
namespace chain {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define T2_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NTH = 512, NWV = 8;
constexpr int SC1 = 16;                    // buffer-instruction cache policy bit: sc1 (agent scope, bypass L1 / write through)
constexpr int PPR = 40;                    // LDS pitch of a 32-wide partial tile row (conflict-free fixed-order reads)
constexpr int CNT_STRIDE = 32;             // one arrival counter per 128-byte line
constexpr unsigned long long SPIN_TICKS = 100000000ull;   // 1 s of the 100 MHz realtime counter: every spin is bounded

__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - __fdividef(2.0f, __expf(2.0f * x) + 1.0f); }
__device__ __forceinline__ float fast_sigmoid(float x) { return __fdividef(1.0f, 1.0f + __expf(-x)); }

// one wave polls one counter; all lanes read the same word (a single request), so the branch is wave-uniform
__device__ __forceinline__ bool poll_counter(const unsigned* cnt, unsigned want, unsigned* err, unsigned code) {
    if (want == 0) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = __hip_atomic_load(cnt, T2_RLX_AGENT);
        if (v >= want) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TICKS) {
            if ((threadIdx.x & 63) == 0) atomicMax(err, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// two counters in one request: lane 0 reads cnt0, every other lane cnt1
__device__ __forceinline__ bool poll_counters2(const unsigned* cnt0, unsigned want0, const unsigned* cnt1, unsigned want1, unsigned* err, unsigned code) {
    const bool first = (threadIdx.x & 63) == 0;
    const unsigned* p = first ? cnt0 : cnt1;
    const unsigned want = first ? want0 : want1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = __hip_atomic_load(p, T2_RLX_AGENT);
        if (__all(v >= want)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TICKS) {
            if (first) atomicMax(err, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// after the payload stores of every wave: drain (every storing wave), workgroup barrier, one lane signals
__device__ __forceinline__ void publish(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, T2_RLX_AGENT);
}

}  // namespace chain
}  // namespace t2
