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
constexpr int NSH = kCntShards;            // shards of an arrival counter
constexpr int CNT_LINE = 32;               // unsigned words per 128-byte line
constexpr int CNT_STRIDE = NSH * CNT_LINE; // words per (sharded) arrival counter
constexpr unsigned long long SPIN_TICKS = 100000000ull;   // 1 s of the 100 MHz realtime counter: every spin is bounded

// Abort reports go to two places: the status word of the pass (in its workspace: which chain of which pass) and the
// device's STICKY status word, which lives in page-locked host memory (c_api.hip: chain_sticky_words) — the host reads it
// with a plain load at every entry point and at its own synchronisation points, and the optimizer kernels read it before
// they touch a parameter.  One copy of the pointer per translation unit, set by persistent_prepare() below.
static __device__ unsigned* t2_sticky_dev = nullptr;
__device__ __forceinline__ void report_abort(unsigned* err, unsigned code) {
    atomicMax(err, code);
    unsigned* sp = t2_sticky_dev;
    if (sp) __hip_atomic_store(sp, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - __fdividef(2.0f, __expf(2.0f * x) + 1.0f); }
__device__ __forceinline__ float fast_sigmoid(float x) { return __fdividef(1.0f, 1.0f + __expf(-x)); }

// Arrival counters are SHARDED: producer i of a group adds to shard i % NSH, each shard on a 128-byte line of its own; the
// polling wave reads all shards in one request (lane -> shard lane % NSH) and is satisfied when every shard has the arrivals
// of `steps` steps.  Why: with one word per group the 64-128 adds of a step serialise on one line while 64-128 workgroups
// poll it; scripts/persist_probe.hip (chain A's traffic pattern, no arithmetic): 9.7 us per two-hop step with one word,
// 8.1 with 4 shards, 7.1 with 16 (replicas on top: 7.0).
// producers of shard sh among `P` producers numbered 0..P-1
__device__ __forceinline__ unsigned shard_share(unsigned P, unsigned sh) { return (P + (unsigned)NSH - 1u - sh) / (unsigned)NSH; }
// one wave polls one counter: `steps` steps of `P` producers each
__device__ __forceinline__ bool poll_counter(const unsigned* cnt, unsigned steps, unsigned P, unsigned* err, unsigned code) {
    if (steps == 0) return true;
    const unsigned sh = threadIdx.x & (NSH - 1);
    const unsigned* p = cnt + sh * CNT_LINE;
    const unsigned want = steps * shard_share(P, sh);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = __hip_atomic_load(p, T2_RLX_AGENT);
        if (__all(v >= want)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TICKS) {
            if ((threadIdx.x & 63) == 0) report_abort(err, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// two counters in one request: lanes 0..31 read cnt0's shards, lanes 32..63 cnt1's
__device__ __forceinline__ bool poll_counters2(const unsigned* cnt0, unsigned steps0, unsigned P0, const unsigned* cnt1, unsigned steps1, unsigned P1,
                                               unsigned* err, unsigned code) {
    const bool first = (threadIdx.x & 63) < 32;
    const unsigned sh = threadIdx.x & (NSH - 1);
    const unsigned* p = (first ? cnt0 : cnt1) + sh * CNT_LINE;
    const unsigned want = (first ? steps0 : steps1) * shard_share(first ? P0 : P1, sh);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = __hip_atomic_load(p, T2_RLX_AGENT);
        if (__all(v >= want)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TICKS) {
            if ((threadIdx.x & 63) == 0) report_abort(err, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// after the payload stores of every wave: drain (every storing wave), workgroup barrier, one lane signals (producer `pidx` of its group)
__device__ __forceinline__ void publish(unsigned* cnt, unsigned pidx) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + (pidx & (NSH - 1)) * CNT_LINE, 1u, T2_RLX_AGENT);
}

// ---- tagged hand-offs (round 3) ------------------------------------------------------------------------------------------
// First form (kept for reference: publish_hint): counters as hints + validated payloads.  Final form, every teacher-forced chain:
// NO counter at all — the consumers load their operands until the tags below are this step's (see chain.hip / chain_bwd.hip).
// The producer's s_waitcnt vmcnt(0) in front of its arrival add costs the consumers the store round trip of the slowest
// producer.  Here the add goes out right behind the stores (a HINT), and the payload validates itself: bit 0 of every stored
// 32-bit word (fp32 payloads) or of the first word of every 16-byte unit (bf16 fragments) is a tag that flips each time a
// slot is rewritten (slots alternate by step parity, so the tag is bit 1 of the step count, inverted: buffers are zeroed per
// launch and the first valid tag is 1).  A consumer whose loads still show the old tag loads again (bounded).  16-byte sc1
// stores have been observed untorn on gfx950 (MI355X_MICROARCH.md, Valid forms; scripts/persist_probe: 0 torn units).
__device__ __forceinline__ unsigned step_tag(unsigned steps_done) { return ((steps_done >> 1) & 1u) ^ 1u; }
__device__ __forceinline__ void publish_hint(unsigned* cnt, unsigned pidx) {
    __syncthreads();                                         // every wave has ISSUED its payload stores
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + (pidx & (NSH - 1)) * CNT_LINE, 1u, T2_RLX_AGENT);
}
__device__ __forceinline__ float tag_f32(float v, unsigned tag) { return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v) & ~1u) | tag); }

}  // namespace chain

// Host side, before every persistent launch: dynamic-LDS attribute (cached per device and kernel), the sticky status
// pointer of this translation unit, and co-residency — a persistent grid makes progress only if ALL its workgroups are
// resident at once, so a launch whose grid exceeds (workgroups the kernel fits per CU) x (CUs) is refused here instead of
// spinning into its 1 s time-outs on the device.
template <class K>
static int persistent_prepare(K kernel, int grid, size_t smem) {
    const void* fn = reinterpret_cast<const void*>(kernel);
    T2_TRY_RC(t2_allow_dynamic_lds(fn, smem));
    int dev = 0;
    T2_CHECK_HIP(hipGetDevice(&dev));
    static std::mutex mu;
    static std::vector<std::tuple<int, const void*, size_t, int>> seen;       // (device, kernel, LDS bytes) -> workgroups per CU
    static bool sticky_set[16] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (!sticky_set[dev & 15]) {
        unsigned* sp = chain_sticky_words();
        T2_REQUIRE(sp, "persistent launch: no status block for device %d", dev);
        T2_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(chain::t2_sticky_dev), &sp, sizeof(sp)));
        sticky_set[dev & 15] = true;
    }
    int per_cu = -1;
    for (auto& e : seen)
        if (std::get<0>(e) == dev && std::get<1>(e) == fn && std::get<2>(e) == smem) per_cu = std::get<3>(e);
    if (per_cu < 0) {
        T2_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, chain::NTH, smem));
        seen.emplace_back(dev, fn, smem, per_cu);
    }
    const int cus = chain_device_cus();
    T2_REQUIRE(per_cu >= 1 && (long)per_cu * cus >= grid,
               "persistent launch refused: %d workgroups cannot be co-resident (%d per CU x %d CUs with %zu bytes of LDS)", grid, per_cu, cus, smem);
    return 0;
}

}  // namespace t2
