// Dev tool: hand-off latencies of the persistent decoder-chain design on MI355X, measured with the exact traffic
// pattern of one chain-A step at B = 64 (csrc/chain.hip): 256 workgroups x 512 threads, one per CU;
//   L phase: poll 64 h-flags + 64 ctx-flags, load 12 x 1 KB activation fragments per wave (sc1), publish 1 KB of h
//            fragments + 16 KB of query partials (sc1), drain, barrier, flag
//   A phase: poll 64 h-flags, load 64 x 512 B query partials (sc1), publish 32 x 16 B context pieces, drain, barrier, flag
// Every payload word carries its step tag, every load is checked: a stale read shows up in the error count.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/persist_probe scripts/persist_probe.hip && scripts/persist_probe [steps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int NTH = 512, NWG = 256, KT = 96, SC1 = 16;
constexpr unsigned long long SPIN_TICKS = 50000000ull;      // 0.5 s of the 100 MHz realtime counter

struct Args {
    unsigned char* X;       // [2 parity][2 s][KT][2 rt][64 lanes][16 B]
    unsigned char* Q;       // [2 s][2 rt][64 ug][32 rows][512 B]
    unsigned* flagH;        // [2 s][2 rt][64]
    unsigned* flagC;        // [2 s][2 rt][64]
    unsigned* err;          // [0] stale words, [1] timeouts
    unsigned long long* stamps;   // [NWG][8] accumulated ticks per segment
    int steps; int qstores; int sleep; int mode; int nodrain;   // nodrain 1: no vmcnt(0) before the signal; consumers validate the tags of what they loaded and re-load until valid (the counter is a hint only)
    unsigned* retries; int shards; int reps;      // mode 1: counter of a group split in `shards` shards (producer idx % shards adds to one), kept in `reps` replicas (every producer adds to all, a consumer polls replica wg % reps)   // mode 0: one flag word per producer (64 polled per wave); 1: one counter per group, 64 atomic arrivals
};

// group g: 0..3 = h flags of (s, rt), 4..7 = ctx flags of (s, rt).  mode 0: F[g][64] words, mode 1: one counter per group on its own line
__device__ __forceinline__ bool poll_ge(const unsigned* F, int g, unsigned step, int lane, unsigned* err, int slp, int mode, int shards, int reps, int rep) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned* p = mode ? F + 8 * 64 + ((g * reps + rep) * shards + lane % shards) * 32 : F + g * 64 + lane;
    const unsigned want = mode ? step * (64 / shards) : step;
    for (;;) {
        const unsigned v = __hip_atomic_load(p, RLX_AGENT);
        if (__all(v >= want)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TICKS) { if (lane == 0) atomicAdd(err + 1, 1u); return false; }
        for (int i = 0; i < slp; ++i) __builtin_amdgcn_s_sleep(1);
    }
}
// called by the first `reps` lanes of wave 0 (mode 1) or by lane 0 (mode 0)
__device__ __forceinline__ void signal(unsigned* F, int g, int idx, unsigned step, int mode, int shards, int reps, int lane) {
    if (mode) __hip_atomic_fetch_add(F + 8 * 64 + ((g * reps + lane) * shards + idx % shards) * 32, 1u, RLX_AGENT);
    else __hip_atomic_store(F + g * 64 + idx, step, RLX_AGENT);
}

__global__ __launch_bounds__(NTH) void probe(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* abortw = reinterpret_cast<unsigned*>(smem);
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int s = wg / 128, ug = (wg % 128) / 2, rt = wg % 2;         // L item
    const int b = (wg % 128) / 2, half = wg % 2, art = b / 32;        // A item (same stream)
    if (tid == 0) *abortw = 0;
    __syncthreads();
    const size_t xs = (size_t)KT * 2 * 1024;                          // one stream of one parity
    auto rsX = __builtin_amdgcn_make_buffer_rsrc(a.X, 0, (int)(4 * xs), 0x00020000);
    auto rsQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, 2 * 2 * 64 * 32 * 512, 0x00020000);
    unsigned bad = 0;
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < a.steps; ++t) {
        // ------------------------------------------------------------------ L(t)
        unsigned long long c0 = __builtin_amdgcn_s_memrealtime();
        if (wave == 0) {
            bool ok = poll_ge(a.flagH, s * 2 + rt, (unsigned)t, lane, a.err, a.sleep, a.mode, a.shards, a.reps, wg % a.reps);
            ok = ok && poll_ge(a.flagH, 4 + s * 2 + rt, (unsigned)t, lane, a.err, a.sleep, a.mode, a.shards, a.reps, wg % a.reps);
            if (!ok && lane == 0) *abortw = 1;
        }
        __syncthreads();
        if (*abortw) return;
        unsigned long long c1 = __builtin_amdgcn_s_memrealtime();
        const unsigned xin = (unsigned)((((t + 1) & 1) * 2 + s) * xs);      // parity of step t-1
        u32x4 v[12];
        for (int pass = 0;; ++pass) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int kt = i < 8 ? wave * 8 + i : 64 + wave * 4 + (i - 8);
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xin + (unsigned)((kt * 2 + rt) * 1024 + lane * 16), 0, SC1);
            }
            if (!a.nodrain) break;
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 12; ++i) ok &= v[i].x == (unsigned)t;          // (the tag word of each 16-byte unit)
            if (__all(ok) || pass > 100000) break;
            if (lane == 0) atomicAdd(a.retries, 1u);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) bad += (v[i].x != (unsigned)t) + (v[i].y != (unsigned)t) + (v[i].z != (unsigned)t) + (v[i].w != (unsigned)t);
        unsigned long long c2 = __builtin_amdgcn_s_memrealtime();
        const unsigned tag = (unsigned)t + 1;
        const u32x4 tv = {tag, tag, tag, tag};
        const unsigned xout = (unsigned)(((t & 1) * 2 + s) * xs);
        if (wave == 0) __builtin_amdgcn_raw_buffer_store_b128(tv, rsX, xout + (unsigned)((ug * 2 + rt) * 1024 + lane * 16), 0, SC1);
        const unsigned qb = (unsigned)((((s * 2 + rt) * 64 + ug) * 32) * 512);
        // qstores: 1 KB stores per wave (2 = the full 16 KB of fp32 partials; 1 = 8 KB; 0 = none: consumers then see stale tags, ignored)
        for (int i = 0; i < a.qstores; ++i) __builtin_amdgcn_raw_buffer_store_b128(tv, rsQ, qb + (unsigned)((wave * 2 + i) * 1024 + lane * 16), 0, SC1);
        if (!a.nodrain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < (a.mode ? a.reps : 1)) signal(a.flagH, s * 2 + rt, ug, tag, a.mode, a.shards, a.reps, tid);
        unsigned long long c3 = __builtin_amdgcn_s_memrealtime();
        // ------------------------------------------------------------------ A(t)
        if (wave == 0) {
            const bool ok = poll_ge(a.flagH, s * 2 + art, tag, lane, a.err, a.sleep, a.mode, a.shards, a.reps, wg % a.reps);
            if (!ok && lane == 0) *abortw = 1;
        }
        __syncthreads();
        if (*abortw) return;
        unsigned long long c4 = __builtin_amdgcn_s_memrealtime();
        u32x4 q[4];
        for (int pass = 0;; ++pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pu = wave * 8 + i * 2 + (lane >> 5);                 // partial (unit group) index
                q[i] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (unsigned)((((s * 2 + art) * 64 + pu) * 32 + (b & 31)) * 512 + (lane & 31) * 16), 0, SC1);
            }
            if (!a.nodrain || a.qstores != 2) break;
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) ok &= q[i].x == tag;
            if (__all(ok) || pass > 100000) break;
            if (lane == 0) atomicAdd(a.retries, 1u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bad += a.qstores == 2 ? (q[i].x != tag) + (q[i].y != tag) + (q[i].z != tag) + (q[i].w != tag) : (q[i].x > tag);
        unsigned long long c5 = __builtin_amdgcn_s_memrealtime();
        if (wave == 0 && lane < 32) {
            const int piece = half * 32 + lane;                            // 8-column piece of the context row
            const int kt = 64 + piece / 2, hk = piece & 1;
            __builtin_amdgcn_raw_buffer_store_b128(tv, rsX, xout + (unsigned)((kt * 2 + art) * 1024 + (hk * 32 + (b & 31)) * 16), 0, SC1);
        }
        if (!a.nodrain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < (a.mode ? a.reps : 1)) signal(a.flagH, 4 + s * 2 + art, (b & 31) * 2 + half, tag, a.mode, a.shards, a.reps, tid);
        unsigned long long c6 = __builtin_amdgcn_s_memrealtime();
        seg[0] += c1 - c0; seg[1] += c2 - c1; seg[2] += c3 - c2; seg[3] += c4 - c3; seg[4] += c5 - c4; seg[5] += c6 - c5;
    }
    if (bad) atomicAdd(a.err, bad);
    if (tid == 0)
        for (int i = 0; i < 6; ++i) a.stamps[wg * 8 + i] = seg[i];
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    Args a{};
    const size_t xbytes = (size_t)4 * KT * 2 * 1024, qbytes = (size_t)2 * 2 * 64 * 32 * 512;
    CK(hipMalloc(&a.X, xbytes)); CK(hipMalloc(&a.Q, qbytes));
    CK(hipMalloc(&a.flagH, 1 << 20)); CK(hipMalloc(&a.flagC, 4 * 64 * 4)); CK(hipMalloc(&a.err, 16));
    CK(hipMalloc(&a.stamps, NWG * 8 * 8));
    a.steps = steps; a.qstores = argc > 2 ? atoi(argv[2]) : 2; a.sleep = argc > 3 ? atoi(argv[3]) : 1; a.mode = argc > 4 ? atoi(argv[4]) : 0;
    a.nodrain = argc > 5 ? atoi(argv[5]) : 0; a.shards = argc > 6 ? atoi(argv[6]) : 1; a.reps = argc > 7 ? atoi(argv[7]) : 1;
    printf("shards %d replicas %d\n", a.shards, a.reps);
    CK(hipMalloc(&a.retries, 16));
    printf("qstores %d sleep %d mode %d nodrain %d\n", a.qstores, a.sleep, a.mode, a.nodrain);
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t lds = 140 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemsetAsync(a.X, 0, xbytes, st)); CK(hipMemsetAsync(a.Q, 0, qbytes, st));
        CK(hipMemsetAsync(a.flagH, 0, 1 << 20, st)); CK(hipMemsetAsync(a.flagC, 0, 4 * 64 * 4, st));
        CK(hipMemsetAsync(a.err, 0, 16, st)); CK(hipMemsetAsync(a.stamps, 0, NWG * 8 * 8, st)); CK(hipMemsetAsync(a.retries, 0, 16, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(probe, dim3(NWG), dim3(NTH), lds, st, a);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err[4]; CK(hipMemcpy(err, a.err, 16, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> sp(NWG * 8);
        CK(hipMemcpy(sp.data(), a.stamps, NWG * 8 * 8, hipMemcpyDeviceToHost));
        double seg[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < NWG; ++w) for (int i = 0; i < 6; ++i) seg[i] += (double)sp[w * 8 + i];
        unsigned rt_[4]; CK(hipMemcpy(rt_, a.retries, 16, hipMemcpyDeviceToHost));
        printf("rep %d: %d steps in %.3f ms = %.2f us/step; stale words %u, timeouts %u, re-load passes %u\n", rep, steps, ms, 1e3 * ms / steps, err[0], err[1], rt_[0]);
        const char* names[6] = {"L poll", "L load+check", "L publish+flag", "A poll", "A load+check", "A publish+flag"};
        for (int i = 0; i < 6; ++i) printf("   %-16s %.2f us (mean over workgroups)\n", names[i], seg[i] / NWG / steps / 100.0);
    }
    return 0;
}
