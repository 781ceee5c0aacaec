// C ABI (include/t2amd.h) and the host-side drivers that sequence the kernels of one decoder
// pass.  No device allocation, no synchronisation except where the header says so.
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <sys/file.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/t2amd.h"
#include "kernels.h"

static thread_local char g_err[512] = "";

void t2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define T2_TRY(expr)                \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != 0) return rc_;   \
    } while (0)

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

using namespace t2;

// ---------------------------------------------------------------------------------------------
// Optional in-situ kernel timing (bench.py's roofline figures): when enabled, the drivers bracket
// the per-step kernel launches with HIP events on the launch stream.  Off by default.
// ---------------------------------------------------------------------------------------------
enum ProfKind { PK_LSTM_ATT_FWD = 0, PK_ATTN_FWD, PK_LSTM_DEC_FWD, PK_ATTN_BWD, PK_LSTM_ATT_BWD_PW, PK_LSTM_ATT_BWD_GEMM,
                PK_LSTM_DEC_BWD_PW, PK_LSTM_DEC_BWD_GEMM, PK_CHAIN_A_FWD, PK_CHAIN_B_FWD, PK_CHAIN_B_BWD, PK_CHAIN_A_BWD, PK_CHAIN_DEC, PK_COUNT };
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> kind;
    size_t used = 0;
};
static Prof g_prof;

struct ProfScope {
    hipStream_t s; bool on;
    ProfScope(int kind, hipStream_t st) : s(st), on(g_prof.on && g_prof.used + 2 <= g_prof.ev.size()) {
        if (on) { g_prof.kind[g_prof.used / 2] = kind; (void)hipEventRecord(g_prof.ev[g_prof.used], s); }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s); g_prof.used += 2; }
    }
};

// ---------------------------------------------------------------------------------------------
// Side stream for the decoder-LSTM recurrence.  Teacher-forced passes have two serial chains that only meet through
// hoisted GEMMs: (A) attention LSTMs + attention, (B) decoder LSTM.  Each per-step launch is bound by what ONE CU can
// pull from L2/MALL and leaves much of the chip idle (the attention step uses B*2 workgroups, the decoder-LSTM step
// H/8), so chain B runs on its own stream one chunk of steps behind (forward) / ahead of (backward) chain A.
// The caller's stream forks at the first chunk and joins before the entry point returns.
// ---------------------------------------------------------------------------------------------
struct Side { hipStream_t s = nullptr; std::vector<hipEvent_t> ev; };
static Side g_side[16];
static int g_overlap = 1;
static int g_chain = getenv("T2_CHAIN") ? atoi(getenv("T2_CHAIN")) : 1;   // persistent chain kernels (chain.hip)
static int g_chain_bwd = getenv("T2_CHAIN_BWD") ? atoi(getenv("T2_CHAIN_BWD")) : 1;   // ... of the backward pass (chain_bwd.hip)
static int side_get(Side** out) {
    int dev = 0;
    T2_CHECK_HIP(hipGetDevice(&dev));
    Side& sd = g_side[dev & 15];
    if (!sd.s) {
        // lowest priority: chain A (the caller's stream) is the critical path, the side chain has slack
        int least = 0, greatest = 0;
        T2_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        T2_CHECK_HIP(hipStreamCreateWithPriority(&sd.s, hipStreamNonBlocking, least));
    }
    *out = &sd;
    return 0;
}
static int side_event(Side& sd, size_t i, hipEvent_t* out) {
    while (sd.ev.size() <= i) {
        hipEvent_t e;
        T2_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sd.ev.push_back(e);
    }
    *out = sd.ev[i];
    return 0;
}
// record on `from`, make `to` wait
static int stream_edge(Side& sd, size_t i, hipStream_t from, hipStream_t to) {
    hipEvent_t e;
    T2_TRY_RC(side_event(sd, i, &e));
    T2_CHECK_HIP(hipEventRecord(e, from));
    T2_CHECK_HIP(hipStreamWaitEvent(to, e, 0));
    return 0;
}

// Pinned slots + events of the decode loop's stop polling (one set per device)
struct StopPoll { hipEvent_t ev[4] = {}; int32_t* host = nullptr; };
static StopPoll g_poll[16];
static int stop_poll_get(StopPoll** out) {
    int dev = 0;
    T2_CHECK_HIP(hipGetDevice(&dev));
    StopPoll& p = g_poll[dev & 15];
    if (!p.host) {
        T2_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&p.host), 4 * sizeof(int32_t), hipHostMallocDefault));
        for (auto& e : p.ev) T2_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *out = &p;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Sticky status of the persistent kernels, one block per device in PAGE-LOCKED HOST memory that the device writes
// directly (chain_common.h report_abort) and reads in the optimizer (optim.hip): word 0 = code of the hand-off that
// timed out (0 = fine).  The host looks at it with a plain load — no copy, no event, no synchronisation — at every
// entry point of the Python layer and right after its own synchronisation points; nothing clears it but
// t2_chain_status_clear, so one aborted chain stops every later optimizer step until the caller has seen it.
// The same block carries this process's CLAIM on the device: a persistent grid needs every CU, so only one process per
// GPU may launch them (an flock on a per-device file; T2_CHAIN_FORCE=1 skips the test, e.g. for a child process whose
// parent holds the claim but is idle).
// ---------------------------------------------------------------------------------------------
struct StatusBlock { unsigned* host = nullptr; int claim = 0; };        // claim: 0 = not asked yet, 1 = held, -1 = another process holds it
static StatusBlock g_status[16];
static std::mutex g_status_mu;

namespace t2 {
unsigned* chain_sticky_words() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_status_mu);
    StatusBlock& b = g_status[dev & 15];
    if (!b.host) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
        memset(p, 0, 64);
        b.host = static_cast<unsigned*>(p);
    }
    return b.host;
}
bool chain_device_claim() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_status_mu);
    StatusBlock& b = g_status[dev & 15];
    if (b.claim == 0) {
        b.claim = 1;
        const char* force = getenv("T2_CHAIN_FORCE");
        if (!(force && atoi(force) != 0)) {
            char bus[64] = "";
            if (hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) snprintf(bus, sizeof(bus), "dev%d", dev);
            for (char* c = bus; *c; ++c) if (*c == ':' || *c == '.' || *c == '/') *c = '_';
            char path[160];
            snprintf(path, sizeof(path), "/tmp/t2amd-persistent-%s.lock", bus);
            const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
            if (fd >= 0 && flock(fd, LOCK_EX | LOCK_NB) != 0) {            // held by another process for as long as it lives
                close(fd);
                b.claim = -1;
                fprintf(stderr, "t2amd: another process runs persistent kernels on GPU %s: this process takes the per-step launch path "
                                "(one process per GPU is the contract; T2_CHAIN_FORCE=1 overrides)\n", bus);
            }                                                               // (fd stays open: the lock lives as long as the process)
        }
    }
    return b.claim > 0;
}
}  // namespace t2

__global__ void status_report_kernel(unsigned* sticky, unsigned code) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(sticky, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// tests: `workgroups` workgroups that each hold a CU's LDS (no persistent workgroup fits next to one) for `ticks` of the
// 100 MHz realtime counter — what a foreign kernel does to a persistent grid
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned hold[];
    hold[threadIdx.x] = threadIdx.x;
    // the highest vector and accumulation registers: each wave then owns a SIMD's whole register file (512 per lane), so
    // no wave of any other kernel — whatever its size — fits on a CU this workgroup sits on
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (hold[(threadIdx.x + 1) & 255] == 0xffffffffu) *sink = 1;            // (keeps the LDS allocation alive)
}
// After the persistent kernels of a pass: if one of them aborted (non-zero status word of the pass), its outputs are
// garbage — overwrite them with NaN so that whatever consumes them (a loss, a vocoder, a file) cannot take them for data.
__global__ __launch_bounds__(256) void poison_if_aborted_kernel(const unsigned* __restrict__ status, int nwords, float* __restrict__ a, size_t na,
                                                                float* __restrict__ b, size_t nb) {
    unsigned any = 0;
    for (int i = 0; i < nwords; ++i) any |= status[i];
    if (!any) return;
    const float qnan = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na; i += (size_t)gridDim.x * 256) a[i] = qnan;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (size_t)gridDim.x * 256) b[i] = qnan;
}

namespace {

struct Sizes {
    int B, T, Tin, Tsub, M, P, E, Ha, Hd, A, WD, WO, NS;     // NS: attention streams (2 = BERT_Tacotron2, 1 = classic Tacotron2)
};

Sizes sizes_of(const t2_dims& d, int B, int T, int Tin, int Tsub) {
    Sizes z{};
    z.B = B; z.T = T; z.Tin = Tin; z.Tsub = Tsub;
    z.M = d.n_mel; z.P = d.prenet_dim; z.E = d.enc_dim; z.Ha = d.att_rnn_dim; z.Hd = d.dec_rnn_dim; z.A = d.att_dim;
    z.NS = d.n_streams == 1 ? 1 : 2;
    z.WD = z.NS * (z.Ha + z.E);
    z.WO = z.Hd + z.NS * z.E;
    return z;
}

int check_dims(const t2_dims& d) {
    T2_REQUIRE(d.n_mel % 4 == 0, "n_mel %d must be a multiple of 4", d.n_mel);
    T2_REQUIRE(d.prenet_dim % 64 == 0 && d.enc_dim % 64 == 0 && d.att_rnn_dim % 64 == 0 && d.dec_rnn_dim % 64 == 0,
               "prenet/encoder/rnn dims must be multiples of 64 (got %d %d %d %d)", d.prenet_dim, d.enc_dim, d.att_rnn_dim, d.dec_rnn_dim);
    T2_REQUIRE(d.att_dim % 4 == 0 && d.att_dim <= 256, "attention_dim %d unsupported", d.att_dim);
    T2_REQUIRE(d.attention_kind == T2_ATTN_SMA || d.attention_kind == T2_ATTN_LSA || d.attention_kind == T2_ATTN_FWD2 || d.attention_kind == T2_ATTN_GMM || d.attention_kind == T2_ATTN_DCA, "unknown attention kind %d", d.attention_kind);
    return 0;
}

// ForwardAttentionV2 as the reference runs it (attention.py:87-151; the caller never updates log_alpha, model.py:266-270,
// 355): log_alpha stays [0, -1e4, -1e4, ...], so the "forward" bias logsumexp(log_alpha_j, log_alpha_{j-1}) is exactly 0
// for j < 2 and about -1e4 beyond, and softmax(bias + energy) is the LSA softmax over the first two positions with
// exact zeros elsewhere.  It therefore runs on the LSA kernels with the valid length clamped to 2.
t2_dims canon_dims(const t2_dims& in, int* max_pos) {
    t2_dims d = in;
    *max_pos = 0;
    if (d.attention_kind == T2_ATTN_FWD2) { d.attention_kind = T2_ATTN_LSA; *max_pos = 2; }
    return d;
}

size_t align4(size_t n) { return (n + 3) & ~(size_t)3; }

// score_mask_value of the stream's attention module; a zero-initialised t2_dims (0.0) means the default, -inf
float mask_value_of(const t2_dims& d, int stream) {
    const float v = stream ? d.score_mask_value_sub : d.score_mask_value;
    return (v == 0.f && !d.score_mask_given) ? -INFINITY : v;
}

// Decode loop, bf16-operand mode: whole-cell weight shadows [W_hh | W_ih[:,P:] | W_ih[:,:P]] (attention LSTMs) and
// [W_ih | W_hh] (decoder LSTM) so that each cell is ONE K-contiguous product, plus the bf16 input rows the producing
// kernels write: att rows [2][NS][B][Ha+E+P] = [h | ctx | prenet], dec rows [2][B][WD+Hd] = [att_h | ctx | ... | dec_h],
// ping-pong on step parity (step t reads buffer t&1 and writes the recurrent parts into (t+1)&1).
struct InferShadows {
    int Ka, Kd; size_t wa[2], wd, rows_a, rows_d, total_floats;
};
InferShadows infer_shadows(const Sizes& z, size_t base) {
    InferShadows m{};
    m.Ka = z.Ha + z.E + z.P; m.Kd = z.WD + z.Hd;
    size_t off = base;
    auto take = [&](size_t elems) { size_t o = off; off += align4((elems + 1) / 2); return o; };
    m.wa[0] = take((size_t)4 * z.Ha * m.Ka); m.wa[1] = take((size_t)4 * z.Ha * m.Ka);
    m.wd = take((size_t)4 * z.Hd * m.Kd);
    m.rows_a = take((size_t)2 * 2 * z.B * m.Ka); m.rows_d = take((size_t)2 * z.B * m.Kd);
    m.total_floats = off - base;
    return m;
}

// Persistent chains: [status word | counters A | counters B | X of chain A | X of chain B | Q], each part 256-byte aligned.
// Sized for the largest tiling chain_plan can choose at this batch size.
struct ChainBufs { unsigned* err; unsigned* cnt_a; unsigned* cnt_b; unsigned char* xa; unsigned char* xb; unsigned char* xm; float* q; size_t xa_bytes, xb_bytes, q_bytes; };
constexpr size_t kChainXmBytes = 16 * 1024;     // decode loop: mel fragments [2][8][1 KB]
size_t chain_part_bytes(const Sizes& z, size_t* xa, size_t* xb, size_t* q) {
    const size_t MT = (z.B + 31) / 32;
    *xa = (size_t)2 * z.NS * ((z.Ha + z.E + z.P) / 16) * MT * 1024;      // (+ the prenet segment of the decode loop)
    *xb = (size_t)2 * (z.Hd / 16) * MT * 1024;
    *q = (size_t)z.NS * MT * 32 * (z.Ha / 8) * z.A * sizeof(float);
    return 256 + 2 * kChainCntBytes + *xa + *xb + kChainXmBytes + *q;
}
size_t chain_region_bytes(const Sizes& z) { size_t a, b, q; return (chain_part_bytes(z, &a, &b, &q) + 255) & ~(size_t)255; }
ChainBufs chain_bufs(const Sizes& z, const t2_decoder_layout& L, float* ws) {
    ChainBufs b{};
    chain_part_bytes(z, &b.xa_bytes, &b.xb_bytes, &b.q_bytes);
    unsigned char* p = reinterpret_cast<unsigned char*>(ws + L.chain);
    b.err = reinterpret_cast<unsigned*>(p); p += 256;
    b.cnt_a = reinterpret_cast<unsigned*>(p); p += kChainCntBytes;
    b.cnt_b = reinterpret_cast<unsigned*>(p); p += kChainCntBytes;
    b.xa = p; p += b.xa_bytes;
    b.xb = p; p += b.xb_bytes;
    b.xm = p; p += kChainXmBytes;
    b.q = reinterpret_cast<float*>(p);
    return b;
}

void layout_of(const t2_dims& d, const Sizes& z, t2_decoder_layout* L) {
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align4(n); return o; };
    const size_t BT = (size_t)z.B * z.T;
    L->x = take(BT * z.M);
    L->p1 = take(BT * z.P); L->p2 = take(BT * z.P); L->p1s = take(BT * z.P); L->p2s = take(BT * z.P);
    L->pm = take((size_t)z.B * z.Tin * z.A); L->pms = take((size_t)z.B * z.Tsub * z.A);
    L->prea = take(BT * 4 * z.Ha); L->preas = take(BT * 4 * z.Ha);
    L->ga = take(BT * 4 * z.Ha); L->gas = take(BT * 4 * z.Ha);
    L->cna = take(BT * z.Ha); L->cnas = take(BT * z.Ha); L->ca = take(BT * z.Ha); L->cas = take(BT * z.Ha);
    L->din = take(BT * z.WD);
    L->psel = take(BT * z.Tin); L->psels = take(BT * z.Tsub);
    L->wcum = take(BT * std::max(z.Tin, kGmmPad)); L->wcums = take(BT * std::max(z.Tsub, kGmmPad));   // LSA cumulative weights / GMM means
    L->pred = take(BT * 4 * z.Hd); L->gd = take(BT * 4 * z.Hd);
    L->cnd = take(BT * z.Hd); L->cd = take(BT * z.Hd);
    L->dout = take(BT * z.WO);
    L->qs = take(BT * z.A); L->qss = take(BT * z.A);
    L->qpart = take((size_t)2 * (z.Ha / 8) * z.B * z.A);
    L->w1t = take((size_t)2 * z.M * z.P);
    // bf16 shadow arena: teacher-forced passes keep [W_hh | W_ih[:,P:]] (+ transposes) per attention stream and W_hh of
    // the decoder LSTM; the decode loop keeps whole-cell shadows and ping-pong input rows (InferShadows) in the same space
    const size_t na = (size_t)4 * z.Ha * (z.Ha + z.E) / 2, nd = (size_t)4 * z.Hd * z.Hd / 2;      // bf16 pairs per float
    const size_t train16 = 4 * na + 2 * nd, infer16 = infer_shadows(z, 0).total_floats;
    const size_t arena = take(train16 > infer16 ? train16 : infer16);
    L->w16a = arena; L->w16as = arena + na; L->w16d = arena + 2 * na;
    L->wt16a = L->w16d + nd; L->wt16as = L->wt16a + na; L->wt16d = L->wt16as + na;
    L->din16 = take(BT * z.WD / 2 + 4); L->dh16 = take(BT * z.Hd / 2 + 4);
    L->gemm_ws_floats = (size_t)16 << 20;                     // 64 MiB of split-K scratch
    L->gemm_ws = take(L->gemm_ws_floats);
    // exchange buffers of the persistent chain kernels (chain.hip), see chain_bufs()
    L->chain_floats = chain_region_bytes(z) / sizeof(float);
    L->chain = take(L->chain_floats);
    // LSA: tanh tile and location features of every step, written by the forward chain for the backward chain (which then
    // repeats neither the location conv nor the tile: 1.3 GB + 0.33 GB per stream at B = 64, T = 400 — HBM is what this part has)
    const bool lsa = d.attention_kind == T2_ATTN_LSA;
    L->usave = take(lsa ? BT * align4(z.Tin) * z.A : 0); L->usaves = take(lsa && z.NS > 1 ? BT * align4(z.Tsub) * z.A : 0);   // [T][B][A][Tin rounded up to 4]
    L->locsave = take(lsa ? BT * z.Tin * d.loc_filters : 0); L->locsaves = take(lsa && z.NS > 1 ? BT * z.Tsub * d.loc_filters : 0);
    L->total_floats = off;
}

struct Dec {
    const t2_dims& d; const t2_decoder_weights& w; Sizes z; t2_decoder_layout L; float* ws;
    const float* memory; const float* memory_sub; const int32_t* len; const int32_t* len_sub;
    float* mel_out; float* gate_out; float* align; float* align_sub;
    bool training; bool prenet_dropout; bool teacher; uint64_t seed; hipStream_t s;
    bool use16 = false;                              // bf16-operand recurrent steps (t2_set_precision(1))
    hipStream_t sd = nullptr;                        // stream of the decoder-LSTM chain (== s unless overlapped)
    int max_pos = 0;                                 // > 0: attention restricted to the first max_pos positions (ForwardAttentionV2)
    InferShadows I{};                                // decode loop only (teacher == false && use16)
    __bf16* RowA(int parity, int s) const { return reinterpret_cast<__bf16*>(ws + I.rows_a) + (size_t)(parity * 2 + s) * z.B * I.Ka; }
    __bf16* RowD(int parity) const { return reinterpret_cast<__bf16*>(ws + I.rows_d) + (size_t)parity * z.B * I.Kd; }
    float* P(size_t off) const { return ws + off; }
    __bf16* P16(size_t off) const { return reinterpret_cast<__bf16*>(ws + off); }
    long R(int t) const { return (long)t * z.B; }      // first row of step t in a time-major [T,B,*] buffer
};

GemmDesc linear(const float* X, long ldx, const float* W, long ldw, float* Y, long ldy, int M, int N, int K) {
    GemmDesc g = gemm_desc();
    g.A = X; g.sam = ldx; g.sak = 1;
    g.B = W; g.sbn = ldw; g.sbk = 1;
    g.C = Y; g.ldc = ldy; g.M = M; g.N = N; g.K = K;
    return g;
}

// prenet for rows [row0, row0+rows) of the [B,T] grid.  teacher: all B*T rows at once.
int prenet(const Dec& c, bool sub, const float* X, long ldx, int M, float* P1, float* P2, long ldp, uint32_t base, uint32_t mstride) {
    const Sizes& z = c.z;
    GemmDesc g = linear(X, ldx, sub ? c.w.prenet_sub_w1 : c.w.prenet_w1, z.M, P1, ldp, M, z.P, z.M);
    g.act = ACT_RELU;
    if (c.prenet_dropout) {
        g.drop_p = c.d.p_prenet_dropout; g.seed = c.seed; g.site = sub ? T2_SITE_PRENET1_SUB : T2_SITE_PRENET1;
        g.drop_base = base; g.drop_mstride = mstride;
    }
    T2_TRY(gemm(g, c.s));
    GemmDesc h = linear(P1, ldp, sub ? c.w.prenet_sub_w2 : c.w.prenet_w2, z.P, P2, ldp, M, z.P, z.P);
    h.act = ACT_RELU;
    if (c.prenet_dropout) {
        h.drop_p = c.d.p_prenet_dropout; h.seed = c.seed; h.site = sub ? T2_SITE_PRENET2_SUB : T2_SITE_PRENET2;
        h.drop_base = base; h.drop_mstride = mstride;
    }
    return gemm(h, c.s);
}

// bf16-operand recurrent steps: precision mode 1, B <= 128, recurrent widths multiples of 256
bool use_bf16_steps(const t2_dims& d, const Sizes& z) {
    return get_precision() == 1 && z.B <= 128 && (z.Ha + z.E) % 256 == 0 && z.Hd % 256 == 0 && (4 * z.Ha) % 2048 == 0 && (4 * z.Hd) % 2048 == 0;
}
// weight shadows for one pass: [W_hh | W_ih[:,P:]] (K-contiguous, forward) and its transpose laid out
// [ctx columns | h columns] x 4H (backward), per attention stream; W_hh and W_hh^T of the decoder LSTM
int cast_shadows(const t2_dims& d, const t2_decoder_weights& w, const Sizes& z, const t2_decoder_layout& L, float* ws, hipStream_t s) {
    auto P16 = [&](size_t off) { return reinterpret_cast<__bf16*>(ws + off); };
    const long K = z.Ha + z.E, ldi = z.P + z.E;
    for (int st = 0; st < z.NS; ++st) {
        const t2_lstm_weights& lw = st ? w.att_sub : w.att;
        __bf16* f = P16(st ? L.w16as : L.w16a);
        T2_TRY(cast_rows_bf16(lw.w_hh, z.Ha, f, K, 4 * z.Ha, z.Ha, s));
        T2_TRY(cast_rows_bf16(lw.w_ih + z.P, ldi, f + z.Ha, K, 4 * z.Ha, z.E, s));
        __bf16* tr = P16(st ? L.wt16as : L.wt16a);
        T2_TRY(cast_transpose_bf16(lw.w_ih + z.P, ldi, tr, 4 * z.Ha, 4 * z.Ha, z.E, s));
        T2_TRY(cast_transpose_bf16(lw.w_hh, z.Ha, tr + (long)z.E * 4 * z.Ha, 4 * z.Ha, 4 * z.Ha, z.Ha, s));
    }
    T2_TRY(cast_rows_bf16(w.dec.w_hh, z.Hd, P16(L.w16d), z.Hd, 4 * z.Hd, z.Hd, s));
    T2_TRY(cast_transpose_bf16(w.dec.w_hh, z.Hd, P16(L.wt16d), 4 * z.Hd, 4 * z.Hd, z.Hd, s));
    return 0;
}

DcaWeights dca_weights(const t2_attention_weights& aw) {
    DcaWeights w{};
    w.bW = aw.mlp_b1; w.V = aw.mlp_w2; w.F = aw.loc_conv; w.U = aw.loc_dense; w.T = aw.dca_T; w.bT = aw.dca_bT; w.v = aw.v; w.P = aw.dca_P;
    return w;
}

int att_lstm_step(const Dec& c, int t) {
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L;
    LstmStepDesc d{};
    d.nstreams = z.NS; d.B = z.B; d.H = z.Ha; d.seed = c.seed;
    d.drop_p = c.training ? c.d.p_att_dropout : 0.f;
    float* DIN = c.P(L.din);
    for (int s = 0; s < z.NS; ++s) {
        LstmStream& st = d.st[s];
        const t2_lstm_weights& lw = s ? c.w.att_sub : c.w.att;
        const int hoff = s ? z.Ha + z.E : 0, coff = hoff + z.Ha;
        int n = 0;
        if (!c.teacher) {
            st.seg[n++] = LstmSeg{c.P(s ? L.p2s : L.p2) + c.R(t) * z.P, (long)z.P, lw.w_ih, (long)(z.P + z.E), z.P};
            st.bias1 = lw.b_ih; st.bias2 = lw.b_hh;
        } else {
            st.pre = c.P(s ? L.preas : L.prea) + c.R(t) * 4 * z.Ha; st.ldpre = 4 * z.Ha;
        }
        if (t > 0) {
            st.seg[n++] = LstmSeg{DIN + c.R(t - 1) * z.WD + coff, (long)z.WD, lw.w_ih + z.P, (long)(z.P + z.E), z.E};
            st.seg[n++] = LstmSeg{DIN + c.R(t - 1) * z.WD + hoff, (long)z.WD, lw.w_hh, (long)z.Ha, z.Ha};
            st.c_prev = c.P(s ? L.cas : L.ca) + c.R(t - 1) * z.Ha; st.ldc_prev = z.Ha;
        }
        st.nseg = n;
        st.gates = c.P(s ? L.gas : L.ga) + c.R(t) * 4 * z.Ha; st.ldgates = 4 * z.Ha;
        st.c_new = c.P(s ? L.cnas : L.cna) + c.R(t) * z.Ha; st.ldc_new = z.Ha;
        st.c_out = c.P(s ? L.cas : L.ca) + c.R(t) * z.Ha; st.ldc_out = z.Ha;
        st.h_out = DIN + c.R(t) * z.WD + hoff; st.ldh_out = z.WD;
        st.site_h = s ? T2_SITE_ATT_H_SUB : T2_SITE_ATT_H; st.site_c = s ? T2_SITE_ATT_C_SUB : T2_SITE_ATT_C;
        st.idx_base = (uint32_t)(c.R(t) * z.Ha); st.idx_bstride = (uint32_t)z.Ha;       // logical [T,B,Ha]
        st.wq = s ? c.w.attn_sub.wq : c.w.attn.wq; st.A = z.A;
        st.qpart = c.P(L.qpart) + (size_t)s * (z.Ha / 8) * z.B * z.A;
        if (c.use16 && !c.teacher) {                 // decode loop: [h | ctx | prenet] x [W_hh | W_ih[:,P:] | W_ih[:,:P]]
            st.nseg = 0;
            st.x16 = c.RowA(t & 1, s); st.ldx16 = c.I.Ka; st.w16 = c.P16(c.I.wa[s]); st.ldw16 = c.I.Ka; st.k16 = c.I.Ka;
            st.h16_out = c.RowA((t + 1) & 1, s); st.ldh16 = c.I.Ka;
            st.h16_out2 = c.RowD(t & 1) + hoff; st.ldh16_2 = c.I.Kd;
        } else if (c.use16) {                        // one K-contiguous bf16 segment [h | ctx] x [W_hh | W_ih[:,P:]]
            __bf16* D16 = c.P16(L.din16);
            st.nseg = 0;
            st.x16 = D16 + (t > 0 ? c.R(t - 1) * z.WD + hoff : 0); st.ldx16 = z.WD;
            st.w16 = c.P16(s ? L.w16as : L.w16a); st.ldw16 = z.Ha + z.E; st.k16 = t > 0 ? z.Ha + z.E : 0;
            st.h16_out = D16 + c.R(t) * z.WD + hoff; st.ldh16 = z.WD;
        }
    }
    ProfScope ps(PK_LSTM_ATT_FWD, c.s);
    return lstm_step_fwd(d, c.s);
}

int attention_step(const Dec& c, int t) {
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L;
    AttnStepDesc d{};
    d.nstreams = z.NS; d.B = z.B; d.A = z.A; d.E = z.E; d.kind = c.d.attention_kind;
    d.F = c.d.loc_filters; d.Kc = c.d.loc_kernel; d.seed = c.seed; d.first = t == 0;
    d.noise_std = (c.training && d.kind == T2_ATTN_SMA) ? 2.0f : 0.f;     // attention.py:315,346-348
    d.max_pos = c.max_pos;
    for (int s = 0; s < z.NS; ++s) {
        AttnStream& st = d.st[s];
        const t2_attention_weights& aw = s ? c.w.attn_sub : c.w.attn;
        const int Tin = s ? z.Tsub : z.Tin;
        float* al = s ? c.align_sub : c.align;                       // [B,T,Tin]: the reference's output layout
        const long ldA = (long)z.T * Tin;
        st.Tin = Tin;
        st.qpart = c.P(L.qpart) + (size_t)s * (z.Ha / 8) * z.B * z.A; st.nparts = z.Ha / 8;
        st.q_out = c.P(s ? L.qss : L.qs) + c.R(t) * z.A; st.ldq_out = z.A;
        st.pm = c.P(s ? L.pms : L.pm); st.memory = s ? c.memory_sub : c.memory;
        st.lengths = s ? c.len_sub : c.len;
        st.a_prev = t > 0 ? al + (long)(t - 1) * Tin : nullptr; st.lda_prev = ldA;
        st.a_out = al + (long)t * Tin; st.lda_out = ldA;
        if (d.kind == T2_ATTN_DCA) {
            st.dca = dca_weights(aw);
        } else if (d.kind == T2_ATTN_GMM) {
            float* mu = c.P(s ? L.wcums : L.wcum);                    // [T,B,kGmmPad]
            st.mu_prev = t > 0 ? mu + c.R(t - 1) * kGmmPad : nullptr; st.mu_out = mu + c.R(t) * kGmmPad;
            st.gmm_b1 = aw.mlp_b1; st.gmm_w2 = aw.mlp_w2; st.gmm_b2 = aw.mlp_b2;
        } else if (d.kind == T2_ATTN_SMA) {
            st.p_out = c.P(s ? L.psels : L.psel) + (long)t * Tin; st.ldp_out = ldA;
        } else {
            float* wc = c.P(s ? L.wcums : L.wcum);
            st.wcum_prev = t > 0 ? wc + (long)(t - 1) * Tin : nullptr; st.ldwcum_prev = ldA;
            st.wcum_out = wc + (long)t * Tin; st.ldwcum_out = ldA;
        }
        st.ctx1 = c.P(L.din) + c.R(t) * z.WD + ((s ? z.Ha + z.E : 0) + z.Ha); st.ldctx1 = z.WD;
        st.ctx2 = c.P(L.dout) + c.R(t) * z.WO + z.Hd + (s ? z.E : 0); st.ldctx2 = z.WO;
        if (c.use16 && !c.teacher) {
            st.ctx16 = c.RowA((t + 1) & 1, s) + z.Ha; st.ldctx16 = c.I.Ka;
            st.ctx16b = c.RowD(t & 1) + (s ? z.Ha + z.E : 0) + z.Ha; st.ldctx16b = c.I.Kd;
        } else if (c.use16) { st.ctx16 = c.P16(L.din16) + c.R(t) * z.WD + ((s ? z.Ha + z.E : 0) + z.Ha); st.ldctx16 = z.WD; }
        st.v = aw.v; st.loc_conv = aw.loc_conv; st.loc_dense = aw.loc_dense;
        st.site_noise = s ? T2_SITE_NOISE_SUB : T2_SITE_NOISE;
        st.mask_value = mask_value_of(c.d, s);                              // attention.py:37,79 / train.py:77-78
        st.idx_base = (uint32_t)(c.R(t) * Tin); st.idx_bstride = (uint32_t)Tin;          // logical [T,B,Tin]
    }
    if (d.kind == T2_ATTN_GMM) d.kind = 2;                       // kernel-level kind (0 SMA, 1 LSA, 2 GMM, 3 DCA)
    else if (d.kind == T2_ATTN_DCA) d.kind = 3;
    ProfScope ps(PK_ATTN_FWD, c.s);
    return attention_step_fwd(d, c.s);
}

int dec_lstm_step(const Dec& c, int t) {
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L;
    LstmStepDesc d{};
    d.nstreams = 1; d.B = z.B; d.H = z.Hd; d.seed = c.seed;
    d.drop_p = c.training ? c.d.p_dec_dropout : 0.f;
    LstmStream& st = d.st[0];
    int n = 0;
    if (!c.teacher) {
        st.seg[n++] = LstmSeg{c.P(L.din) + c.R(t) * z.WD, (long)z.WD, c.w.dec.w_ih, (long)z.WD, z.WD};
        st.bias1 = c.w.dec.b_ih; st.bias2 = c.w.dec.b_hh;
    } else {
        st.pre = c.P(L.pred) + c.R(t) * 4 * z.Hd; st.ldpre = 4 * z.Hd;
    }
    if (t > 0) {
        st.seg[n++] = LstmSeg{c.P(L.dout) + c.R(t - 1) * z.WO, (long)z.WO, c.w.dec.w_hh, (long)z.Hd, z.Hd};
        st.c_prev = c.P(L.cd) + c.R(t - 1) * z.Hd; st.ldc_prev = z.Hd;
    }
    st.nseg = n;
    st.gates = c.P(L.gd) + c.R(t) * 4 * z.Hd; st.ldgates = 4 * z.Hd;
    st.c_new = c.P(L.cnd) + c.R(t) * z.Hd; st.ldc_new = z.Hd;
    st.c_out = c.P(L.cd) + c.R(t) * z.Hd; st.ldc_out = z.Hd;
    st.h_out = c.P(L.dout) + c.R(t) * z.WO; st.ldh_out = z.WO;
    st.site_h = T2_SITE_DEC_H; st.site_c = T2_SITE_DEC_C;
    st.idx_base = (uint32_t)(c.R(t) * z.Hd); st.idx_bstride = (uint32_t)z.Hd;             // logical [T,B,Hd]
    if (c.use16 && !c.teacher) {                     // decode loop: [att_h | ctx | att_h_sub | ctx_sub | dec_h] x [W_ih | W_hh]
        st.nseg = 0;
        st.x16 = c.RowD(t & 1); st.ldx16 = c.I.Kd; st.w16 = c.P16(c.I.wd); st.ldw16 = c.I.Kd; st.k16 = c.I.Kd;
        st.h16_out = c.RowD((t + 1) & 1) + z.WD; st.ldh16 = c.I.Kd;
    } else if (c.use16) {
        __bf16* H16 = c.P16(L.dh16);
        st.nseg = 0;
        st.x16 = H16 + (t > 0 ? c.R(t - 1) * z.Hd : 0); st.ldx16 = z.Hd;
        st.w16 = c.P16(L.w16d); st.ldw16 = z.Hd; st.k16 = t > 0 ? z.Hd : 0;
        st.h16_out = H16 + c.R(t) * z.Hd; st.ldh16 = z.Hd;
    }
    hipStream_t sd = c.sd ? c.sd : c.s;
    ProfScope ps(PK_LSTM_DEC_FWD, sd);
    return lstm_step_fwd(d, sd);
}

// Persistent-kernel descriptors of the two teacher-forced chains (chain.hip).  Returns false when the shape, mode or
// device is not covered: the caller then launches the per-step kernels.
bool chain_a_desc(const Dec& c, ChainDesc* out) {
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L;
    if (!c.use16 || !c.teacher) return false;
    if (c.d.attention_kind != T2_ATTN_SMA && c.d.attention_kind != T2_ATTN_LSA) return false;
    ChainDesc d{};
    d.NS = z.NS; d.B = z.B; d.T = z.T; d.H = z.Ha; d.E = z.E; d.A = z.A; d.WD = z.WD; d.WO = z.WO;
    d.din = c.P(L.din); d.din16 = c.P16(L.din16); d.dout = c.P(L.dout);
    d.kind = c.d.attention_kind == T2_ATTN_SMA ? CHAIN_SMA : CHAIN_LSA;
    d.F = c.d.loc_filters; d.Kc = c.d.loc_kernel; d.max_pos = c.max_pos;
    d.drop_p = c.training ? c.d.p_att_dropout : 0.f;
    d.noise_std = (c.training && d.kind == CHAIN_SMA) ? 2.0f : 0.f;
    d.seed = c.seed;
    for (int s = 0; s < z.NS; ++s) {
        ChainStream& st = d.st[s];
        const t2_attention_weights& aw = s ? c.w.attn_sub : c.w.attn;
        const int hoff = s ? z.Ha + z.E : 0;
        st.w16 = c.P16(s ? L.w16as : L.w16a); st.ldw16 = z.Ha + z.E;
        st.pre = c.P(s ? L.preas : L.prea); st.wq = aw.wq;
        st.gates = c.P(s ? L.gas : L.ga); st.c_new = c.P(s ? L.cnas : L.cna); st.c_out = c.P(s ? L.cas : L.ca);
        st.h_out = c.P(L.din) + hoff; st.ldh = z.WD; st.h16_out = c.P16(L.din16) + hoff; st.ldh16 = z.WD;
        st.coff = hoff + z.Ha; st.ctx2off = z.Hd + (s ? z.E : 0);
        st.pm = c.P(s ? L.pms : L.pm); st.memory = s ? c.memory_sub : c.memory; st.lengths = s ? c.len_sub : c.len;
        st.Tin = s ? z.Tsub : z.Tin;
        st.align = s ? c.align_sub : c.align; st.psel = c.P(s ? L.psels : L.psel); st.wcum = c.P(s ? L.wcums : L.wcum);
        st.qs = c.P(s ? L.qss : L.qs);
        st.v = aw.v; st.loc_conv = aw.loc_conv; st.loc_dense = aw.loc_dense;
        if (d.kind == CHAIN_LSA) { st.usave = c.P(s ? L.usaves : L.usave); st.locsave = c.P(s ? L.locsaves : L.locsave); }
        st.site_h = s ? T2_SITE_ATT_H_SUB : T2_SITE_ATT_H; st.site_c = s ? T2_SITE_ATT_C_SUB : T2_SITE_ATT_C;
        st.site_noise = s ? T2_SITE_NOISE_SUB : T2_SITE_NOISE;
        st.mask_value = mask_value_of(c.d, s);
    }
    if (!chain_plan(d)) return false;
    const ChainBufs b = chain_bufs(z, L, c.ws);
    size_t xb = 0, qb = 0;
    chain_exchange_bytes(d, &xb, &qb);
    if (xb > b.xa_bytes || qb > b.q_bytes) return false;
    d.X = b.xa; d.Q = b.q; d.cnt = b.cnt_a; d.err = b.err; d.q_bytes = (unsigned)qb;
    *out = d;
    return true;
}
bool chain_b_desc(const Dec& c, ChainDesc* out) {
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L;
    if (!c.use16 || !c.teacher) return false;
    ChainDesc d{};
    d.NS = 1; d.B = z.B; d.T = z.T; d.H = z.Hd; d.E = 0; d.A = 0; d.WD = z.WD; d.WO = z.WO;
    d.kind = CHAIN_LSTM;
    d.drop_p = c.training ? c.d.p_dec_dropout : 0.f;
    d.seed = c.seed;
    ChainStream& st = d.st[0];
    st.w16 = c.P16(L.w16d); st.ldw16 = z.Hd; st.pre = c.P(L.pred);
    st.gates = c.P(L.gd); st.c_new = c.P(L.cnd); st.c_out = c.P(L.cd);
    st.h_out = c.P(L.dout); st.ldh = z.WO; st.h16_out = c.P16(L.dh16); st.ldh16 = z.Hd;
    st.site_h = T2_SITE_DEC_H; st.site_c = T2_SITE_DEC_C; st.Tin = 4;
    if (!chain_plan(d)) return false;
    const ChainBufs b = chain_bufs(z, L, c.ws);
    size_t xb = 0, qb = 0;
    chain_exchange_bytes(d, &xb, &qb);
    if (xb > b.xb_bytes) return false;
    d.X = b.xb; d.Q = nullptr; d.cnt = b.cnt_b; d.err = b.err + 1; d.q_bytes = 0;
    *out = d;
    return true;
}

// Persistent decode loop (chain.hip, dec mode): every phase of Decoder.inference's step inside one launch per step range
bool chain_dec_desc(const Dec& c, const t2_decoder_weights& w, const t2_decoder_infer_args& a, ChainDesc* out) {
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L;
    if (!c.use16 || z.NS != 2) return false;
    if (c.d.attention_kind != T2_ATTN_SMA && c.d.attention_kind != T2_ATTN_LSA) return false;
    ChainDesc d{};
    d.dec = 1; d.P = z.P; d.M = z.M; d.Hd = z.Hd;
    d.NS = z.NS; d.B = z.B; d.T = z.T; d.H = z.Ha; d.E = z.E; d.A = z.A; d.WD = z.WD; d.WO = z.WO;
    d.kind = c.d.attention_kind == T2_ATTN_SMA ? CHAIN_SMA : CHAIN_LSA;
    d.F = c.d.loc_filters; d.Kc = c.d.loc_kernel; d.max_pos = c.max_pos;
    d.drop_p = 0.f; d.noise_std = 0.f; d.seed = c.seed;                 // inference runs in eval mode (inference.py:263)
    for (int s = 0; s < z.NS; ++s) {
        ChainStream& st = d.st[s];
        const t2_attention_weights& aw = s ? w.attn_sub : w.attn;
        const t2_lstm_weights& lw = s ? w.att_sub : w.att;
        st.w16 = c.P16(c.I.wa[s]); st.ldw16 = c.I.Ka; st.wq = aw.wq;
        st.pm = c.P(s ? L.pms : L.pm); st.memory = s ? c.memory_sub : c.memory; st.lengths = s ? c.len_sub : c.len;
        st.Tin = s ? z.Tsub : z.Tin;
        st.align = s ? c.align_sub : c.align; st.wcum = c.P(s ? L.wcums : L.wcum);
        st.v = aw.v; st.loc_conv = aw.loc_conv; st.loc_dense = aw.loc_dense;
        st.site_h = s ? T2_SITE_ATT_H_SUB : T2_SITE_ATT_H; st.site_c = s ? T2_SITE_ATT_C_SUB : T2_SITE_ATT_C;
        st.site_noise = s ? T2_SITE_NOISE_SUB : T2_SITE_NOISE;
        st.mask_value = mask_value_of(c.d, s);
        d.bias1[s] = lw.b_ih; d.bias2[s] = lw.b_hh;
        d.att_c[s] = c.P(s ? L.cas : L.ca);                              // (row 0 of the per-frame buffers: unused in decode)
        d.pw1[s] = s ? w.prenet_sub_w1 : w.prenet_w1; d.pw2[s] = s ? w.prenet_sub_w2 : w.prenet_w2;
        d.psite1[s] = s ? T2_SITE_PRENET1_SUB : T2_SITE_PRENET1; d.psite2[s] = s ? T2_SITE_PRENET2_SUB : T2_SITE_PRENET2;
    }
    d.wd16 = c.P16(c.I.wd); d.ldwd = c.I.Kd; d.dbias1 = w.dec.b_ih; d.dbias2 = w.dec.b_hh; d.dec_c = c.P(L.cd);
    d.proj_w = w.proj_w; d.proj_b = w.proj_b; d.gate_w = w.gate_w; d.gate_b = w.gate_b;
    d.mel_out = a.mel_out; d.ldmel = (long)z.T * z.M; d.gate_out = a.gate_out; d.ldgate = z.T;
    d.thr = a.gate_threshold; d.stop_index = a.stop_index; d.done = a.done_count;
    d.pdrop = c.prenet_dropout ? c.d.p_prenet_dropout : 0.f;
    if (!chain_plan(d)) return false;
    const ChainBufs b = chain_bufs(z, L, c.ws);
    size_t xb = 0, qb = 0;
    chain_exchange_bytes(d, &xb, &qb);
    if (xb > b.xa_bytes || qb > b.q_bytes || (size_t)2 * (z.Hd / 16) * 1024 > b.xb_bytes) return false;
    d.X = b.xa; d.Q = b.q; d.cnt = b.cnt_a; d.err = b.err; d.q_bytes = (unsigned)qb; d.XD = b.xb; d.XM = b.xm;
    *out = d;
    return true;
}

// mel / gate projection (model.py:382-388); permute_tb: rows come time-major, outputs are [B,T,*]
int projection(const Dec& c, const float* X, long ldx, int M, float* mel, long ldmel, float* gate, long ldgate, bool permute_tb) {
    const Sizes& z = c.z;
    GemmDesc g = linear(X, ldx, c.w.proj_w, z.WO, mel, ldmel, M, z.M, z.WO);
    g.bias1 = c.w.proj_b;
    if (permute_tb) { g.crow_mod = z.B; g.crow_mul = z.T; }          // row (t,b) of DOUT -> row (b,t) of the output
    T2_TRY(gemm(g, c.s));
    GemmDesc h = linear(X, ldx, c.w.gate_w, z.WO, gate, ldgate, M, 1, z.WO);
    h.bias1 = c.w.gate_b;
    if (permute_tb) { h.crow_mod = z.B; h.crow_mul = z.T; }
    return gemm(h, c.s);
}

int processed_memory(const Dec& c) {
    const Sizes& z = c.z;
    if (c.d.attention_kind == T2_ATTN_GMM || c.d.attention_kind == T2_ATTN_DCA) return 0;   // purely location-based: memory_layer is never used
    GemmDesc g = linear(c.memory, z.E, c.w.attn.wm, z.E, c.P(c.L.pm), z.A, z.B * z.Tin, z.A, z.E);
    T2_TRY(gemm(g, c.s));
    if (z.NS == 1) return 0;
    GemmDesc h = linear(c.memory_sub, z.E, c.w.attn_sub.wm, z.E, c.P(c.L.pms), z.A, z.B * z.Tsub, z.A, z.E);
    return gemm(h, c.s);
}

__global__ void init_stop_kernel(int32_t* stop_index, int32_t* done, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) stop_index[b] = -1;
    if (b == 0) *done = 0;
}


// ------------------------------------------------------------------------------- backward
// SMA attention backward: two workgroups per (item, stream) when one each would leave CUs idle
int attn_bwd_nsplit(const t2_dims& d, const Sizes& z) {
    return (d.attention_kind == T2_ATTN_SMA && z.B * z.NS <= 128 && std::min(z.Tin, z.NS == 2 ? z.Tsub : z.Tin) >= 16) ? 2 : 1;
}

void bwd_layout_of(const t2_dims& d, const Sizes& z, t2_decoder_bwd_layout* L) {
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align4(n); return o; };
    const size_t BT = (size_t)z.B * z.T;
    const int ksd = lstm_bwd_ksplit(4 * z.Hd), ksa = lstm_bwd_ksplit(4 * z.Ha);
    L->ddout = take(BT * z.WO); L->ddin = take(BT * z.WD);
    L->dgd = take(BT * 4 * z.Hd); L->dga = take(BT * 4 * z.Ha); L->dgas = take(BT * 4 * z.Ha);
    L->dctx = take(BT * z.E); L->dctxs = take(BT * z.E);
    L->dq = take(BT * 2 * z.A); L->dqs = take(BT * 2 * z.A);              // rows [2][A]: one partial per position split
    L->dv = take((size_t)2 * z.B * z.A); L->dvs = take((size_t)2 * z.B * z.A);
    L->dpm = take((size_t)z.B * z.Tin * z.A); L->dpms = take((size_t)z.B * z.Tsub * z.A);
    L->carry = take((size_t)2 * z.B * z.Tin); L->carrys = take((size_t)2 * z.B * z.Tsub);   // ping-pong by step parity
    const bool lsa = d.attention_kind == T2_ATTN_LSA;          // LSA: cumulative-weight carry + per-item location-layer gradients
    const bool gmm = d.attention_kind == T2_ATTN_GMM;          // GMM: mean carry [B,8] + per-item db2 [B,16] / dW2 [B,3K*A]
    const bool dca = d.attention_kind == T2_ATTN_DCA;          // DCA: all per-item accumulators in one block (dldense)
    // (LSA: two rows per item — the persistent backward chain keeps one accumulator per position split)
    const size_t ncv = lsa ? (size_t)2 * z.B * d.loc_filters * 2 * d.loc_kernel : gmm ? (size_t)z.B * 16 : 0;
    const size_t nds = lsa ? (size_t)2 * z.B * z.A * d.loc_filters : gmm ? (size_t)z.B * 3 * kGmmK * z.A : dca ? (size_t)z.B * dca_acc_floats(z.A) : 0;
    L->carryc = take(lsa ? (size_t)z.B * z.Tin : gmm ? (size_t)z.B * kGmmPad : 0); L->carrycs = take(lsa ? (size_t)z.B * z.Tsub : gmm ? (size_t)z.B * kGmmPad : 0);
    L->dlconv = take(ncv); L->dlconvs = take(ncv); L->dldense = take(nds); L->dldenses = take(nds);
    L->dcd = take((size_t)z.B * z.Hd); L->dca = take((size_t)z.B * z.Ha); L->dcas = take((size_t)z.B * z.Ha);
    L->partd = take((size_t)ksd * z.B * z.Hd);
    L->parta = take((size_t)2 * ksa * z.B * (z.E + z.Ha));
    L->dp2 = take(BT * z.P); L->dp2s = take(BT * z.P); L->dp1 = take(BT * z.P);
    L->dmel_t = take(BT * z.M); L->dgate_t = take(BT);
    L->dg16a = take((size_t)2 * z.B * 4 * z.Ha / 2 + 4); L->dg16d = take((size_t)z.B * 4 * z.Hd / 2 + 4);
    L->colsum_ws = take((size_t)64 * 4 * (z.Ha > z.Hd ? z.Ha : z.Hd));
    L->gemm_ws_floats = (size_t)192 << 20;                    // 768 MiB: split-K partials + bf16 operand staging (gemm.hip)
    L->gemm_ws = take(L->gemm_ws_floats);
    {   // persistent backward chains (chain_bwd.hip): [counters | dg fragments | K-split partials] of the decoder-LSTM chain,
        // then [counters | dg fragments | h partials | ctx partials | dq partials | boundary carries] of the attention chain
        ChainBwdDesc cd{}; cd.kind = CHAIN_LSTM; cd.B = z.B; cd.H = z.Hd;
        size_t xb = 0, pb = 0;
        chain_bwd_exchange_bytes(cd, &xb, &pb);
        ChainBwdDesc ca{}; ca.kind = lsa ? CHAIN_LSA : CHAIN_SMA; ca.B = z.B; ca.H = z.Ha; ca.E = z.E; ca.A = z.A; ca.NS = z.NS;
        ca.F = d.loc_filters; ca.Kc = d.loc_kernel;
        size_t x2, ph, pc, dq, cr;
        const size_t att = chain_bwd_att_exchange_bytes(ca, &x2, &ph, &pc, &dq, &cr);
        L->chain_floats = (2 * kChainBwdCntBytes + xb + pb + att + 255) / sizeof(float);
        L->chain = take(L->chain_floats);
    }
    L->total_floats = off;
}

struct Bwd {
    const t2_dims& d; const t2_decoder_weights& w; const t2_decoder_grads& g; const t2_decoder_bwd_args& a;
    Sizes z; t2_decoder_layout L; t2_decoder_bwd_layout BL; hipStream_t s;
    const float* W(size_t off) const { return a.ws + off; }
    float* S(size_t off) const { return a.bws + off; }
    long R(int t) const { return (long)t * z.B; }
    bool use16 = false;
    hipStream_t sd = nullptr;                        // stream of the decoder-LSTM chain (== s unless overlapped)
    const __bf16* W16(size_t off) const { return reinterpret_cast<const __bf16*>(a.ws + off); }
    __bf16* S16(size_t off) const { return reinterpret_cast<__bf16*>(a.bws + off); }
    float* gemm_ws() const { return a.bws + BL.gemm_ws; }
    size_t gemm_ws_bytes() const { return BL.gemm_ws_floats * sizeof(float); }
};

// C[M,N] = X[M,K] . W[K,N]   (W row-major with leading dimension ldw: "NN")
GemmDesc matmul_nn(const float* X, long ldx, const float* W, long ldw, float* Y, long ldy, int M, int N, int K) {
    GemmDesc g = gemm_desc();
    g.A = X; g.sam = ldx; g.sak = 1;
    g.B = W; g.sbk = ldw; g.sbn = 1;
    g.C = Y; g.ldc = ldy; g.M = M; g.N = N; g.K = K;
    return g;
}
// C[M,N] = G[K,M]^T . X[K,N]   (weight gradients: K = B*T rows)
GemmDesc matmul_tn(const Bwd& c, const float* G, long ldg, const float* X, long ldx, float* Y, long ldy, int M, int N, int K) {
    GemmDesc g = gemm_desc();
    g.A = G; g.sam = 1; g.sak = ldg;
    g.B = X; g.sbk = ldx; g.sbn = 1;
    g.C = Y; g.ldc = ldy; g.M = M; g.N = N; g.K = K;
    g.ws = c.gemm_ws(); g.ws_bytes = c.gemm_ws_bytes();
    return g;
}

int dec_bwd_step(const Bwd& c, int t) {
    const Sizes& z = c.z;
    const int ks = lstm_bwd_ksplit(4 * z.Hd);
    LstmBwdPointDesc p{};
    p.nstreams = 1; p.B = z.B; p.H = z.Hd; p.seed = c.a.seed; p.first = t == z.T - 1;
    p.drop_p = c.a.training ? c.d.p_dec_dropout : 0.f;
    LstmBwdStream& st = p.st[0];
    st.dh1 = c.S(c.BL.ddout) + c.R(t) * z.WO; st.lddh1 = z.WO;
    st.part = c.S(c.BL.partd); st.nparts = ks; st.part_stride = (long)z.B * z.Hd; st.ldpart = z.Hd; st.part_col = 0;
    st.gates = c.W(c.L.gd) + c.R(t) * 4 * z.Hd; st.ldgates = 4 * z.Hd;
    st.c_new = c.W(c.L.cnd) + c.R(t) * z.Hd; st.ldc_new = z.Hd;
    if (t > 0) { st.c_prev = c.W(c.L.cd) + c.R(t - 1) * z.Hd; st.ldc_prev = z.Hd; }
    st.dc_state = c.S(c.BL.dcd);
    st.dg = c.S(c.BL.dgd) + c.R(t) * 4 * z.Hd; st.lddg = 4 * z.Hd;
    st.site_h = T2_SITE_DEC_H; st.site_c = T2_SITE_DEC_C;
    st.idx_base = (uint32_t)(c.R(t) * z.Hd); st.idx_bstride = (uint32_t)z.Hd;
    if (c.use16) st.dg16 = c.S16(c.BL.dg16d);
    hipStream_t sd = c.sd ? c.sd : c.s;
    { ProfScope ps(PK_LSTM_DEC_BWD_PW, sd); T2_TRY(lstm_bwd_pointwise(p, sd)); }
    if (t == 0) return 0;
    LstmBwdGemmDesc g{};
    g.nstreams = 1; g.B = z.B; g.H4 = 4 * z.Hd; g.KS = ks; g.NC = z.Hd;
    g.st[0].dg = st.dg; g.st[0].lddg = st.lddg;
    g.st[0].seg[0] = LstmBwdSeg{c.w.dec.w_hh, (long)z.Hd, z.Hd}; g.st[0].nseg = 1;
    g.st[0].part = c.S(c.BL.partd);
    if (c.use16) { g.st[0].dg16 = c.S16(c.BL.dg16d); g.st[0].wt16 = c.W16(c.L.wt16d); }
    ProfScope ps(PK_LSTM_DEC_BWD_GEMM, sd);
    return lstm_bwd_gemm(g, sd);
}

// Persistent BPTT of the decoder LSTM (chain_bwd.hip); false = not covered, per-step launches instead
bool chain_b_bwd_desc(const Bwd& c, ChainBwdDesc* out) {
    const Sizes& z = c.z;
    if (!c.use16) return false;
    ChainBwdDesc d{};
    d.NS = 1; d.B = z.B; d.T = z.T; d.H = z.Hd; d.kind = CHAIN_LSTM;
    d.drop_p = c.a.training ? c.d.p_dec_dropout : 0.f; d.seed = c.a.seed;
    ChainBwdStream& st = d.st[0];
    st.wt16 = c.W16(c.L.wt16d); st.ldwt = 4 * z.Hd;
    st.dh1 = c.S(c.BL.ddout); st.lddh1 = z.WO;
    st.gates = c.W(c.L.gd); st.c_new = c.W(c.L.cnd); st.c_out = c.W(c.L.cd);
    st.dg = c.S(c.BL.dgd); st.dc_state = c.S(c.BL.dcd);
    st.dbias_part = c.S(c.BL.partd);                                 // (the launch path's K-split scratch: idle when the chain runs) [MT][4Hd]
    st.site_h = T2_SITE_DEC_H; st.site_c = T2_SITE_DEC_C;
    if (!chain_bwd_plan(d)) return false;
    size_t xb = 0, pb = 0;
    chain_bwd_exchange_bytes(d, &xb, &pb);
    if ((2 * kChainBwdCntBytes + xb + pb + 3) / 4 > c.BL.chain_floats) return false;
    unsigned char* p = reinterpret_cast<unsigned char*>(c.S(c.BL.chain));
    d.cnt = reinterpret_cast<unsigned*>(p); p += kChainBwdCntBytes;
    d.X = p; p += xb;
    d.PB = p; d.pb_bytes = (unsigned)pb;
    d.err = reinterpret_cast<unsigned*>(const_cast<float*>(c.a.ws) + c.L.chain) + 2;      // status word 2 of the forward block
    *out = d;
    return true;
}

// The LSA backward chain reads the tanh tile and the location features the forward CHAIN saved (layout.usave / locsave); a
// workspace filled by the per-step launch path does not hold them.  Which workspaces do is kept here, per base pointer
// (every forward pass notes its own; host-side only, no device round trip).
std::mutex g_saved_mu;
std::unordered_map<const void*, bool> g_saved_tiles;
void saved_tiles_note(const void* ws, bool saved) {
    std::lock_guard<std::mutex> lk(g_saved_mu);
    if (g_saved_tiles.size() > 4096) g_saved_tiles.clear();
    g_saved_tiles[ws] = saved;
}
bool saved_tiles_have(const void* ws) {
    std::lock_guard<std::mutex> lk(g_saved_mu);
    auto it = g_saved_tiles.find(ws);
    return it != g_saved_tiles.end() && it->second;
}

// Persistent BPTT of the attention chain (both attention LSTMs + SMA attention); false = not covered
bool chain_a_bwd_desc(const Bwd& c, ChainBwdDesc* out) {
    const Sizes& z = c.z;
    const bool lsa = c.d.attention_kind == T2_ATTN_LSA;
    if (!c.use16 || !(lsa || (c.d.attention_kind == T2_ATTN_SMA && attn_bwd_nsplit(c.d, z) == 2))) return false;
    if (lsa && !saved_tiles_have(c.a.ws)) return false;              // forward ran on the launch path: its workspace has no tanh tile
    ChainBwdDesc d{};
    d.NS = z.NS; d.B = z.B; d.T = z.T; d.H = z.Ha; d.E = z.E; d.A = z.A; d.kind = lsa ? CHAIN_LSA : CHAIN_SMA;
    d.F = c.d.loc_filters; d.Kc = c.d.loc_kernel;
    d.drop_p = c.a.training ? c.d.p_att_dropout : 0.f; d.seed = c.a.seed;
    for (int s = 0; s < z.NS; ++s) {
        ChainBwdStream& st = d.st[s];
        const t2_attention_weights& aw = s ? c.w.attn_sub : c.w.attn;
        const int hoff = s ? z.Ha + z.E : 0, coff = hoff + z.Ha;
        st.wt16 = c.W16(s ? c.L.wt16as : c.L.wt16a); st.ldwt = 4 * z.Ha;
        st.dh1 = c.S(c.BL.ddin) + hoff; st.lddh1 = z.WD;
        st.gates = c.W(s ? c.L.gas : c.L.ga); st.c_new = c.W(s ? c.L.cnas : c.L.cna); st.c_out = c.W(s ? c.L.cas : c.L.ca);
        st.dg = c.S(s ? c.BL.dgas : c.BL.dga); st.dc_state = c.S(s ? c.BL.dcas : c.BL.dca);
        st.dbias_part = c.S(c.BL.parta) + (size_t)s * 2 * 4 * z.Ha;     // [MT][4Ha] per stream (the launch path's K-split scratch)
        st.site_h = s ? T2_SITE_ATT_H_SUB : T2_SITE_ATT_H; st.site_c = s ? T2_SITE_ATT_C_SUB : T2_SITE_ATT_C;
        st.dctx_a = c.S(c.BL.ddout) + z.Hd + (s ? z.E : 0); st.lddctx_a = z.WO;
        st.dctx_b = c.S(c.BL.ddin) + coff; st.lddctx_b = z.WD;
        st.dalign = s ? c.a.d_align_sub : c.a.d_align;
        st.qs = c.W(s ? c.L.qss : c.L.qs); st.pm = c.W(s ? c.L.pms : c.L.pm); st.memory = s ? c.a.memory_sub : c.a.memory;
        st.Tin = s ? z.Tsub : z.Tin;
        st.psel = c.W(s ? c.L.psels : c.L.psel); st.align = s ? c.a.align_sub : c.a.align;
        st.v = aw.v; st.wq = aw.wq;
        st.dctx_out = c.S(s ? c.BL.dctxs : c.BL.dctx); st.dq_out = c.S(s ? c.BL.dqs : c.BL.dq);
        st.dv_acc = c.S(s ? c.BL.dvs : c.BL.dv); st.dpm_acc = c.S(s ? c.BL.dpms : c.BL.dpm);
        if (lsa) {
            st.wcum = c.W(s ? c.L.wcums : c.L.wcum); st.loc_conv = aw.loc_conv; st.loc_dense = aw.loc_dense;
            st.usave = c.W(s ? c.L.usaves : c.L.usave); st.locsave = c.W(s ? c.L.locsaves : c.L.locsave);
            st.dconv_acc = c.S(s ? c.BL.dlconvs : c.BL.dlconv); st.ddense_acc = c.S(s ? c.BL.dldenses : c.BL.dldense);
        }
    }
    if (!chain_bwd_plan(d)) return false;
    ChainBwdDesc cd{}; cd.kind = CHAIN_LSTM; cd.B = z.B; cd.H = z.Hd;
    size_t xb0 = 0, pb0 = 0;
    chain_bwd_exchange_bytes(cd, &xb0, &pb0);
    size_t xb, ph, pc, dq, cr;
    const size_t att = chain_bwd_att_exchange_bytes(d, &xb, &ph, &pc, &dq, &cr);
    if ((2 * kChainBwdCntBytes + xb0 + pb0 + att + 3) / 4 > c.BL.chain_floats) return false;
    unsigned char* p = reinterpret_cast<unsigned char*>(c.S(c.BL.chain)) + kChainBwdCntBytes + xb0 + pb0;
    d.cnt = reinterpret_cast<unsigned*>(p); p += kChainBwdCntBytes;
    d.X = p; p += xb;
    d.PB = p; p += ph; d.pb_bytes = (unsigned)ph;
    d.PBC = p; p += pc; d.pbc_bytes = (unsigned)(2 * (size_t)z.NS * 4 * z.B * z.E * sizeof(float));
    d.DQX = reinterpret_cast<float*>(p); p += dq;
    d.CARRYX = reinterpret_cast<float*>(p);
    if (lsa)                                                              // bf16 Wd^T copies behind the tagged part (filled by the caller)
        for (int s = 0; s < z.NS; ++s)
            d.st[s].wdt16 = reinterpret_cast<const __bf16*>(p + chain_bwd_lsa_tagged_bytes(d)) + (size_t)s * d.F * d.A;
    d.err = reinterpret_cast<unsigned*>(const_cast<float*>(c.a.ws) + c.L.chain) + 3;      // status word 3 of the forward block
    *out = d;
    return true;
}

int att_bwd_step(const Bwd& c, int t) {
    const Sizes& z = c.z;
    const int ks = lstm_bwd_ksplit(4 * z.Ha);
    const int NC = z.E + z.Ha;
    const bool first = t == z.T - 1;
    // 1. attention backward (needs dctx(t) incl. the recurrent partials of step t+1)
    AttnBwdDesc ab{};
    ab.nstreams = z.NS; ab.B = z.B; ab.A = z.A; ab.E = z.E; ab.first = first;
    ab.kind = c.d.attention_kind; ab.F = c.d.loc_filters; ab.Kc = c.d.loc_kernel;
    ab.nsplit = attn_bwd_nsplit(c.d, z);
    for (int s = 0; s < z.NS; ++s) {
        AttnBwdStream& st = ab.st[s];
        const int Tin = s ? z.Tsub : z.Tin;
        const int hoff = s ? z.Ha + z.E : 0, coff = hoff + z.Ha;
        st.Tin = Tin;
        st.dctx[0] = c.S(c.BL.ddout) + c.R(t) * z.WO + z.Hd + (s ? z.E : 0); st.lddctx[0] = z.WO;
        st.dctx[1] = c.S(c.BL.ddin) + c.R(t) * z.WD + coff; st.lddctx[1] = z.WD;
        st.part = c.S(c.BL.parta) + (size_t)s * ks * z.B * NC; st.nparts = ks; st.part_stride = (long)z.B * NC; st.ldpart = NC; st.part_col = 0;
        const float* dal = s ? c.a.d_align_sub : c.a.d_align;
        if (dal) { st.dalign = dal + (long)t * Tin; st.lddalign = (long)z.T * Tin; }
        st.q = c.W(s ? c.L.qss : c.L.qs) + c.R(t) * z.A; st.ldq = z.A;
        st.pm = c.W(s ? c.L.pms : c.L.pm); st.memory = s ? c.a.memory_sub : c.a.memory;
        const float* al = s ? c.a.align_sub : c.a.align;
        const long ldA = (long)z.T * Tin;
        if (t > 0) { st.a_prev = al + (long)(t - 1) * Tin; st.lda_prev = ldA; }
        const t2_attention_weights& aw = s ? c.w.attn_sub : c.w.attn;
        st.v = aw.v;
        if (ab.kind == T2_ATTN_DCA) {
            st.w = al + (long)t * Tin; st.ldw = ldA;
            st.dca = dca_weights(aw);
            st.dca_acc = c.S(s ? c.BL.dldenses : c.BL.dldense);
        } else if (ab.kind == T2_ATTN_GMM) {
            st.w = al + (long)t * Tin; st.ldw = ldA;
            st.gmm_w2 = aw.mlp_w2; st.gmm_b2 = aw.mlp_b2;
            st.mu = c.W(s ? c.L.wcums : c.L.wcum) + c.R(t) * kGmmPad; st.ldmu = kGmmPad;
            st.mu_carry = c.S(s ? c.BL.carrycs : c.BL.carryc);
            st.db2_acc = c.S(s ? c.BL.dlconvs : c.BL.dlconv);
            st.dw2_acc = c.S(s ? c.BL.dldenses : c.BL.dldense);
        } else if (ab.kind == T2_ATTN_SMA) {
            st.p = c.W(s ? c.L.psels : c.L.psel) + (long)t * Tin; st.ldp = ldA;
        } else {
            st.w = al + (long)t * Tin; st.ldw = ldA;
            if (t > 0) { st.wcum_prev = c.W(s ? c.L.wcums : c.L.wcum) + (long)(t - 1) * Tin; st.ldwcum_prev = ldA; }
            st.loc_conv = aw.loc_conv; st.loc_dense = aw.loc_dense;
            st.carry_cum = c.S(s ? c.BL.carrycs : c.BL.carryc);
            st.dconv_acc = c.S(s ? c.BL.dlconvs : c.BL.dlconv);
            st.ddense_acc = c.S(s ? c.BL.dldenses : c.BL.dldense);
        }
        float* cbuf = c.S(s ? c.BL.carrys : c.BL.carry);
        if (ab.kind == T2_ATTN_SMA) { st.carry = cbuf + (size_t)((t + 1) & 1) * z.B * Tin; st.carry_out = cbuf + (size_t)(t & 1) * z.B * Tin; }
        else st.carry = cbuf;
        st.dctx_out = c.S(s ? c.BL.dctxs : c.BL.dctx) + c.R(t) * z.E; st.lddctx_out = z.E;
        st.dq_out = c.S(s ? c.BL.dqs : c.BL.dq) + c.R(t) * 2 * z.A; st.lddq_out = 2 * z.A;
        st.dv_acc = c.S(s ? c.BL.dvs : c.BL.dv);
        st.dpm_acc = c.S(s ? c.BL.dpms : c.BL.dpm);
    }
    if (ab.kind == T2_ATTN_GMM) ab.kind = 2;                     // kernel-level kind
    else if (ab.kind == T2_ATTN_DCA) ab.kind = 3;
    { ProfScope ps(PK_ATTN_BWD, c.s); T2_TRY(attention_step_bwd(ab, c.s)); }
    // 2. LSTM pointwise backward
    LstmBwdPointDesc p{};
    p.nstreams = z.NS; p.B = z.B; p.H = z.Ha; p.seed = c.a.seed; p.first = first;
    p.drop_p = c.a.training ? c.d.p_att_dropout : 0.f;
    for (int s = 0; s < z.NS; ++s) {
        LstmBwdStream& st = p.st[s];
        const int hoff = s ? z.Ha + z.E : 0;
        st.dh1 = c.S(c.BL.ddin) + c.R(t) * z.WD + hoff; st.lddh1 = z.WD;
        st.part = c.S(c.BL.parta) + (size_t)s * ks * z.B * NC; st.nparts = ks; st.part_stride = (long)z.B * NC; st.ldpart = NC; st.part_col = z.E;
        st.dq = c.S(s ? c.BL.dqs : c.BL.dq) + c.R(t) * 2 * z.A; st.lddq = 2 * z.A; st.dq_parts = ab.nsplit;
        st.wq = s ? c.w.attn_sub.wq : c.w.attn.wq; st.A = z.A;
        st.gates = c.W(s ? c.L.gas : c.L.ga) + c.R(t) * 4 * z.Ha; st.ldgates = 4 * z.Ha;
        st.c_new = c.W(s ? c.L.cnas : c.L.cna) + c.R(t) * z.Ha; st.ldc_new = z.Ha;
        if (t > 0) { st.c_prev = c.W(s ? c.L.cas : c.L.ca) + c.R(t - 1) * z.Ha; st.ldc_prev = z.Ha; }
        st.dc_state = c.S(s ? c.BL.dcas : c.BL.dca);
        st.dg = c.S(s ? c.BL.dgas : c.BL.dga) + c.R(t) * 4 * z.Ha; st.lddg = 4 * z.Ha;
        st.site_h = s ? T2_SITE_ATT_H_SUB : T2_SITE_ATT_H; st.site_c = s ? T2_SITE_ATT_C_SUB : T2_SITE_ATT_C;
        st.idx_base = (uint32_t)(c.R(t) * z.Ha); st.idx_bstride = (uint32_t)z.Ha;
        if (c.use16) st.dg16 = c.S16(c.BL.dg16a) + (size_t)s * z.B * 4 * z.Ha;
    }
    { ProfScope ps(PK_LSTM_ATT_BWD_PW, c.s); T2_TRY(lstm_bwd_pointwise(p, c.s)); }
    if (t == 0) return 0;
    // 3. recurrent-input gradients of this step: dg(t) . [W_ih[:, P:] | W_hh]  ->  partials for step t-1
    LstmBwdGemmDesc g{};
    g.nstreams = z.NS; g.B = z.B; g.H4 = 4 * z.Ha; g.KS = ks; g.NC = NC;
    for (int s = 0; s < z.NS; ++s) {
        const t2_lstm_weights& lw = s ? c.w.att_sub : c.w.att;
        g.st[s].dg = p.st[s].dg; g.st[s].lddg = p.st[s].lddg;
        g.st[s].seg[0] = LstmBwdSeg{lw.w_ih + z.P, (long)(z.P + z.E), z.E};
        g.st[s].seg[1] = LstmBwdSeg{lw.w_hh, (long)z.Ha, z.Ha};
        g.st[s].nseg = 2;
        g.st[s].part = c.S(c.BL.parta) + (size_t)s * ks * z.B * NC;
        if (c.use16) { g.st[s].dg16 = c.S16(c.BL.dg16a) + (size_t)s * z.B * 4 * z.Ha; g.st[s].wt16 = c.W16(s ? c.L.wt16as : c.L.wt16a); }
    }
    ProfScope ps(PK_LSTM_ATT_BWD_GEMM, c.s);
    return lstm_bwd_gemm(g, c.s);
}

}  // namespace

// Step ranges handed from one chain to the other.  One range when the chains share a stream; otherwise about eight,
// with short ranges (16, 32 steps) at the END of time: that is where the forward pass's decoder-LSTM chain finishes
// after the attention chain, and where the backward pass's attention chain waits for the first decoder-LSTM range,
// so whatever the last range holds is exposed.  Even boundaries keep rows-per-range a multiple of 128 at B = 64.
static std::vector<int> chunk_bounds(int T, bool overlap) {
    std::vector<int> b{0};
    if (!overlap) { b.push_back(T); return b; }
    const int CH = std::max(16, (T + 7) / 8);
    std::vector<int> tail;
    int rem = T;
    for (int s = 16; s < CH && rem - s >= CH; s *= 2) { tail.push_back(s); rem -= s; }
    const int n = std::max(1, rem / CH);
    for (int i = 1; i < n; ++i) {
        const int e = (int)((long)rem * i / n) & ~1;
        if (e > b.back()) b.push_back(e);
    }
    b.push_back(rem);
    for (auto it = tail.rbegin(); it != tail.rend(); ++it) b.push_back(b.back() + *it);
    return b;
}

extern "C" {

const char* t2_last_error(void) { return g_err; }
int t2_version(void) { return T2_ABI_VERSION; }
int t2_chain_status(uint32_t* out_host) {
    T2_REQUIRE(out_host, "null argument");
    const unsigned* w = chain_sticky_words();
    T2_REQUIRE(w, "t2_chain_status: no status block");
    for (int i = 0; i < 4; ++i) out_host[i] = __atomic_load_n(w + i, __ATOMIC_RELAXED);
    return 0;
}
int t2_chain_status_clear(void) {
    unsigned* w = chain_sticky_words();
    T2_REQUIRE(w, "t2_chain_status_clear: no status block");
    for (int i = 0; i < 4; ++i) __atomic_store_n(w + i, 0u, __ATOMIC_RELAXED);
    return 0;
}
int t2_debug_report_abort(uint32_t code, void* stream) {
    unsigned* w = chain_sticky_words();
    T2_REQUIRE(w, "t2_debug_report_abort: no status block");
    hipLaunchKernelGGL(status_report_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, code);
    T2_LAUNCH_CHECK();
    return 0;
}
int t2_chain_claimed(void) { return chain_device_claim() ? 1 : 0; }
int t2_debug_occupy(int workgroups, int milliseconds, void* stream) {
    T2_REQUIRE(workgroups >= 1 && workgroups <= 256 && milliseconds >= 1 && milliseconds <= 5000, "t2_debug_occupy: 1..256 workgroups, 1..5000 ms");
    unsigned* w = chain_sticky_words();
    T2_REQUIRE(w, "t2_debug_occupy: no status block");
    const size_t smem = 96 * 1024;                                          // more than half a CU's LDS: one such workgroup per CU, no chain workgroup beside it
    T2_TRY_RC(t2_allow_dynamic_lds(reinterpret_cast<const void*>(occupy_kernel), smem));
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(256), smem, (hipStream_t)stream, (unsigned long long)milliseconds * 100000ull, w + 8);
    T2_LAUNCH_CHECK();
    return 0;
}
int t2_set_precision(int mode) {
    T2_REQUIRE(mode == 0 || mode == 1, "t2_set_precision: mode must be 0 (fp32) or 1 (bf16 operands)");
    set_precision(mode);
    return 0;
}
int t2_get_precision(void) { return get_precision(); }
int t2_set_overlap(int on) { g_overlap = on != 0; return 0; }
int t2_set_chain(int on) { g_chain = on != 0; return 0; }
int t2_get_chain(void) { return g_chain; }
int t2_set_chain_bwd(int on) { g_chain_bwd = on != 0; return 0; }
int t2_set_gemm_staging(int on) { set_gemm_staging(on); return 0; }
int t2_side_join(void* stream) {
    Side* side = nullptr;
    T2_TRY(side_get(&side));
    hipEvent_t e;
    T2_TRY(side_event(*side, 0, &e));          // slot 0 is free between calls: every call re-records its events before use
    T2_CHECK_HIP(hipEventRecord(e, side->s));
    T2_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, e, 0));
    return 0;
}

int t2_decoder_layout_query(const t2_dims* dims_in, int B, int T, int Tin, int Tsub, t2_decoder_layout* out) {
    T2_REQUIRE(dims_in && out, "null argument");
    T2_TRY(check_dims(*dims_in));
    int max_pos = 0;
    const t2_dims dd = canon_dims(*dims_in, &max_pos);
    const t2_dims* dims = &dd;
    (void)max_pos;
    T2_REQUIRE(B >= 1 && B <= 256 && T >= 1 && Tin >= 1 && Tsub >= 1, "bad shape B=%d T=%d Tin=%d Tsub=%d", B, T, Tin, Tsub);
    layout_of(*dims, sizes_of(*dims, B, T, Tin, Tsub), out);
    return 0;
}

int t2_decoder_forward(const t2_dims* dims_in, const t2_decoder_weights* w, const t2_decoder_fwd_args* a, void* stream) {
    T2_REQUIRE(dims_in && w && a, "null argument");
    T2_TRY(check_dims(*dims_in));
    int max_pos = 0;
    const t2_dims dd = canon_dims(*dims_in, &max_pos);
    const t2_dims* dims = &dd;
    (void)max_pos;
    T2_REQUIRE(a->B >= 1 && a->B <= 256 && a->T >= 1, "bad shape B=%d T=%d", a->B, a->T);
    T2_REQUIRE((long)a->B * a->T * 4 * dims->att_rnn_dim < (1l << 32), "B*T too large for 32-bit RNG indices");
    Dec c{*dims, *w, sizes_of(*dims, a->B, a->T, a->Tin, a->Tsub), {}, a->ws,
          a->memory, a->memory_sub, a->mem_lengths, a->sub_lengths,
          a->mel_out, a->gate_out, a->align, a->align_sub,
          a->training != 0, a->prenet_dropout != 0, true, a->seed, (hipStream_t)stream};
    c.max_pos = max_pos;
    layout_of(*dims, c.z, &c.L);
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L;
    const int BT = z.B * z.T;

    T2_REQUIRE(a->phase >= 0 && a->phase <= 2, "t2_decoder_forward: phase must be 0, 1 or 2");
    c.use16 = use_bf16_steps(*dims, z);
    // bf16 steps keep a bf16 shadow of every DIN row (din16): with one bf16 copy of W_ih at the head of the scratch the
    // decoder-LSTM input GEMMs below read both operands as bf16 and stage nothing
    const size_t w16_bytes = ((size_t)4 * z.Hd * z.WD * sizeof(__bf16) + 255) & ~(size_t)255;
    const bool pre16 = c.use16 && (4 * z.Hd) % 64 == 0 && z.WD % 64 == 0 && L.gemm_ws_floats * sizeof(float) > w16_bytes;
    __bf16* w16 = reinterpret_cast<__bf16*>(c.P(L.gemm_ws));
    if (a->phase != 2) {                                                // ---- everything that does not read the memories
        if (c.use16) T2_TRY(cast_shadows(*dims, *w, z, L, a->ws, c.s));
        // teacher inputs and both prenets over all frames (model.py:407-413)
        T2_TRY(teacher_inputs(a->mels, c.P(L.x), z.B, z.M, z.T, c.s));
        T2_TRY(prenet(c, false, c.P(L.x), z.M, BT, c.P(L.p1), c.P(L.p2), z.P, 0, 0));      // rows time-major: (t,b)
        if (z.NS == 2) T2_TRY(prenet(c, true, c.P(L.x), z.M, BT, c.P(L.p1s), c.P(L.p2s), z.P, 0, 0));
        // hoisted input half of both attention LSTMs:  P2 . W_ih[:, :P]^T + b_ih + b_hh
        for (int s = 0; s < z.NS; ++s) {
            const t2_lstm_weights& lw = s ? w->att_sub : w->att;
            GemmDesc g = linear(c.P(s ? L.p2s : L.p2), z.P, lw.w_ih, z.P + z.E, c.P(s ? L.preas : L.prea), 4 * z.Ha, BT, 4 * z.Ha, z.P);
            g.bias1 = lw.b_ih; g.bias2 = lw.b_hh;
            g.ws = c.P(L.gemm_ws); g.ws_bytes = L.gemm_ws_floats * sizeof(float);       // bf16 staging (gemm.hip)
            T2_TRY(gemm(g, c.s));
        }
        if (pre16) T2_TRY(stage_bf16(w->dec.w_ih, true, z.WD, w16, 4 * z.Hd, z.WD, c.s));
        if (a->phase == 1) return 0;
    }
    T2_TRY(processed_memory(c));                                    // model.py:258,261
    // Two serial chains, overlapped in chunks of steps:
    //   A (caller's stream): attention LSTMs + attention — the only truly recurrent chain through the contexts
    //   B (side stream):     hoisted input half of the decoder LSTM for the chunk A just finished
    //                        ([att_h|ctx|att_h_sub|ctx_sub] . W_ih^T + b), then the decoder-LSTM recurrence over it
    // Persistent chains (chain.hip): every step of a chain in ONE launch, weights resident on chip.  Two persistent
    // grids must never be in flight together (each needs the whole device to make progress), so with them the chains
    // run back to back on the caller's stream: A over all steps, one input GEMM, B over all steps.
    ChainDesc ca{}, cb{};
    const bool chain_a = g_chain && chain_a_desc(c, &ca), chain_b = g_chain && chain_b_desc(c, &cb);
    saved_tiles_note(a->ws, chain_a && ca.kind == CHAIN_LSA);
    {   // status words always (0 = OK / not used); counters and the zero state of step -1 when a chain runs
        const ChainBufs bufs = chain_bufs(z, L, a->ws);
        // (the query partials carry step tags: their buffer starts out cleared too)
        T2_CHECK_HIP(hipMemsetAsync(bufs.err, 0, (chain_a || chain_b) ? 256 + 2 * kChainCntBytes + bufs.xa_bytes + bufs.xb_bytes + kChainXmBytes + bufs.q_bytes : 256, c.s));
    }
    Side* side = nullptr;
    const bool overlap = g_overlap && z.T >= 32 && !chain_a && !chain_b;
    if (overlap) { T2_TRY(side_get(&side)); c.sd = side->s; }
    const std::vector<int> bounds = chunk_bounds(z.T, overlap);
    size_t ne = 0;
    for (size_t ci = 0; ci + 1 < bounds.size(); ++ci) {
        const int t0 = bounds[ci], t1 = bounds[ci + 1];
        if (chain_a) {
            ca.t0 = t0; ca.t1 = t1;
            ProfScope ps(PK_CHAIN_A_FWD, c.s);
            T2_TRY(chain_fwd(ca, c.s));
        } else {
            for (int t = t0; t < t1; ++t) {
                T2_TRY(att_lstm_step(c, t));
                T2_TRY(attention_step(c, t));
            }
        }
        hipStream_t sb = overlap ? side->s : c.s;
        if (overlap) T2_TRY(stream_edge(*side, ne++, c.s, sb));
        GemmDesc g = linear(c.P(L.din) + c.R(t0) * z.WD, z.WD, w->dec.w_ih, z.WD, c.P(L.pred) + c.R(t0) * 4 * z.Hd, 4 * z.Hd,
                            (t1 - t0) * z.B, 4 * z.Hd, z.WD);
        g.bias1 = w->dec.b_ih; g.bias2 = w->dec.b_hh;
        g.ws = c.P(L.gemm_ws); g.ws_bytes = L.gemm_ws_floats * sizeof(float);       // chain A launches no GEMM: the scratch is chain B's
        if (pre16) {
            g.A16 = c.P16(L.din16) + c.R(t0) * z.WD; g.lda16 = z.WD;
            g.B16 = w16; g.ldb16 = z.WD;
            g.ws = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(g.ws) + w16_bytes); g.ws_bytes -= w16_bytes;
        }
        T2_TRY(gemm(g, sb));
        if (chain_b) {
            cb.t0 = t0; cb.t1 = t1;
            ProfScope ps(PK_CHAIN_B_FWD, sb);
            T2_TRY(chain_fwd(cb, sb));
        } else {
            for (int t = t0; t < t1; ++t) T2_TRY(dec_lstm_step(c, t));
        }
    }
    if (overlap) T2_TRY(stream_edge(*side, ne++, side->s, c.s));         // join
    // projections over all frames
    T2_TRY(projection(c, c.P(L.dout), z.WO, BT, a->mel_out, z.M, a->gate_out, 1, true));
    if (chain_a || chain_b) {            // an aborted chain must not pass for data (its status words: 0 and 1 of the block)
        hipLaunchKernelGGL(poison_if_aborted_kernel, dim3(64), dim3(256), 0, c.s, chain_bufs(z, L, a->ws).err, 2, a->mel_out, (size_t)BT * z.M, a->gate_out, (size_t)BT);
        T2_LAUNCH_CHECK();
    }
    return 0;
}


int t2_decoder_bwd_layout_query(const t2_dims* dims_in, int B, int T, int Tin, int Tsub, t2_decoder_bwd_layout* out) {
    T2_REQUIRE(dims_in && out, "null argument");
    T2_TRY(check_dims(*dims_in));
    int max_pos = 0;
    const t2_dims dd = canon_dims(*dims_in, &max_pos);
    const t2_dims* dims = &dd;
    (void)max_pos;
    T2_REQUIRE(B >= 1 && B <= 256 && T >= 1 && Tin >= 1 && Tsub >= 1, "bad shape B=%d T=%d Tin=%d Tsub=%d", B, T, Tin, Tsub);
    bwd_layout_of(*dims, sizes_of(*dims, B, T, Tin, Tsub), out);
    return 0;
}

int t2_decoder_backward(const t2_dims* dims_in, const t2_decoder_weights* w, const t2_decoder_grads* g,
                        const t2_decoder_bwd_args* a, void* stream) {
    T2_REQUIRE(dims_in && w && g && a, "null argument");
    T2_TRY(check_dims(*dims_in));
    int max_pos = 0;
    const t2_dims dd = canon_dims(*dims_in, &max_pos);
    const t2_dims* dims = &dd;
    (void)max_pos;
    Bwd c{*dims, *w, *g, *a, sizes_of(*dims, a->B, a->T, a->Tin, a->Tsub), {}, {}, (hipStream_t)stream};
    layout_of(*dims, c.z, &c.L);
    bwd_layout_of(*dims, c.z, &c.BL);
    c.use16 = use_bf16_steps(*dims, c.z);          // must match the forward pass (the shadows live in its workspace)
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L; const t2_decoder_bwd_layout& BL = c.BL;
    const int BT = z.B * z.T;
    float* cws = c.S(BL.colsum_ws);

    // ---- incoming gradients arrive in the output layout [B,T,*]; everything inside is time-major
    float* dmel = c.S(BL.dmel_t); float* dgate = c.S(BL.dgate_t);
    T2_TRY(permute_rows(a->d_mel, dmel, z.B, z.T, z.M, c.s));
    T2_TRY(permute_rows(a->d_gate, dgate, z.B, z.T, 1, c.s));
    // ---- projections (model.py:382-388): dDOUT = d_mel . Wproj + d_gate . Wgate ; weight gradients
    {
        GemmDesc x = matmul_nn(dmel, z.M, w->proj_w, z.WO, c.S(BL.ddout), z.WO, BT, z.WO, z.M);
        T2_TRY(gemm(x, c.s));
        GemmDesc y = matmul_nn(dgate, 1, w->gate_w, z.WO, c.S(BL.ddout), z.WO, BT, z.WO, 1);
        y.beta = 1.f;
        T2_TRY(gemm(y, c.s));
        T2_TRY(gemm(matmul_tn(c, dmel, z.M, c.W(L.dout), z.WO, g->proj_w, z.WO, z.M, z.WO, BT), c.s));
        T2_TRY(gemm(matmul_tn(c, dgate, 1, c.W(L.dout), z.WO, g->gate_w, z.WO, 1, z.WO, BT), c.s));
        T2_TRY(colsum(dmel, z.M, BT, z.M, g->proj_b, nullptr, cws, c.s));
        T2_TRY(colsum(dgate, 1, BT, 1, g->gate_b, nullptr, cws, c.s));
    }
    // ---- two reverse-time chains, overlapped in chunks of steps (see Side above):
    //   B (side stream):     decoder-LSTM BPTT of a chunk, then dDIN rows of the chunk = dG . W_ih
    //   A (caller's stream): attention-LSTM + attention BPTT of the chunk B finished
    // Persistent chains (chain_bwd.hip): a persistent grid needs the whole device, and two of them must never be in flight
    // together, so with the attention chain persistent everything runs on the caller's stream, one step range per chain.
    ChainBwdDesc cab{};
    const bool chain_a = g_chain && g_chain_bwd && chain_a_bwd_desc(c, &cab);
    Side* side = nullptr;
    const bool overlap = g_overlap && z.T >= 32 && !chain_a;
    size_t ne = 0;
    if (overlap) {
        T2_TRY(side_get(&side)); c.sd = side->s;
        T2_TRY(stream_edge(*side, ne++, c.s, side->s));                 // fork: dDOUT is complete
    }
    const std::vector<int> bounds = chunk_bounds(z.T, overlap);
    const float* DGd = c.S(BL.dgd);
    // bf16 mode: W_ih^T ([WD][4Hd], K contiguous) is staged once at the head of the scratch for all dDIN chunks
    unsigned char* const ws8 = reinterpret_cast<unsigned char*>(c.gemm_ws());
    const size_t wt_bytes = ((size_t)4 * z.Hd * z.WD * sizeof(__bf16) + 255) & ~(size_t)255;
    const bool pre16 = get_precision() == 1 && (4 * z.Hd) % 64 == 0 && z.WD % 64 == 0 && c.gemm_ws_bytes() > 2 * wt_bytes;
    if (pre16) T2_TRY(stage_bf16(w->dec.w_ih, false, z.WD, reinterpret_cast<__bf16*>(ws8), z.WD, 4 * z.Hd, overlap ? side->s : c.s));
    ChainBwdDesc cbb{};
    // (next to per-step launches of the attention chain a persistent decoder-LSTM grid only takes CUs away from them:
    //  measured 24.8 -> 26.3 ms; it runs when the attention chain is persistent too)
    const bool chain_b = chain_a && chain_b_bwd_desc(c, &cbb);
    for (size_t ci = bounds.size() - 1; ci > 0; --ci) {
        const int t0 = bounds[ci - 1], t1 = bounds[ci];
        hipStream_t sb = overlap ? side->s : c.s;
        if (chain_b) {
            cbb.t0 = t0; cbb.t1 = t1;
            ProfScope ps(PK_CHAIN_B_BWD, sb);
            T2_TRY(chain_bwd(cbb, sb));
        } else {
            for (int t = t1 - 1; t >= t0; --t) T2_TRY(dec_bwd_step(c, t));
        }
        GemmDesc dd = matmul_nn(DGd + c.R(t0) * 4 * z.Hd, 4 * z.Hd, w->dec.w_ih, z.WD, c.S(BL.ddin) + c.R(t0) * z.WD, z.WD,
                                (t1 - t0) * z.B, z.WD, 4 * z.Hd);
        dd.ws = c.gemm_ws(); dd.ws_bytes = c.gemm_ws_bytes();             // between fork and join the scratch is chain B's
        if (pre16) {
            dd.B16 = reinterpret_cast<const __bf16*>(ws8); dd.ldb16 = 4 * z.Hd;
            dd.ws = reinterpret_cast<float*>(ws8 + wt_bytes); dd.ws_bytes -= wt_bytes;
        }
        T2_TRY(gemm(dd, sb));
        if (overlap) T2_TRY(stream_edge(*side, ne++, sb, c.s));
        if (t0 == 0) {
            // the decoder LSTM's weight gradients need nothing from chain A: they run on chain B's stream once its
            // recurrence is done, underneath the rest of chain A (whose launches leave most CUs idle)
            const float* DG = DGd;
            // bf16 mode: both products read ONE bf16 transpose of dG (the shifted one starts B columns in)
            const size_t dgt_bytes = ((size_t)4 * z.Hd * BT * sizeof(__bf16) + 255) & ~(size_t)255;
            const bool share = get_precision() == 1 && BT % 64 == 0 && z.B % 8 == 0 && (4 * z.Hd) % 128 == 0 && z.T > 1 &&
                               c.gemm_ws_bytes() >= dgt_bytes + ((size_t)z.WD * BT * sizeof(__bf16) + 256);
            __bf16* dgT = reinterpret_cast<__bf16*>(ws8);
            if (share) T2_TRY(stage_bf16(DG, false, 4 * z.Hd, dgT, 4 * z.Hd, BT, sb));
            auto with_dgT = [&](GemmDesc m, long col0) {
                if (share) {
                    m.A16 = dgT + col0; m.lda16 = BT;
                    m.ws = reinterpret_cast<float*>(ws8 + dgt_bytes); m.ws_bytes = c.gemm_ws_bytes() - dgt_bytes;
                }
                return m;
            };
            // dW_ih = dG^T . DIN ; recurrent half: dW_hh = dG^T . dec_h(t-1)
            T2_TRY(gemm(with_dgT(matmul_tn(c, DG, 4 * z.Hd, c.W(L.din), z.WD, g->dec.w_ih, z.WD, 4 * z.Hd, z.WD, BT), 0), sb));
            // h(t-1) pairs with dG(t): drop the first step's rows of dG and the last step's rows of dec_h
            if (z.T > 1) T2_TRY(gemm(with_dgT(matmul_tn(c, DG + (long)z.B * 4 * z.Hd, 4 * z.Hd, c.W(L.dout), z.WO, g->dec.w_hh, z.Hd, 4 * z.Hd, z.Hd, BT - z.B), z.B), sb));
            else T2_TRY(fill_f32(g->dec.w_hh, 0.f, (size_t)4 * z.Hd * z.Hd, sb));
            if (chain_b) {                                                // the chain summed dG over steps and rows: add the row tiles
                T2_TRY(batch_sum(cbb.st[0].dbias_part, (z.B + 31) / 32, 4 * z.Hd, g->dec.b_ih, sb));
                T2_CHECK_HIP(hipMemcpyAsync(g->dec.b_hh, g->dec.b_ih, (size_t)4 * z.Hd * sizeof(float), hipMemcpyDeviceToDevice, sb));
            } else T2_TRY(colsum(DG, 4 * z.Hd, BT, 4 * z.Hd, g->dec.b_ih, g->dec.b_hh, cws, sb));
        }
        if (chain_a) {
            cab.t0 = t0; cab.t1 = t1;
            if (cab.kind == CHAIN_LSA)
                for (int s = 0; s < z.NS; ++s)                            // [A][F] -> bf16 [F][A]
                    T2_TRY(cast_transpose_bf16(cab.st[s].loc_dense, cab.F, const_cast<__bf16*>(cab.st[s].wdt16), z.A, z.A, cab.F, c.s));
            ProfScope ps(PK_CHAIN_A_BWD, c.s);
            T2_TRY(chain_bwd(cab, c.s));
        } else {
            for (int t = t1 - 1; t >= t0; --t) T2_TRY(att_bwd_step(c, t));
        }
    }
    // ---- after the chains.  Only d(memory) feeds the caller's next backward nodes (the encoders); every weight gradient
    // is a leaf.  defer_weight_grads: the weight-gradient tail (and the decoder-LSTM weight gradients already queued
    // there) stays on the side stream, un-joined, underneath the encoders' backward — the caller joins with
    // t2_side_join() before it reads a gradient or releases a workspace.  Otherwise: join here, one stream.
    const bool defer = overlap && a->defer_weight_grads != 0;
    hipStream_t ts = c.s;                                                // stream of the weight-gradient tail
    if (defer) {
        T2_TRY(stream_edge(*side, ne++, c.s, side->s));                 // chain A is complete: dG, dq, d(pm) of every step
        ts = side->s;
    } else if (overlap) {
        T2_TRY(stream_edge(*side, ne++, side->s, c.s));                 // join (split-K scratch and colsum scratch are shared)
    }
    for (int s = 0; s < z.NS; ++s) {
        const t2_lstm_weights& lw = s ? w->att_sub : w->att;
        const t2_lstm_grads& lg = s ? g->att_sub : g->att;
        const t2_attention_weights& aw = s ? w->attn_sub : w->attn;
        const t2_attention_grads& ag = s ? g->attn_sub : g->attn;
        const int hoff = s ? z.Ha + z.E : 0, coff = hoff + z.Ha;
        const int Tin = s ? z.Tsub : z.Tin;
        const float* DG = c.S(s ? BL.dgas : BL.dga);
        const float* DIN = c.W(L.din);
        const float* P1 = c.W(s ? L.p1s : L.p1); const float* P2 = c.W(s ? L.p2s : L.p2);
        float* dP2 = c.S(s ? BL.dp2s : BL.dp2); float* dP1 = c.S(BL.dp1);
        const long ldw = z.P + z.E;
        // LSTM weights: W_ih = [prenet part | ctx part], W_hh, biases.  bf16 mode: the three products share ONE bf16
        // transpose of dG ([4Ha][BT] at the head of the scratch; the shifted products start B columns in)
        const size_t dgt_bytes = ((size_t)4 * z.Ha * BT * sizeof(__bf16) + 255) & ~(size_t)255;
        const bool share = get_precision() == 1 && BT % 64 == 0 && z.B % 8 == 0 && (4 * z.Ha) % 128 == 0 && z.T > 1 &&
                           c.gemm_ws_bytes() >= 2 * dgt_bytes;
        __bf16* dgT = reinterpret_cast<__bf16*>(c.gemm_ws());
        if (share) T2_TRY(stage_bf16(DG, false, 4 * z.Ha, dgT, 4 * z.Ha, BT, ts));
        auto dw_gemm = [&](const float* G, long col0, const float* X, long ldx, float* Y, long ldy, int N, int K) -> int {
            GemmDesc m = matmul_tn(c, G, 4 * z.Ha, X, ldx, Y, ldy, 4 * z.Ha, N, K);
            if (share) {
                m.A16 = dgT + col0; m.lda16 = BT;
                m.ws = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(c.gemm_ws()) + dgt_bytes); m.ws_bytes = c.gemm_ws_bytes() - dgt_bytes;
            }
            return gemm(m, ts);
        };
        T2_TRY(dw_gemm(DG, 0, P2, z.P, lg.w_ih, ldw, z.P, BT));
        if (z.T > 1) {
            const float* DG1 = DG + (long)z.B * 4 * z.Ha;             // rows of steps 1..T-1 pair with ctx/h of steps 0..T-2
            T2_TRY(dw_gemm(DG1, z.B, DIN + coff, z.WD, lg.w_ih + z.P, ldw, z.E, BT - z.B));
            T2_TRY(dw_gemm(DG1, z.B, DIN + hoff, z.WD, lg.w_hh, z.Ha, z.Ha, BT - z.B));
        } else {
            GemmDesc zc = matmul_tn(c, DG, 4 * z.Ha, DIN + coff, z.WD, lg.w_ih + z.P, ldw, 4 * z.Ha, z.E, BT);
            zc.alpha = 0.f;
            T2_TRY(gemm(zc, ts));
            T2_TRY(fill_f32(lg.w_hh, 0.f, (size_t)4 * z.Ha * z.Ha, ts));
        }
        if (chain_a) {
            T2_TRY(batch_sum(cab.st[s].dbias_part, (z.B + 31) / 32, 4 * z.Ha, lg.b_ih, ts));
            T2_CHECK_HIP(hipMemcpyAsync(lg.b_hh, lg.b_ih, (size_t)4 * z.Ha * sizeof(float), hipMemcpyDeviceToDevice, ts));
        } else T2_TRY(colsum(DG, 4 * z.Ha, BT, 4 * z.Ha, lg.b_ih, lg.b_hh, cws, ts));
        // prenet (model.py:13-24): dP2 = dG . W_ih[:, :P] ; through ReLU+dropout ; layer 2 ; layer 1
        const float scale = a->prenet_dropout ? 1.0f / (1.0f - dims->p_prenet_dropout) : 1.0f;
        GemmDesc gp = matmul_nn(DG, 4 * z.Ha, lw.w_ih, ldw, dP2, z.P, BT, z.P, 4 * z.Ha);
        gp.ws = c.gemm_ws(); gp.ws_bytes = c.gemm_ws_bytes();
        T2_TRY(gemm(gp, ts));
        T2_TRY(relu_drop_bwd(dP2, P2, dP2, scale, (size_t)BT * z.P, ts));
        T2_TRY(gemm(matmul_tn(c, dP2, z.P, P1, z.P, s ? g->prenet_sub_w2 : g->prenet_w2, z.P, z.P, z.P, BT), ts));
        T2_TRY(gemm(matmul_nn(dP2, z.P, s ? w->prenet_sub_w2 : w->prenet_w2, z.P, dP1, z.P, BT, z.P, z.P), ts));
        T2_TRY(relu_drop_bwd(dP1, P1, dP1, scale, (size_t)BT * z.P, ts));
        T2_TRY(gemm(matmul_tn(c, dP1, z.P, c.W(L.x), z.M, s ? g->prenet_sub_w1 : g->prenet_w1, z.M, z.P, z.M, BT), ts));
        // attention parameters
        const bool lsa_chain = chain_a && cab.kind == CHAIN_LSA;          // the persistent LSA backward: one partial per position split
        const int nsp = lsa_chain ? 2 : attn_bwd_nsplit(*dims, z);
        float* DQ = c.S(s ? BL.dqs : BL.dq);
        if (nsp == 2) T2_TRY(fold_halves(DQ, BT, z.A, ts));               // dq row = partial 0 + partial 1
        T2_TRY(gemm(matmul_tn(c, DQ, 2 * z.A, DIN + hoff, z.WD, ag.wq, z.Ha, z.A, z.Ha, BT), ts));
        const bool dcak = dims->attention_kind == T2_ATTN_DCA;
        const bool gmm = dims->attention_kind == T2_ATTN_GMM || dcak;     // both: no processed-memory term
        if (dcak) {
            // W.weight went through the query-projection path above; W.bias = column sums of dq; the rest from the per-item
            // accumulators dv | dbT | dU | dT | dF | dV (attention.hip)
            T2_REQUIRE(ag.mlp_b1 && ag.mlp_w2 && ag.loc_conv && ag.loc_dense && ag.dca_T && ag.dca_bT && ag.v, "t2_decoder_backward: DCA gradient buffers missing");
            T2_TRY(colsum(DQ, 2 * z.A, BT, z.A, ag.mlp_b1, nullptr, cws, ts));
            const size_t na = dca_acc_floats(z.A);
            float* full = c.gemm_ws();                                   // [na] floats of the split-K scratch (idle here)
            T2_REQUIRE(na * sizeof(float) <= c.gemm_ws_bytes(), "t2_decoder_backward: scratch too small");
            T2_TRY(batch_sum(c.S(s ? BL.dldenses : BL.dldense), z.B, (int)na, full, ts));
            auto cp = [&](float* dst, size_t off, size_t n) { return hipMemcpyAsync(dst, full + off, n * sizeof(float), hipMemcpyDeviceToDevice, ts); };
            size_t off = 0;
            T2_CHECK_HIP(cp(ag.v, off, z.A)); off += z.A;
            T2_CHECK_HIP(cp(ag.dca_bT, off, z.A)); off += z.A;
            T2_CHECK_HIP(cp(ag.loc_dense, off, (size_t)z.A * kDcaC)); off += (size_t)z.A * kDcaC;
            T2_CHECK_HIP(cp(ag.dca_T, off, (size_t)z.A * kDcaC)); off += (size_t)z.A * kDcaC;
            T2_CHECK_HIP(cp(ag.loc_conv, off, kDcaC * kDcaK)); off += kDcaC * kDcaK;
            T2_CHECK_HIP(cp(ag.mlp_w2, off, (size_t)kDcaC * kDcaK * z.A));
        } else if (gmm) {
            // mlp.0.weight was handled as the query projection above; mlp.0.bias = column sums of dq; second layer from
            // the per-item accumulators; memory_layer takes no part in the arithmetic (its gradient is None in the reference)
            T2_REQUIRE(ag.mlp_b1 && ag.mlp_w2 && ag.mlp_b2, "t2_decoder_backward: GMM needs mlp_b1 / mlp_w2 / mlp_b2 gradient buffers");
            T2_TRY(colsum(DQ, 2 * z.A, BT, z.A, ag.mlp_b1, nullptr, cws, ts));
            T2_TRY(batch_sum(c.S(s ? BL.dldenses : BL.dldense), z.B, 3 * kGmmK * z.A, ag.mlp_w2, ts));
            float* b2tmp = cws;                                          // 16 floats of scratch (slot 15 is padding)
            T2_TRY(batch_sum(c.S(s ? BL.dlconvs : BL.dlconv), z.B, 16, b2tmp, ts));
            T2_CHECK_HIP(hipMemcpyAsync(ag.mlp_b2, b2tmp, 3 * kGmmK * sizeof(float), hipMemcpyDeviceToDevice, ts));
        } else {
            T2_TRY(batch_sum(c.S(s ? BL.dvs : BL.dv), nsp * z.B, z.A, ag.v, ts));
        }
        if (dims->attention_kind == T2_ATTN_LSA) {
            T2_REQUIRE(ag.loc_conv && ag.loc_dense, "t2_decoder_backward: LSA needs loc_conv / loc_dense gradient buffers");
            T2_TRY(batch_sum(c.S(s ? BL.dlconvs : BL.dlconv), (lsa_chain ? 2 : 1) * z.B, dims->loc_filters * 2 * dims->loc_kernel, ag.loc_conv, ts));
            T2_TRY(batch_sum(c.S(s ? BL.dldenses : BL.dldense), (lsa_chain ? 2 : 1) * z.B, z.A * dims->loc_filters, ag.loc_dense, ts));
        }
        const float* mem = s ? a->memory_sub : a->memory;
        float* dmem = s ? a->d_memory_sub : a->d_memory;
        if (!gmm) {
            const float* DPM = c.S(s ? BL.dpms : BL.dpm);
            T2_TRY(gemm(matmul_tn(c, DPM, z.A, mem, z.E, ag.wm, z.E, z.A, z.E, z.B * Tin), ts));
            // d(memory) = dPM . Wm  +  per item: align^T [Tin x T] . dctx [T x E]
            T2_TRY(gemm(matmul_nn(DPM, z.A, aw.wm, z.E, dmem, z.E, z.B * Tin, z.E, z.A), c.s));     // caller's stream: feeds the encoders
        }
        GemmDesc dm = gemm_desc();
        dm.A = s ? a->align_sub : a->align; dm.sam = 1; dm.sak = Tin; dm.bsA = (long)z.T * Tin;
        dm.B = c.S(s ? BL.dctxs : BL.dctx); dm.sbk = (long)z.B * z.E; dm.sbn = 1; dm.bsB = z.E;     // dctx is [T,B,E]
        dm.C = dmem; dm.ldc = z.E; dm.bsC = (long)Tin * z.E;
        dm.M = Tin; dm.N = z.E; dm.K = z.T; dm.batch = z.B; dm.beta = gmm ? 0.f : 1.f;
        T2_TRY(gemm(dm, c.s));
    }
    return 0;
}

int t2_decoder_infer(const t2_dims* dims_in, const t2_decoder_weights* w, const t2_decoder_infer_args* a, void* stream) {
    T2_REQUIRE(dims_in && w && a && a->steps_run_host, "null argument");
    T2_TRY(check_dims(*dims_in));
    int max_pos = 0;
    const t2_dims dd = canon_dims(*dims_in, &max_pos);
    const t2_dims* dims = &dd;
    (void)max_pos;
    T2_REQUIRE(a->B >= 1 && a->B <= 256 && a->max_steps >= 1, "bad shape B=%d max_steps=%d", a->B, a->max_steps);
    Dec c{*dims, *w, sizes_of(*dims, a->B, a->max_steps, a->Tin, a->Tsub), {}, a->ws,
          a->memory, a->memory_sub, a->mem_lengths, a->sub_lengths,
          a->mel_out, a->gate_out, a->align, a->align_sub,
          false, a->prenet_dropout != 0, false, a->seed, (hipStream_t)stream};
    c.max_pos = max_pos;
    layout_of(*dims, c.z, &c.L);
    const Sizes& z = c.z; const t2_decoder_layout& L = c.L;
    const int T = z.T;
    int poll = a->poll_every > 0 ? a->poll_every : 16;

    hipLaunchKernelGGL(init_stop_kernel, dim3((z.B + 63) / 64), dim3(64), 0, c.s, a->stop_index, a->done_count, z.B);
    T2_LAUNCH_CHECK();
    T2_TRY(processed_memory(c));
    for (int s = 0; s < z.NS; ++s)                                   // W1 [P,M] -> [M,P]: coalesced thread-per-output reads
        T2_TRY(permute_rows(s ? w->prenet_sub_w1 : w->prenet_w1, c.P(L.w1t) + (size_t)s * z.M * z.P, z.P, z.M, 1, c.s));
    c.I = infer_shadows(z, L.w16a);
    c.use16 = get_precision() == 1 && z.B <= 128 && c.I.Ka % 256 == 0 && c.I.Kd % 256 == 0;
    if (c.use16) {
        const long ldi = z.P + z.E;
        for (int s = 0; s < z.NS; ++s) {
            const t2_lstm_weights& lw = s ? w->att_sub : w->att;
            __bf16* f = c.P16(c.I.wa[s]);
            T2_TRY(cast_rows_bf16(lw.w_hh, z.Ha, f, c.I.Ka, 4 * z.Ha, z.Ha, c.s));
            T2_TRY(cast_rows_bf16(lw.w_ih + z.P, ldi, f + z.Ha, c.I.Ka, 4 * z.Ha, z.E, c.s));
            T2_TRY(cast_rows_bf16(lw.w_ih, ldi, f + z.Ha + z.E, c.I.Ka, 4 * z.Ha, z.P, c.s));
        }
        T2_TRY(cast_rows_bf16(w->dec.w_ih, z.WD, c.P16(c.I.wd), c.I.Kd, 4 * z.Hd, z.WD, c.s));
        T2_TRY(cast_rows_bf16(w->dec.w_hh, z.Hd, c.P16(c.I.wd) + z.WD, c.I.Kd, 4 * z.Hd, z.Hd, c.s));
        // zero recurrent state of step 0: h, ctx (attention rows) and dec_h (decoder rows)
        T2_CHECK_HIP(hipMemsetAsync(c.RowA(0, 0), 0, (size_t)2 * 2 * z.B * c.I.Ka * sizeof(__bf16), c.s));
        T2_CHECK_HIP(hipMemsetAsync(c.RowD(0), 0, (size_t)2 * z.B * c.I.Kd * sizeof(__bf16), c.s));
    }
    // one launch per step for projection + stop rule + the next step's prenets (infer.hip)
    auto tail = [&](int t, bool proj) -> int {
        StepTailDesc d{};
        d.B = z.B; d.M = z.M; d.P = z.P; d.WO = z.WO; d.NS = z.NS; d.t = t;
        d.do_proj = proj; d.do_prenet = proj ? (t + 1 < T) : 1;
        const int tn = proj ? t + 1 : 0;                             // the step whose prenet outputs are produced
        if (proj) {
            d.dout = c.P(L.dout) + c.R(t) * z.WO; d.lddout = z.WO;
            d.proj_w = w->proj_w; d.proj_b = w->proj_b; d.gate_w = w->gate_w; d.gate_b = w->gate_b;
            d.mel_out = a->mel_out + (long)t * z.M; d.ldmel = (long)T * z.M;
            d.gate_out = a->gate_out + t; d.ldgate = T;
            d.thr = a->gate_threshold; d.stop_index = a->stop_index; d.done = a->done_count;
        }
        for (int s = 0; s < z.NS; ++s) {
            d.w1t[s] = c.P(L.w1t) + (size_t)s * z.M * z.P; d.w2[s] = s ? w->prenet_sub_w2 : w->prenet_w2;
            d.p1[s] = c.P(s ? L.p1s : L.p1) + c.R(tn) * z.P; d.p2[s] = c.P(s ? L.p2s : L.p2) + c.R(tn) * z.P;
            d.site1[s] = s ? T2_SITE_PRENET1_SUB : T2_SITE_PRENET1; d.site2[s] = s ? T2_SITE_PRENET2_SUB : T2_SITE_PRENET2;
        }
        d.ldp = z.P;
        if (c.use16) { for (int s = 0; s < z.NS; ++s) d.p2_16[s] = c.RowA(tn & 1, s) + z.Ha + z.E; d.ldp16 = c.I.Ka; }
        d.drop_p = c.prenet_dropout ? dims->p_prenet_dropout : 0.f; d.seed = c.seed;
        d.drop_base = (uint32_t)(c.R(tn) * z.P); d.drop_mstride = (uint32_t)z.P;      // logical [T,B,P]
        return step_tail(d, c.s);
    };
    int steps = 0;
    StopPoll* pl = nullptr;
    T2_TRY(stop_poll_get(&pl));
    ChainDesc cdec{};
    const bool chain = g_chain && chain_dec_desc(c, *w, *a, &cdec);
    if (chain && a->poll_every <= 0) poll = 32;                      // one persistent launch per polling interval: 32 steps amortise its start-up
    if (chain) {                                                     // status, counters, zero state of step -1 (h, ctx, dec_h, go-frame prenet = 0)
        const ChainBufs bufs = chain_bufs(z, L, a->ws);
        T2_CHECK_HIP(hipMemsetAsync(bufs.err, 0, 256 + 2 * kChainCntBytes + bufs.xa_bytes + bufs.xb_bytes + kChainXmBytes + bufs.q_bytes, c.s));   // (+ the tagged query partials)
    } else {
        const ChainBufs bufs = chain_bufs(z, L, a->ws);
        T2_CHECK_HIP(hipMemsetAsync(bufs.err, 0, 256, c.s));
        T2_TRY(tail(0, false));                                      // prenet of the go frame (model.py:444-450)
    }
    for (int t = 0; t < T; ++t) {
        if (chain) {
            if (t % poll == 0) {                                     // one persistent launch per polling interval
                cdec.t0 = t; cdec.t1 = std::min(T, t + poll);
                ProfScope ps(PK_CHAIN_DEC, c.s);
                T2_TRY(chain_fwd(cdec, c.s));
            }
        } else {
            T2_TRY(att_lstm_step(c, t));
            T2_TRY(attention_step(c, t));
            T2_TRY(dec_lstm_step(c, t));
            T2_TRY(tail(t, true));                                   // mel_t, gate_t, stop rule, prenets of step t+1 (:470-471)
        }
        steps = t + 1;
        if (steps % poll == 0 && steps < T) {
            // Stop rule without draining the queue: the counter is copied to pinned memory behind an event; the host reads
            // the copy made TWO polls ago, so it blocks only when it is more than 2*poll steps ahead of the GPU and the
            // GPU always has work queued.  The loop overshoots the last stop by at most 3*poll steps (their frames are
            // past every item's stop index).
            const int k = steps / poll, slot = k & 3;
            T2_CHECK_HIP(hipMemcpyAsync(pl->host + slot, a->done_count, sizeof(int32_t), hipMemcpyDeviceToHost, c.s));
            T2_CHECK_HIP(hipEventRecord(pl->ev[slot], c.s));
            if (k >= 3) {
                const int prev = (k - 2) & 3;
                T2_CHECK_HIP(hipEventSynchronize(pl->ev[prev]));
                if (pl->host[prev] >= z.B) break;
            }
        }
    }
    *a->steps_run_host = steps;
    if (chain) {
        hipLaunchKernelGGL(poison_if_aborted_kernel, dim3(64), dim3(256), 0, c.s, chain_bufs(z, L, a->ws).err, 1, a->mel_out, (size_t)z.B * T * z.M, a->gate_out, (size_t)z.B * T);
        T2_LAUNCH_CHECK();
    }
    return 0;
}


int t2_conv_bn_forward(const t2_conv_bn_args* a, void* stream) {
    T2_REQUIRE(a, "null argument");
    const size_t nw = (size_t)a->Cout * a->Cin * a->K;
    T2_REQUIRE(a->ws_floats >= nw + 128 * (size_t)a->Cout, "t2_conv_bn_forward: workspace too small");
    ConvBnFwd f{};
    f.x = a->x; f.B = a->B; f.T = a->T; f.Cin = a->Cin; f.Cout = a->Cout; f.K = a->K;
    f.w = a->w; f.bias = a->bias; f.gamma = a->gamma; f.beta = a->beta; f.run_mean = a->run_mean; f.run_var = a->run_var;
    f.training = a->training; f.eps = a->eps; f.act = a->act; f.drop_p = a->drop_p; f.seed = a->seed; f.site = a->site;
    f.residual = a->residual; f.z = a->z; f.mean = a->mean; f.invstd = a->invstd; f.var = a->var; f.y = a->y;
    f.wperm = a->ws; f.scratch = a->ws + align4(nw);
    const size_t used = align4(nw) + 128 * (size_t)a->Cout;
    if (a->ws_floats > used) { f.gemm_ws = a->ws + used; f.gemm_ws_bytes = (a->ws_floats - used) * sizeof(float); }
    return conv_bn_fwd(f, (hipStream_t)stream);
}

int t2_conv_bn_backward(const t2_conv_bn_bwd_args* a, void* stream) {
    T2_REQUIRE(a, "null argument");
    const size_t nw = align4((size_t)a->Cout * a->Cin * a->K), ndz = align4((size_t)a->B * a->T * a->Cout), nsc = 128 * (size_t)a->Cout;
    T2_REQUIRE(a->ws_floats >= nw + ndz + nsc, "t2_conv_bn_backward: workspace too small");
    ConvBnBwd f{};
    f.x = a->x; f.B = a->B; f.T = a->T; f.Cin = a->Cin; f.Cout = a->Cout; f.K = a->K;
    f.w = a->w; f.gamma = a->gamma; f.beta = a->beta; f.z = a->z; f.mean = a->mean; f.invstd = a->invstd;
    f.training = a->training; f.eps = a->eps; f.act = a->act; f.drop_p = a->drop_p; f.seed = a->seed; f.site = a->site;
    f.dy = a->dy; f.dw = a->dw; f.dbias = a->dbias; f.dgamma = a->dgamma; f.dbeta = a->dbeta;
    f.dx = a->dx; f.dx_accumulate = a->dx_accumulate;
    f.dz = a->ws; f.wperm = a->ws + ndz; f.scratch = a->ws + ndz + nw;
    f.gemm_ws = a->ws + ndz + nw + nsc; f.gemm_ws_bytes = (a->ws_floats - (ndz + nw + nsc)) * sizeof(float);
    return conv_bn_bwd(f, (hipStream_t)stream);
}

int t2_embedding_forward(const int64_t* ids, const float* table, float* out, int rows, int dim, void* stream) {
    return embedding_fwd(reinterpret_cast<const long*>(ids), table, out, rows, dim, (hipStream_t)stream);
}
int t2_embedding_backward(const int64_t* ids, const float* dout, float* dtable, int rows, int dim, int vocab, void* stream) {
    return embedding_bwd(reinterpret_cast<const long*>(ids), dout, dtable, rows, dim, vocab, (hipStream_t)stream);
}

int t2_lstm_seq_forward(const t2_lstm_seq_args* a, void* stream) {
    T2_REQUIRE(a && a->nstreams >= 1 && a->nstreams <= kMaxLstmStreams, "t2_lstm_seq_forward: bad nstreams");
    T2_REQUIRE(a->H % 64 == 0, "t2_lstm_seq_forward: H=%d must be a multiple of 64", a->H);
    const int B = a->B, T = a->T, H = a->H;
    // every step of both directions in one persistent launch (chain_enc.hip) when the shape is covered and the caller
    // gave exchange space; otherwise one launch per step
    if (g_chain && a->ws && a->ws_floats >= enc_chain_ws_floats(a->nstreams, B, H, 0) && enc_chain_covers(a->nstreams, B, H)) {
        EncChainDesc d{};
        d.ND = a->nstreams; d.B = B; d.T = T; d.H = H; d.lengths = a->lengths; d.ldh = a->ldh;
        for (int s = 0; s < a->nstreams; ++s) {
            d.pre[s] = a->pre[s]; d.w_hh[s] = a->w_hh[s]; d.reverse[s] = a->reverse[s];
            d.h[s] = a->h[s]; d.c[s] = a->c[s]; d.gates[s] = a->gates[s];
        }
        return enc_chain_fwd(d, a->ws, a->ws_floats, (hipStream_t)stream);
    }
    for (int step = 0; step < T; ++step) {
        LstmStepDesc d{};
        d.nstreams = a->nstreams; d.B = B; d.H = H; d.drop_p = 0.f;
        for (int s = 0; s < a->nstreams; ++s) {
            LstmStream& st = d.st[s];
            const int t = a->reverse[s] ? T - 1 - step : step;
            const int tp = a->reverse[s] ? t + 1 : t - 1;            // time index processed in the previous step
            st.pre = a->pre[s] + (long)t * B * 4 * H; st.ldpre = 4 * H;
            if (step > 0) {
                st.seg[0] = LstmSeg{a->h[s] + (long)tp * B * a->ldh, a->ldh, a->w_hh[s], (long)H, H};
                st.nseg = 1;
                st.c_prev = a->c[s] + (long)tp * B * H; st.ldc_prev = H;
            }
            st.gates = a->gates[s] + (long)t * B * 4 * H; st.ldgates = 4 * H;
            st.c_out = a->c[s] + (long)t * B * H; st.ldc_out = H;
            st.h_out = a->h[s] + (long)t * B * a->ldh; st.ldh_out = a->ldh;
            st.lengths = a->lengths; st.t = t;
        }
        T2_TRY(lstm_step_fwd(d, (hipStream_t)stream));
    }
    return 0;
}

size_t t2_lstm_seq_chain_ws_floats(int nstreams, int B, int H, int backward) { return enc_chain_ws_floats(nstreams, B, H, backward); }

int t2_lstm_seq_backward(const t2_lstm_seq_bwd_args* a, void* stream) {
    T2_REQUIRE(a && a->nstreams >= 1 && a->nstreams <= kMaxLstmStreams, "t2_lstm_seq_backward: bad nstreams");
    const int B = a->B, T = a->T, H = a->H, ns = a->nstreams;
    const int ks = lstm_bwd_ksplit(4 * H);
    hipStream_t s = (hipStream_t)stream;
    float* dc = a->ws;                                        // [ns][B*H]
    float* part = dc + align4((size_t)ns * B * H);            // [ns][ks][B][H]
    float* gws = part + align4((size_t)ns * ks * B * H);
    T2_REQUIRE(a->ws_floats >= (size_t)(gws - a->ws), "t2_lstm_seq_backward: workspace too small");
    const size_t gws_bytes = (a->ws_floats - (size_t)(gws - a->ws)) * sizeof(float);
    const bool chain = g_chain && g_chain_bwd && a->ws_floats >= enc_chain_ws_floats(ns, B, H, 1) && enc_chain_covers(ns, B, H);
    if (chain) {                                               // whole BPTT in one persistent launch (chain_enc.hip); its exchange space = this scratch
        EncChainBwdDesc d{};
        d.ND = ns; d.B = B; d.T = T; d.H = H; d.lddh = a->lddh;
        for (int i = 0; i < ns; ++i) {
            d.w_hh[i] = a->w_hh[i]; d.reverse[i] = a->reverse[i]; d.c[i] = a->c[i]; d.gates[i] = a->gates[i];
            d.dh[i] = a->dh[i]; d.dpre[i] = a->dpre[i];
        }
        T2_TRY(enc_chain_bwd(d, a->ws, a->ws_floats, s));
    }
    for (int step = 0; step < (chain ? 0 : T); ++step) {      // reverse of the processing order
        LstmBwdPointDesc p{};
        p.nstreams = ns; p.B = B; p.H = H; p.drop_p = 0.f; p.first = step == 0;
        LstmBwdGemmDesc g{};
        g.nstreams = ns; g.B = B; g.H4 = 4 * H; g.KS = ks; g.NC = H;
        for (int i = 0; i < ns; ++i) {
            // processing order: forward dir t = 0..T-1, reverse dir t = T-1..0; BPTT visits it backwards
            const int t = a->reverse[i] ? step : T - 1 - step;
            const int tprev = a->reverse[i] ? t + 1 : t - 1;  // the step whose state entered step t
            const bool has_prev = a->reverse[i] ? (t + 1 < T) : (t > 0);
            LstmBwdStream& st = p.st[i];
            st.dh1 = a->dh[i] + (long)t * B * a->lddh; st.lddh1 = a->lddh;
            st.part = part + (size_t)i * ks * B * H; st.nparts = ks; st.part_stride = (long)B * H; st.ldpart = H; st.part_col = 0;
            st.gates = a->gates[i] + (long)t * B * 4 * H; st.ldgates = 4 * H;
            st.c_new = a->c[i] + (long)t * B * H; st.ldc_new = H;
            if (has_prev) { st.c_prev = a->c[i] + (long)tprev * B * H; st.ldc_prev = H; }
            st.dc_state = dc + (size_t)i * B * H;
            st.dg = a->dpre[i] + (long)t * B * 4 * H; st.lddg = 4 * H;
            g.st[i].dg = st.dg; g.st[i].lddg = 4 * H;
            g.st[i].seg[0] = LstmBwdSeg{a->w_hh[i], (long)H, H}; g.st[i].nseg = 1;
            g.st[i].part = part + (size_t)i * ks * B * H;
        }
        T2_TRY(lstm_bwd_pointwise(p, s));
        if (step + 1 < T) T2_TRY(lstm_bwd_gemm(g, s));
    }
    // d(W_hh) = sum_t dpre(t)^T . h(previous processed step)
    for (int i = 0; i < ns; ++i) {
        if (T < 2) { T2_TRY(fill_f32(a->dw_hh[i], 0.f, (size_t)4 * H * H, s)); continue; }
        GemmDesc w = gemm_desc();
        const long rowsB = (long)B;
        w.A = a->reverse[i] ? a->dpre[i] : a->dpre[i] + rowsB * 4 * H; w.sam = 1; w.sak = 4 * H;
        w.B = a->reverse[i] ? a->h[i] + rowsB * a->ldh : a->h[i]; w.sbk = a->ldh; w.sbn = 1;
        w.C = a->dw_hh[i]; w.ldc = H; w.M = 4 * H; w.N = H; w.K = (T - 1) * B;
        w.ws = gws; w.ws_bytes = gws_bytes;
        T2_TRY(gemm(w, s));
    }
    return 0;
}

int t2_gemm_ex(const t2_gemm_args* a, void* stream) {
    T2_REQUIRE(a, "null argument");
    GemmDesc g = gemm_desc();
    g.A = a->A; g.B = a->B; g.C = a->C; g.M = a->M; g.N = a->N; g.K = a->K;
    g.sam = a->sam; g.sak = a->sak; g.sbn = a->sbn; g.sbk = a->sbk; g.ldc = a->ldc;
    g.batch = a->batch > 0 ? a->batch : 1; g.bsA = a->bsA; g.bsB = a->bsB; g.bsC = a->bsC;
    g.alpha = a->alpha; g.beta = a->beta; g.bias1 = a->bias; g.act = a->act;
    g.crow_mod = a->crow_mod; g.crow_mul = a->crow_mul;
    g.ws = a->ws; g.ws_bytes = a->ws_bytes; g.splitk = a->splitk;
    return gemm(g, (hipStream_t)stream);
}
// bench.py's GEMM roofline figure: `reps` launches of one product bracketed by HIP events on the launch stream, once as
// t2_gemm_ex runs it (fp32 operands in, staging casts included) and once with both bf16 operand copies made beforehand,
// so that the second figure is the matrix kernel (+ its split-K reduce) alone.
int t2_prof_gemm(const t2_gemm_args* a, int reps, float* ms_total, float* ms_kernel, void* stream) {
    T2_REQUIRE(a && reps > 0 && ms_total && ms_kernel, "t2_prof_gemm: bad arguments");
    T2_REQUIRE(get_precision() == 1 && a->M % 128 == 0 && a->N % 128 == 0 && a->K % 64 == 0 && (a->batch <= 1) && a->ws,
               "t2_prof_gemm: bf16 mode, whole-tile shapes and scratch only");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    T2_CHECK_HIP(hipEventCreate(&e0)); T2_CHECK_HIP(hipEventCreate(&e1));
    int rc = t2_gemm_ex(a, stream);                                       // warm-up
    T2_CHECK_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps && rc == 0; ++i) rc = t2_gemm_ex(a, stream);
    T2_CHECK_HIP(hipEventRecord(e1, s));
    T2_CHECK_HIP(hipEventSynchronize(e1));
    T2_CHECK_HIP(hipEventElapsedTime(ms_total, e0, e1));
    *ms_total /= reps;
    if (rc == 0) {
        const size_t na = ((size_t)a->M * a->K * sizeof(__bf16) + 255) & ~(size_t)255, nb = ((size_t)a->N * a->K * sizeof(__bf16) + 255) & ~(size_t)255;
        T2_REQUIRE(a->ws_bytes > na + nb, "t2_prof_gemm: scratch too small");
        unsigned char* w8 = reinterpret_cast<unsigned char*>(a->ws);
        __bf16* a16 = reinterpret_cast<__bf16*>(w8); __bf16* b16 = reinterpret_cast<__bf16*>(w8 + na);
        const bool akc = a->sak == 1, bkc = a->sbk == 1;
        rc = stage_bf16(a->A, akc, akc ? a->sam : a->sak, a16, a->M, a->K, s);
        if (rc == 0) rc = stage_bf16(a->B, bkc, bkc ? a->sbn : a->sbk, b16, a->N, a->K, s);
        GemmDesc g = gemm_desc();
        g.A = a->A; g.B = a->B; g.C = a->C; g.M = a->M; g.N = a->N; g.K = a->K;
        g.sam = a->sam; g.sak = a->sak; g.sbn = a->sbn; g.sbk = a->sbk; g.ldc = a->ldc;
        g.alpha = a->alpha; g.beta = a->beta; g.bias1 = a->bias; g.act = a->act; g.splitk = a->splitk;
        g.A16 = a16; g.lda16 = a->K; g.B16 = b16; g.ldb16 = a->K;
        g.ws = reinterpret_cast<float*>(w8 + na + nb); g.ws_bytes = a->ws_bytes - na - nb;
        if (rc == 0) rc = gemm(g, s);
        T2_CHECK_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < reps && rc == 0; ++i) rc = gemm(g, s);
        T2_CHECK_HIP(hipEventRecord(e1, s));
        T2_CHECK_HIP(hipEventSynchronize(e1));
        T2_CHECK_HIP(hipEventElapsedTime(ms_kernel, e0, e1));
        *ms_kernel /= reps;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}
int t2_colsum(const float* x, long ld, int M, int N, float* out, float* scratch, void* stream) {
    return colsum(x, ld, M, N, out, nullptr, scratch, (hipStream_t)stream);
}
int t2_mask_btc(float* x, int B, int T, int C, const int32_t* lengths, float fill, void* stream) {
    T2_REQUIRE(lengths, "t2_mask_btc: lengths is null");
    return mask_btc(x, B, T, C, lengths, fill, (hipStream_t)stream);
}

int t2_prof_enable(int max_launches) {
    if (max_launches <= 0) { g_prof.on = false; return 0; }
    const size_t need = (size_t)max_launches * 2;
    while (g_prof.ev.size() < need) {
        hipEvent_t e;
        T2_CHECK_HIP(hipEventCreate(&e));
        g_prof.ev.push_back(e);
    }
    g_prof.kind.assign(g_prof.ev.size() / 2, 0);
    g_prof.used = 0;
    g_prof.on = true;
    return 0;
}

int t2_prof_collect(int n_kinds, double* total_ms_host, int* launches_host) {
    T2_REQUIRE(n_kinds >= PK_COUNT, "t2_prof_collect: need %d kinds", (int)PK_COUNT);
    g_prof.on = false;
    for (int i = 0; i < n_kinds; ++i) { total_ms_host[i] = 0.0; launches_host[i] = 0; }
    if (g_prof.used == 0) return 0;
    T2_CHECK_HIP(hipDeviceSynchronize());                // events live on two streams
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float ms = 0.f;
        T2_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
        total_ms_host[g_prof.kind[i / 2]] += ms;
        launches_host[g_prof.kind[i / 2]] += 1;
    }
    g_prof.used = 0;
    return 0;
}

int t2_adam_chunks(long numel) { return adam_chunks(numel); }
int t2_adam_step(const t2_adam_tensor* table, int n_tensors, int n_chunks, float* partial, float* norm_out, float max_norm,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
    static_assert(sizeof(t2_adam_tensor) == sizeof(AdamTensor), "table row layout");
    return adam_step(reinterpret_cast<const AdamTensor*>(table), n_tensors, n_chunks, partial, norm_out, max_norm, lr, beta1, beta2, eps,
                     weight_decay, step, (hipStream_t)stream);
}

int t2_adam_norm(const t2_adam_tensor* table, int n_tensors, int n_chunks, float* partial, float* norm_out, float max_norm, void* stream) {
    return adam_norm(reinterpret_cast<const AdamTensor*>(table), n_tensors, n_chunks, partial, norm_out, max_norm, (hipStream_t)stream);
}

int t2_finalize_bct(const float* in_btc, float* out_bct, int B, int T, int C, const int32_t* lengths, float fill, void* stream) {
    return transpose_btc_to_bct(in_btc, out_bct, B, T, C, lengths, fill, (hipStream_t)stream);
}
int t2_mask_bt(float* x, int B, int T, const int32_t* lengths, float fill, void* stream) {
    T2_REQUIRE(lengths, "t2_mask_bt: lengths is null");
    return mask_bt(x, B, T, lengths, fill, (hipStream_t)stream);
}

int t2_gemm(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbn, long sbk, long ldc,
            const float* bias, int act, float alpha, float beta, float* ws, size_t ws_bytes, int splitk, void* stream) {
    GemmDesc g = gemm_desc();
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.sam = sam; g.sak = sak; g.sbn = sbn; g.sbk = sbk; g.ldc = ldc;
    g.bias1 = bias; g.act = act; g.alpha = alpha; g.beta = beta;
    g.ws = ws; g.ws_bytes = ws_bytes; g.splitk = splitk;
    return gemm(g, (hipStream_t)stream);
}
int t2_rng_keep_mask(uint64_t seed, uint32_t site, uint32_t n, float p, uint8_t* out, void* stream) {
    return rng_keep_mask(seed, site, n, p, out, (hipStream_t)stream);
}
int t2_rng_normal(uint64_t seed, uint32_t site, uint32_t n, float* out, void* stream) {
    return rng_normal(seed, site, n, out, (hipStream_t)stream);
}

}  // extern "C"
