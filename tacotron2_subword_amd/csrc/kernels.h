// Internal launcher interface between the C-ABI layer (c_api.hip) and the kernel files.
// Everything takes raw device pointers and enqueues on the given stream; nothing allocates
// or synchronises.
#pragma once
#include "common.h"

namespace t2 {

// ------------------------------------------------------------------ GEMM (gemm.hip)
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct GemmDesc {
    const float* A; const float* B; float* C;
    int M, N, K;
    long sam, sak;      // A(m,k) = A[m*sam + k*sak]   (one of the two strides must be 1)
    long sbn, sbk;      // B(k,n) = B[n*sbn + k*sbk]   (one of the two strides must be 1)
    long ldc;           // C(m,n) = C[m*ldc + n]
    int batch; long bsA, bsB, bsC;
    float alpha, beta;  // C = act(alpha*A.B + bias1 + bias2) [dropout] + beta*C_old
    const float* bias1; const float* bias2;   // per-n, nullable
    int act;
    float drop_p; uint64_t seed; uint32_t site;   // dropout keep-bit index = drop_base + m*drop_mstride + n (batch==1 only)
    uint32_t drop_base, drop_mstride;             // drop_mstride 0 -> N
    float* ws; size_t ws_bytes;                   // split-K scratch (nullable -> no split)
    int splitk;                                   // 0 = choose automatically
    int conv_a, conv_b, conv_T, conv_C, conv_pad; // implicit im2col operand (gemm.hip ConvAddr): A rows / B k-rows are
                                                  // frames of X[B*T, C], the other index is dk*C + ci
    int fp32_only;                                // never use the bf16-operand kernel for this product
    const __bf16* A16; long lda16;                // optional pre-staged bf16 copies of the operands, K contiguous:
    const __bf16* B16; long ldb16;                // A16[m*lda16 + k], B16[n*ldb16 + k] (bf16 mode, whole 128x128x64 tiles)
    int crow_mod; long crow_mul;                  // output row = (m % crow_mod) * crow_mul + m / crow_mod (0 = identity):
                                                  // writes time-major rows (t,b) in batch-major order (b,t) or back
};
inline GemmDesc gemm_desc() {
    GemmDesc d{}; d.batch = 1; d.alpha = 1.f; d.beta = 0.f; d.act = ACT_NONE; d.drop_p = 0.f; return d;
}
int gemm(const GemmDesc& d, hipStream_t s);
// bf16 copy of an fp32 operand, K contiguous: dst[row*K + k] = src[row*ld + k] (kc) or src[k*ld + row] (!kc);
// rows % 64 == 0, K % 64 == 0, 16-byte aligned rows.  For GemmDesc::A16 / B16 shared by several products.
int stage_bf16(const float* src, bool kc, long ld, __bf16* dst, int rows, int K, hipStream_t s);
void set_precision(int p);   // 0 fp32 operands, 1 bf16 operands (fp32 accumulate) for large GEMMs and LSTM steps
int get_precision();
void set_gemm_staging(int on);   // bf16 mode: stage fp32 operands as bf16 copies for the bf16-source kernel (default on)

// ------------------------------------------------------------------ LSTM (lstm.hip)
constexpr int kMaxSeg = 6;
struct LstmSeg { const float* x; long ldx; const float* w; long ldw; int k; };
struct LstmStream {
    LstmSeg seg[kMaxSeg]; int nseg;
    const float* pre; long ldpre;         // [B,4H] input-side pre-activations (nullable)
    const float* bias1; const float* bias2;   // [4H] nullable (added when given)
    const float* c_prev; long ldc_prev;   // [B,H]
    float* gates; long ldgates;           // [B,4H] activated i,f,g,o (nullable)
    float* c_new; long ldc_new;           // [B,H] cell before dropout (nullable)
    float* h_out; long ldh_out;           // [B,H] after dropout
    float* h_out2; long ldh_out2;         // optional second copy of h (nullable)
    float* c_out; long ldc_out;           // [B,H] after dropout
    uint32_t site_h, site_c; uint32_t idx_base, idx_bstride;   // dropout index = idx_base + b*idx_bstride + u
    const int* lengths; int t;            // packed sequences: item active iff t < lengths[b] (nullable);
                                          // an inactive item writes zeros (state and output)
    const float* wq; int A; float* qpart; // optional fused query partials: qpart[H/8][B][A] (wq: [A,H])
    // bf16-operand mode (one K-contiguous segment): x16 [B,k16] and w16 [4H,k16] shadows; h16_out = bf16 copy of h
    const __bf16* x16; long ldx16; const __bf16* w16; long ldw16; int k16;
    __bf16* h16_out; long ldh16;
    __bf16* h16_out2; long ldh16_2;       // second bf16 destination (decode loop: the decoder LSTM's input row)
};
constexpr int kMaxLstmStreams = 4;
struct LstmStepDesc { LstmStream st[kMaxLstmStreams]; int nstreams; int B, H; float drop_p; uint64_t seed; };
int lstm_step_fwd(const LstmStepDesc& d, hipStream_t s);

// Backward of one LSTM step, part 1 (pointwise): total dL/dh_out(t), dL/dc_out(t) -> dL/d(gate pre-activations).
struct LstmBwdStream {
    const float* dh1; long lddh1;          // [B,H] direct gradient on h_out(t) (nullable)
    const float* dh2; long lddh2;          // second direct source (nullable)
    const float* part; int nparts; long part_stride; long ldpart; int part_col;   // recurrent partials from
                                           // lstm_bwd_gemm of step t+1: sum_z part[z*part_stride + b*ldpart + part_col + u]
    const float* dq; long lddq; const float* wq; int A;   // + dq[b,:] . wq[:,u]  (query projection, nullable)
    int dq_parts;                          // dq row = dq_parts partials of A columns each, summed here (0 -> 1)
    const float* gates; long ldgates;      // saved activated gates of step t
    const float* c_new; long ldc_new;      // saved cell before dropout
    const float* c_prev; long ldc_prev;    // cell carried INTO step t (after dropout), null -> 0
    float* dc_state;                       // [B,H] running dL/dc_out, read (unless first) and replaced by dL/dc_prev
    float* dg; long lddg;                  // [B,4H] out: gradient wrt gate pre-activations
    __bf16* dg16;                          // optional dense [B,4H] bf16 copy (bf16-operand recurrent GEMM)
    uint32_t site_h, site_c, idx_base, idx_bstride;
};
struct LstmBwdPointDesc { LstmBwdStream st[kMaxLstmStreams]; int nstreams; int B, H; float drop_p; uint64_t seed; int first; };
int lstm_bwd_pointwise(const LstmBwdPointDesc& d, hipStream_t s);

// part 2 (skinny GEMM): partial products dL/dx_rec = dg(t) . W for the recurrent input columns,
// split over column tiles x K-splits so that the whole chip works on one step.
struct LstmBwdSeg { const float* w; long ldw; int ncols; };
struct LstmBwdGemmStream {
    const float* dg; long lddg; LstmBwdSeg seg[3]; int nseg; float* part;
    const __bf16* dg16; const __bf16* wt16;   // bf16 variant: dense [B,4H] gradient copy, transposed weight shadow [NC,4H]
};
struct LstmBwdGemmDesc { LstmBwdGemmStream st[kMaxLstmStreams]; int nstreams; int B, H4, KS, NC; };
int lstm_bwd_gemm(const LstmBwdGemmDesc& d, hipStream_t s);
int lstm_bwd_ksplit(int H4);
// bf16 shadows: dst[r*ld_dst + c] = bf16(src[r*ld_src + c]) ; transposed: dst[c*ld_dst + r]
int cast_rows_bf16(const float* src, long ld_src, __bf16* dst, long ld_dst, int R, int C, hipStream_t s);
int cast_transpose_bf16(const float* src, long ld_src, __bf16* dst, long ld_dst, int R, int C, hipStream_t s);

// ------------------------------------------------------------------ attention (attention.hip)
// DynamicConvolutionAttention (attention.py:195-289): 8 static + 8 dynamic channels, 21 taps, 11-tap prior
constexpr int kDcaC = 8, kDcaK = 21, kDcaP = 11, kDcaPad = (kDcaK - 1) / 2;
struct DcaWeights {                       // wq / qpart of the stream carry W (Linear Ha->A)
    const float* bW;                      // W.bias [A]
    const float* V;                       // V.weight [C*K, A]
    const float* F;                       // F.weight [C, 1, K]
    const float* U;                       // U.weight [A, C]
    const float* T; const float* bT;      // T.weight [A, C], T.bias [A]
    const float* v;                       // v.weight [1, A]
    const float* P;                       // prior buffer [11] (already flipped, attention.py:221)
};
struct AttnStream {
    const float* query; long ldq;         // [B,A] processed query (W_q h), or null when qpart is given
    const float* qpart; int nparts;       // [nparts][B][A] partial queries from lstm_step_fwd (summed in order)
    float* q_out; long ldq_out;           // [B,A] the query actually used, saved for backward (nullable)
    const float* pm;                      // [B,Tin,A] processed memory
    const float* memory;                  // [B,Tin,E]
    const int* lengths;                   // [B] valid memory length (nullable = all valid)
    const float* a_prev; long lda_prev;   // [B,Tin] previous alignment / weights (nullable at t=0 -> init)
    float* a_out; long lda_out;           // [B,Tin] new alignment / weights
    float* p_out; long ldp_out;           // [B,Tin] SMA selection probability (nullable)
    const float* wcum_prev; long ldwcum_prev;   // LSA cumulative weights before this step (null at the first step)
    float* wcum_out; long ldwcum_out;           // LSA cumulative weights after this step (nullable for SMA)
    float* ctx1; long ldctx1;             // [B,E]
    float* ctx2; long ldctx2;             // second destination (nullable)
    __bf16* ctx16; long ldctx16;          // bf16 copy for the bf16-operand LSTM step (nullable)
    __bf16* ctx16b; long ldctx16b;        // second bf16 destination (decode loop)
    const float* v;                       // [A]
    const float* loc_conv; const float* loc_dense;   // LSA: [F,2,Kc], [A,F]
    uint32_t site_noise; uint32_t idx_base, idx_bstride;   // SMA noise index = idx_base + b*idx_bstride + j
    int Tin;
    // GMM attention (kind 2): wq / qpart carry mlp.0 (Linear Ha->A); second layer and state below
    const float* gmm_b1; const float* gmm_w2; const float* gmm_b2;   // [A], [3K, A], [3K]
    const float* mu_prev; float* mu_out;  // [B, kGmmPad] mixture means before / after this step (mu_prev null at t=0 -> 0)
    DcaWeights dca;                       // kind 3
    float mask_value;                     // energy of positions past the item's length (the module's score_mask_value)
};
constexpr int kGmmK = 5, kGmmPad = 8;     // mixtures (attention.py:409), row pitch of the mean buffers
struct AttnStepDesc {
    AttnStream st[2]; int nstreams; int B, A, E; int kind;   // kind 0 = SMA, 1 = LSA
    int F, Kc; float noise_std; uint64_t seed; int first;                     // kind 2 = GMM (attention_gmm_step_fwd)
    int lsa_pa;                                               // set by the launcher (MFMA path of the LSA dense projection)
    int max_pos;                                              // > 0: valid length clamped to max_pos (ForwardAttentionV2, see c_api.hip)
};
int attention_step_fwd(const AttnStepDesc& d, hipStream_t s);

// Backward of one attention step (reverse time): SMA (kind 0) or LSA (kind 1).
struct AttnBwdStream {
    const float* dctx[3]; long lddctx[3];  // direct gradient sources on ctx(t) [B,E] (nullable entries)
    const float* part; int nparts; long part_stride; long ldpart; int part_col;   // recurrent partials (ctx columns)
    const float* dalign; long lddalign;    // optional external gradient on the alignment row [B,Tin]
    const float* q; long ldq;              // saved query [B,A]
    const float* pm; const float* memory;  // [B,Tin,A], [B,Tin,E]
    const float* p; long ldp;              // saved selection probability of step t
    const float* a_prev; long lda_prev;    // alignment of step t-1 (null at t=0 -> one-hot)
    const float* v;
    float* carry;                          // [B,Tin] gradient flowing into a_{t} from step t+1 (LSA: updated in place)
    float* carry_out;                      // SMA: [B,Tin] gradient for step t-1 (a second buffer; the two swap every step)
    float* dctx_out; long lddctx_out;      // [B,E] total ctx gradient, saved for the d(memory) GEMM
    float* dq_out; long lddq_out;          // [B, nsplit*A]: one partial per position split (columns split*A ...)
    float* dv_acc;                         // [nsplit][B][A] accumulated over steps
    float* dpm_acc;                        // [B,Tin,A] accumulated over steps
    int Tin;
    // LSA only
    const float* w; long ldw;              // saved attention weights of step t [B,Tin] (a_prev = weights of step t-1)
    const float* wcum_prev; long ldwcum_prev;   // cumulative weights before step t (null at t=0)
    const float* loc_conv; const float* loc_dense;   // [F,2,Kc], [A,F]
    float* carry_cum;                      // [B,Tin] gradient on the cumulative weights (in/out); `carry` holds the w_{t-1} part
    float* dconv_acc; float* ddense_acc;   // [B,F*2*Kc], [B,A*F] per-item weight gradients accumulated over steps
    // GMM only (kind 2): q = saved pre-activation of mlp.0 (incl. bias); w = saved weights of step t
    const float* gmm_w2; const float* gmm_b2;
    const float* mu; long ldmu;            // [B, kGmmPad] mixture means of step t
    float* mu_carry;                       // [B, kGmmPad] gradient on the means flowing in from step t+1 (in/out)
    float* dw2_acc; float* db2_acc;        // [B, 3K*A], [B, 16] per-item accumulators
    // DCA only (kind 3): q = saved pre-activation of W (incl. bias); w = weights of step t; a_prev = weights of step t-1
    // (null at t=0 -> one-hot at 0); carry = gradient on a_{t} from step t+1 (in/out)
    DcaWeights dca;
    float* dca_acc;                        // [B][A (dv) + A (dbT) + A*C (dU) + A*C (dT) + C*K (dF) + C*K*A (dV)] per-item accumulators
};
__host__ __device__ inline size_t dca_acc_floats(int A) { return (size_t)2 * A + 2 * (size_t)A * kDcaC + kDcaC * kDcaK + (size_t)kDcaC * kDcaK * A; }
struct AttnBwdDesc { AttnBwdStream st[2]; int nstreams; int B, A, E; int first; int kind, F, Kc; int nsplit; };   // nsplit: SMA only
int attention_step_bwd(const AttnBwdDesc& d, hipStream_t s);


// ------------------------------------------------------------------ persistent decoder chains (chain.hip)
// Many consecutive steps of a teacher-forced recurrence in one launch, recurrent weights resident in registers.
enum ChainKind : int { CHAIN_LSTM = 0, CHAIN_SMA = 1, CHAIN_LSA = 2 };
struct ChainStream {
    const __bf16* w16; long ldw16;        // [4H][K] K-contiguous bf16 shadow: [W_hh | W_ih[:, P:]] (K = H + E) or W_hh (K = H)
    const float* pre;                     // [T][B][4H] hoisted input-side pre-activations (+ biases)
    const float* wq;                      // [A][H] query projection (attention kinds)
    float* gates; float* c_new; float* c_out;   // saved activations [T][B][4H], [T][B][H], [T][B][H]
    float* h_out; long ldh;               // h(t) fp32: h_out[(t*B + b)*ldh + u]
    __bf16* h16_out; long ldh16;          // bf16 copy, same indexing
    int coff, ctx2off;                    // column of this stream's context in a DIN / DOUT row
    const float* pm; const float* memory; const int* lengths; int Tin;   // [B,Tin,A], [B,Tin,E], [B] (nullable)
    float* align; float* psel; float* wcum; float* qs;                   // [B,T,Tin] x3, [T,B,A]
    const float* v; const float* loc_conv; const float* loc_dense;
    float* usave; float* locsave;         // LSA, teacher-forced: tanh tile [T][B][A][Tin rounded up to 4] and location features [T][B][Tin][F] for the backward chain (nullable)
    uint32_t site_h, site_c, site_noise;
    float mask_value;                     // energy of positions past the item's length
};
struct ChainDesc {
    ChainStream st[2]; int NS, B, T, t0, t1;
    int H, E, A, WD, WO;
    float* din; __bf16* din16; float* dout;        // [T][B][WD] (fp32 + bf16 shadow), [T][B][WO]
    int kind, F, Kc, max_pos;
    float drop_p, noise_std; uint64_t seed;
    unsigned char* X; float* Q; unsigned* cnt; unsigned* err; unsigned q_bytes;   // exchange buffers (chain_exchange_bytes)
    int UT, RT, CS;                                // tiling chosen by chain_plan
    // decode loop (dec = 1; Decoder.inference, model.py:430-492): whole-cell shadows [W_hh | W_ih[:,P:] | W_ih[:,:P]] in st[].w16,
    // biases instead of hoisted pre-activations, and the decoder LSTM, the projections + stop rule and both prenets inside the launch
    int dec, P, M, Hd;
    const float* bias1[2]; const float* bias2[2];  // attention-LSTM biases [4H]
    float* att_c[2];                               // [B][H] attention cell states carried between launches
    const __bf16* wd16; long ldwd;                 // decoder-LSTM shadow [4Hd][WD + Hd] = [W_ih | W_hh]
    const float* dbias1; const float* dbias2; float* dec_c;   // [4Hd] x2, [B][Hd] cell state between launches
    const float* proj_w; const float* proj_b; const float* gate_w; const float* gate_b;   // [M][WO], [M], [WO], [1]
    const float* pw1[2]; const float* pw2[2];      // prenet weights per stream [P][M], [P][P]
    float* mel_out; long ldmel; float* gate_out; long ldgate;   // mel_out[b*ldmel + t*M + m], gate_out[b*ldgate + t]
    float thr; int32_t* stop_index; int32_t* done;
    float pdrop; uint32_t psite1[2], psite2[2];
    unsigned char* XD; unsigned char* XM;          // exchange: dec_h fragments [2][Hd/16][1 KB], mel fragments [2][6][1 KB]
    int lds_Tin, lds_Jp, lds_Jm; int Jp[2], Jm[2]; // LDS residency: processed-memory / memory rows kept on chip
};
bool chain_plan(ChainDesc& d);                     // fills UT/RT/CS and the residency fields; false = shape not covered
size_t chain_exchange_bytes(const ChainDesc& d, size_t* x_bytes, size_t* q_bytes);
constexpr int kCntShards = 16;                     // every arrival counter is kept as 16 shards, each on a 128-byte line of its own (chain_common.h)
constexpr size_t kChainCntBytes = (size_t)2 * 4 * 2 * 128 * kCntShards; // arrival counters: [NS][row groups <= 4][2] sharded counters
int chain_fwd(const ChainDesc& d, hipStream_t s);
int chain_device_cus();
// c_api.hip: the current device's sticky status words (page-locked host memory the device writes directly; the pointer is
// valid on both sides), and this process's claim on the device's persistent kernels (one process per GPU)
unsigned* chain_sticky_words();
bool chain_device_claim();

// Backward (BPTT) of a chain in one launch (chain_bwd.hip)
struct ChainBwdStream {
    const __bf16* wt16; long ldwt;        // [N][4H] transposed bf16 shadow of the recurrent weights, K (= gate columns) contiguous
    const float* dh1; long lddh1;         // direct gradient on h_out(t): dh1[(t*B + b)*lddh1 + u]
    const float* gates; const float* c_new; const float* c_out;   // saved by the forward pass: [T][B][4H], [T][B][H] x2
    float* dg;                            // out: gate pre-activation gradients [T][B][4H]
    float* dc_state;                      // [B][H] dL/dc carried between launches of one pass (chunked ranges)
    float* dbias_part;                    // [row tiles][4H] sum over steps and the tile's rows of dg: the bias gradients (nullable)
    uint32_t site_h, site_c;
    // attention chain (CHAIN_SMA): gradient sources on ctx(t), saved forward quantities, parameters, outputs
    const float* dctx_a; long lddctx_a;   // dctx_a[(t*B + b)*ld + c]: through the projections / decoder LSTM (dDOUT)
    const float* dctx_b; long lddctx_b;   // through the decoder-LSTM input (dDIN)
    const float* dalign;                  // [B,T,Tin] external gradient on the alignments (nullable)
    const float* qs;                      // [T][B][A] processed queries
    const float* pm; const float* memory; int Tin;   // [B,Tin,A], [B,Tin,E]
    const float* psel; const float* align;           // [B,T,Tin] selection probabilities, alignments
    const float* v; const float* wq;      // [A], [A][H]
    float* dctx_out;                      // [T][B][E] total context gradient (d(memory) GEMM)
    float* dq_out;                        // [T][B][2A] one partial per position split (dWq GEMM)
    float* dv_acc;                        // [2][B][A]
    float* dpm_acc;                       // [B][Tin][A]
    // CHAIN_LSA (attention.py:26-85): saved cumulative weights, location-layer weights, per-(split, item) accumulators
    const float* wcum;                    // [B,T,Tin]
    const float* usave; const float* locsave;        // saved by the forward chain: tanh tile [T][B][A][Tin rounded up to 4], location features [T][B][Tin][F]
    const float* loc_conv; const float* loc_dense;   // [F][2][Kc], [A][F]
    float* dconv_acc; float* ddense_acc;  // [2][B][F][2Kc], [2][B][A][F]
    const __bf16* wdt16;                  // [F][A] bf16 transpose of loc_dense (made by the caller in the exchange area)
};
struct ChainBwdDesc {
    ChainBwdStream st[2]; int NS, B, T, t0, t1, H, kind;
    int E, A, F, Kc;
    float drop_p; uint64_t seed;
    unsigned char* X; unsigned char* PB; unsigned* cnt; unsigned* err; unsigned pb_bytes;   // exchange: dg fragments, K-split partials
    unsigned char* PBC; unsigned pbc_bytes; float* DQX; float* CARRYX;                       // attention chain: ctx partials, dq partials, boundary carry (LSA: halo rows of dloc + softmax-dot partials)
    int lds_Tc;                                                                              // positions per split of the longest memory (LDS carve)
};
constexpr size_t kChainBwdCntBytes = (size_t)64 * 128 * kCntShards;     // 64 sharded arrival counters
bool chain_bwd_plan(ChainBwdDesc& d);
size_t chain_bwd_exchange_bytes(const ChainBwdDesc& d, size_t* x_bytes, size_t* pb_bytes);
size_t chain_bwd_lsa_tagged_bytes(const ChainBwdDesc& d);   // LSA: leading part of the carry area (halo rows, tagged softmax-dot slots)
size_t chain_bwd_att_exchange_bytes(const ChainBwdDesc& d, size_t* x_bytes, size_t* pbh_bytes, size_t* pbc_bytes, size_t* dqx_bytes, size_t* carry_bytes);
int chain_bwd(const ChainBwdDesc& d, hipStream_t s);

// Persistent encoder BiLSTM chains (chain_enc.hip): all steps of up to two directions in one launch, exact fp32
struct EncChainDesc {
    int ND, B, T, H;                       // directions (streams), batch, steps, hidden units per direction
    const float* pre[2];                   // [T][B][4H] input-side pre-activations incl. biases
    const float* w_hh[2];                  // [4H][H]
    int reverse[2];
    const int* lengths;                    // [B] or null (unpacked)
    float* h[2]; long ldh;                 // h[s][(t*B + b)*ldh + u]
    float* c[2]; float* gates[2];          // [T][B][H], [T][B][4H]
    float* X; unsigned* cnt; unsigned* err;   // filled by enc_chain_fwd from the exchange space
};
struct EncChainBwdDesc {
    int ND, B, T, H;
    const float* w_hh[2]; int reverse[2];
    const float* c[2]; const float* gates[2];
    const float* dh[2]; long lddh;         // gradient on the outputs: dh[s][(t*B + b)*lddh + u]
    float* dpre[2];                        // out: [T][B][4H]
    float* X; float* PB; unsigned* cnt; unsigned* err;
};
bool enc_chain_covers(int ND, int B, int H);
size_t enc_chain_ws_floats(int ND, int B, int H, int backward);
int enc_chain_fwd(EncChainDesc d, float* ws, size_t ws_floats, hipStream_t s);
int enc_chain_bwd(EncChainBwdDesc d, float* ws, size_t ws_floats, hipStream_t s);

// ------------------------------------------------------------------ decode-step tail (infer.hip)
// projection + stop rule of step t and both prenets of step t+1, one workgroup per batch item
struct StepTailDesc {
    int B, M, P, WO, NS, t;
    int do_proj, do_prenet;
    const float* dout; long lddout;                 // [B,WO] rows [dec_h | ctx | ctx_sub]
    const float* proj_w; const float* proj_b; const float* gate_w; const float* gate_b;
    float* mel_out; long ldmel; float* gate_out; long ldgate;
    float thr; int32_t* stop_index; int32_t* done;  // stop_index nullable = no stop rule
    const float* x_in; long ldx_in;                 // do_proj == 0: prenet input rows [B,M] (null = zeros, the go frame)
    const float* w1t[2]; const float* w2[2];        // prenet weights per stream: first layer TRANSPOSED [M,P], second [P,P]
    float* p1[2]; float* p2[2]; long ldp;           // [B,P] outputs (p1 nullable)
    __bf16* p2_16[2]; long ldp16;                   // optional bf16 copy of p2
    float drop_p; uint64_t seed; uint32_t site1[2], site2[2]; uint32_t drop_base, drop_mstride;   // keep index = base + b*mstride + n
};
int step_tail(const StepTailDesc& d, hipStream_t s);

// ------------------------------------------------------------------ conv + BN stacks, embedding (conv.hip)
struct ConvBnFwd {
    const float* x; int B, T, Cin, Cout, K;      // x: [B*T, Cin] channels-last frames
    const float* w; const float* bias;           // [Cout,Cin,K] (reference layout), [Cout]
    const float* gamma; const float* beta; float* run_mean; float* run_var;   // BatchNorm1d; running stats updated iff training
    int training; float eps; int act; float drop_p; uint64_t seed; uint32_t site;
    const float* residual;                       // optional [B*T, Cout] added after dropout
    float* z; float* mean; float* invstd; float* var;   // saved: conv+bias output [B*T,Cout], batch (or running) stats [Cout]
    float* y;                                    // [B*T, Cout]
    float* wperm; float* scratch;                // [Cout*Cin*K], [128*Cout]
    float* gemm_ws; size_t gemm_ws_bytes;        // optional: bf16 operand staging (gemm.hip)
};
int conv_bn_fwd(const ConvBnFwd& a, hipStream_t s);
struct ConvBnBwd {
    const float* x; int B, T, Cin, Cout, K;
    const float* w; const float* gamma; const float* beta;
    const float* z; const float* mean; const float* invstd;
    int training; float eps; int act; float drop_p; uint64_t seed; uint32_t site;
    const float* dy;                             // [B*T, Cout]
    float* dz;                                   // scratch [B*T, Cout]
    float* dw; float* dbias; float* dgamma; float* dbeta;
    float* dx; int dx_accumulate;                // [B*T, Cin], nullable
    float* wperm; float* scratch; float* gemm_ws; size_t gemm_ws_bytes;
};
int conv_bn_bwd(const ConvBnBwd& a, hipStream_t s);
int embedding_fwd(const long* ids, const float* table, float* out, int rows, int D, hipStream_t s);
int embedding_bwd(const long* ids, const float* dout, float* dtable, int rows, int D, int vocab, hipStream_t s);

// ------------------------------------------------------------------ optimizer (optim.hip)
struct AdamTensor { float* p; const float* g; float* m; float* v; long numel; int first_chunk; int pad_; };   // 48 bytes, mirrors t2_adam_tensor
int adam_chunks(long numel);
// norm_out: 4 floats — [0] total norm, [1] clip coefficient, [2] 1 = update skipped (the device's sticky status word was
// set: an earlier persistent kernel aborted and the gradients are invalid), [3] unused
int adam_norm(const AdamTensor* table_dev, int n_tensors, int n_chunks, float* partial, float* norm_out, float max_norm, hipStream_t s);
int adam_step(const AdamTensor* table_dev, int n_tensors, int n_chunks, float* partial, float* norm_out, float max_norm,
              float lr, float b1, float b2, float eps, float wd, int step, hipStream_t s);

// ------------------------------------------------------------------ elementwise (elementwise.hip)
int rng_keep_mask(uint64_t seed, uint32_t site, uint32_t n, float p, uint8_t* out, hipStream_t s);
int rng_normal(uint64_t seed, uint32_t site, uint32_t n, float* out, hipStream_t s);
// X[t,b,:] = (t == 0) ? 0 : mel[b,:,t-1]   (go frame + teacher forcing shift; mel is [B,M,T])
int teacher_inputs(const float* mel, float* X, int B, int M, int T, hipStream_t s);
// out[b,c,t] = in[b,t,c] with optional padding fill for t >= lengths[b]
int transpose_btc_to_bct(const float* in, float* out, int B, int T, int C, const int* lengths, float fill, hipStream_t s);
int mask_bt(float* x, int B, int T, const int* lengths, float fill, hipStream_t s);
// x[b,t,:] = fill for t >= lengths[b]   (x is [B,T,C])
int mask_btc(float* x, int B, int T, int C, const int* lengths, float fill, hipStream_t s);
int fill_f32(float* p, float v, size_t n, hipStream_t s);
// out[r2, r1, :] = in[r1, r2, :]   ([R1,R2,W] -> [R2,R1,W])
int permute_rows(const float* in, float* out, int R1, int R2, int W, hipStream_t s);
// dz = dy * (y > 0 ? scale : 0)   (ReLU + dropout backward from the saved output; in place allowed)
int relu_drop_bwd(const float* dy, const float* y, float* dz, float scale, size_t n, hipStream_t s);
// out[n] = sum_m X[m*ld + n]  (bias gradients; two fixed-order stages, scratch >= 64*N floats); out2 optional copy
int colsum(const float* X, long ld, int M, int N, float* out, float* out2, float* scratch, hipStream_t s);
// X[r, 0:A] += X[r, A:2A]   (rows of 2A floats)
int fold_halves(float* X, size_t rows, int A, hipStream_t s);
// out[i] = sum_b X[b*n + i]
int batch_sum(const float* X, int B, int n, float* out, hipStream_t s);

}  // namespace t2
