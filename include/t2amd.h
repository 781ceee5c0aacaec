/* t2amd.h — C ABI of the MI355X-native Tacotron2 (BERT_Tacotron2) hot path.
 *
 * The reference (PhucNguyenAH/tacotron2_subword) is pure Python on PyTorch and has no FFI;
 * its operator boundary for this path is the Python module API (model.py / attention.py).
 * Each entry point below names the reference interface it replaces (file:line, relative to
 * the reference checkout).  A ctypes binding a maintainer would add is shown in
 * INTEGRATION.md; tacotron2_subword_amd/_lib.py is that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; fp32, row-major;
 *   - the caller owns every buffer (workspace sizes come from the *_ws_bytes queries);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no call allocates device
 *     memory; no call synchronises except t2_decoder_infer (documented there);
 *   - return value 0 = OK, < 0 = error; t2_last_error() gives a thread-local message;
 *   - activations are channels-last: [B, T, C].  Conversion to the reference's [B, C, T]
 *     happens in t2_finalize_outputs.
 */
#ifndef T2AMD_H
#define T2AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2_ATTN_SMA 0 /* StepwiseMonotonicAttention, attention.py:291-398 (hparams default) */
#define T2_ATTN_LSA 1 /* LocationSensitiveAttention, attention.py:25-85 */
#define T2_ATTN_DCA 4  /* DynamicConvolutionAttention (attention.py:195-289) */
#define T2_ATTN_GMM 3  /* GMMAttention version '2', K = 5 (attention.py:401-506) */
#define T2_ATTN_FWD2 2 /* ForwardAttentionV2 as model.py drives it (attention.py:87-151 with the never-updated log_alpha of
                          model.py:266-270,355): LSA energies, softmax over the first two positions; LSA weight layout */

/* RNG sites: a dropout keep-bit / noise sample is a pure function of (seed, site, index). */
enum {
    T2_SITE_PRENET1 = 1, T2_SITE_PRENET2 = 2, T2_SITE_PRENET1_SUB = 3, T2_SITE_PRENET2_SUB = 4,
    T2_SITE_ATT_H = 5, T2_SITE_ATT_C = 6, T2_SITE_ATT_H_SUB = 7, T2_SITE_ATT_C_SUB = 8,
    T2_SITE_DEC_H = 9, T2_SITE_DEC_C = 10, T2_SITE_NOISE = 11, T2_SITE_NOISE_SUB = 12,
    T2_SITE_ENC0 = 16,      /* +layer (0..2) */
    T2_SITE_ENCSUB0 = 20,   /* +layer (0..2) */
    T2_SITE_POSTNET0 = 24   /* +layer (0..4) */
};

const char* t2_last_error(void);
/* ABI version: bumped whenever a struct below grows or an argument changes meaning.  2 (round 3): t2_dims carries
 * score_mask_value[_sub], the layouts carry chain / chain_floats, norm_out of t2_adam_* is 4 floats.  A caller compiled
 * against another version passes structs of another size: check t2_version() == T2_ABI_VERSION before anything else.
 * 3: t2_decoder_layout carries usave / usaves / locsave / locsaves (LSA: tanh tile and location features of every step). */
#define T2_ABI_VERSION 3
int t2_version(void);
/* Sticky status of the persistent kernels of the current device (the reference's nearest analogue: train.py:335-340,
 * which at least notices a NaN gradient norm).  A chain whose hand-off timed out writes a non-zero code into a word in
 * page-locked host memory; out_host[0] = that code (0 = fine), read with a plain load: no synchronisation.  While it is
 * non-zero t2_adam_step changes no parameter (norm_out[2] = 1), and t2_decoder_forward / t2_decoder_infer overwrite the
 * outputs of an aborted pass with NaN.  Only t2_chain_status_clear resets it. */
int t2_chain_status(uint32_t* out_host /* [4] */);
int t2_chain_status_clear(void);
/* tests: enqueue a kernel that reports `code` exactly as an aborting chain would */
int t2_debug_report_abort(uint32_t code, void* stream);
/* tests: occupy `workgroups` CUs (96 KB of LDS each, so no persistent workgroup fits beside one) for `milliseconds` on
 * `stream` — what a foreign kernel does to a persistent grid; every wait is bounded */
int t2_debug_occupy(int workgroups, int milliseconds, void* stream);
/* 1 when this process holds the device's claim on persistent kernels (an flock on /tmp/t2amd-persistent-<pci id>.lock,
 * taken at the first pass that could use them; env T2_CHAIN_FORCE=1 skips the test).  A process that does not get it runs
 * the per-step launch path: a persistent grid needs every CU, one process per GPU. */
int t2_chain_claimed(void);
/* Arithmetic type of the GEMM operands: 0 = fp32 (exact fp32 fma chains; the parity path, default),
 * 1 = bf16 operands with fp32 accumulation for the large GEMMs (fp32 storage, converted while
 * staging); recurrent state, BatchNorm statistics and attention recurrences stay fp32. */
int t2_set_precision(int mode);
int t2_get_precision(void);
/* 1 (default): teacher-forced passes run the decoder-LSTM chain on an internal side stream, one chunk of steps
 * apart from the attention chain (fork/join inside the call; the caller's stream semantics are unchanged).  0: one stream. */
int t2_set_overlap(int on);
/* Persistent chain kernels (csrc/chain.hip), bf16 mode, default dims, B <= 128, SMA: 1 (default; env T2_CHAIN=0 turns it
 * off) runs ALL steps of the attention chain (both attention LSTMs + attention; model.py:322-369) and of the decoder-LSTM
 * chain (model.py:371-373) of a teacher-forced pass in one launch each, with the recurrent weights resident in registers
 * and h / ctx / query partials exchanged between workgroups through write-through stores whose 16-byte units carry a step
 * tag the consumers validate (teacher-forced passes; the decode loop and the encoder chains use arrival counters).  They need
 * the whole device (256 co-resident workgroups): ONE process per GPU, as the reference runs (distributed.py:181-200).
 * A chain that could not make progress for 1 s gives up and leaves a non-zero status word in the workspace
 * (t2_decoder_layout.chain); 0: one launch per step and kernel, as in round 1. */
int t2_set_chain(int on);
int t2_get_chain(void);
/* The backward pass's persistent chains (csrc/chain_bwd.hip; in effect only while t2_set_chain is on): 1 (default; env
 * T2_CHAIN_BWD=0 turns it off) runs the BPTT of the decoder LSTM (autograd of model.py:371-373) in one launch per step
 * range, W_hh^T resident in registers.  t2_decoder_backward then writes status word 2 of the forward workspace's chain block. */
int t2_set_chain_bwd(int on);
/* bf16 mode only.  1 (default): a large GEMM whose extents are whole 128x128x64 tiles first writes bf16 copies of its
 * fp32 operands (K contiguous) into the caller's scratch and runs the bf16-source kernel on them (half the operand
 * bytes per MFMA; implicit-conv operands included).  0: always convert while staging through LDS.  Same rounding
 * of the operands either way; only the summation order inside a dot product differs. */
int t2_set_gemm_staging(int on);
/* Makes `stream` wait for everything the library has queued on its internal side stream of the current device
 * (t2_decoder_bwd_args.defer_weight_grads). */
int t2_side_join(void* stream);

/* Model dimensions (hparams.py:55-95). */
typedef struct t2_dims {
    int n_mel;          /* n_mel_channels * n_frames_per_step (80) */
    int prenet_dim;     /* 256 */
    int enc_dim;        /* encoder_embedding_dim E (512) */
    int att_rnn_dim;    /* 1024 */
    int dec_rnn_dim;    /* 1024 */
    int att_dim;        /* attention_dim A (128) */
    int loc_filters;    /* 32 */
    int loc_kernel;     /* 31 */
    int attention_kind; /* T2_ATTN_* */
    float p_att_dropout, p_dec_dropout, p_prenet_dropout; /* 0.1, 0.1, 0.5 */
    int n_streams;      /* 2 = BERT_Tacotron2 (phone + sub-word, model.py:142-207); 1 = classic single-stream Tacotron2
                           (the API GTA.py:6,57-59 expects): decoder_rnn takes [att_h|ctx], projections [dec_h|ctx];
                           the *_sub weights / memory_sub / align_sub are then ignored (0 is read as 2) */
    float score_mask_value, score_mask_value_sub; /* energy written over positions past an item's length: the attention
                           modules' score_mask_value (attention.py:37,79; train.py:77-78 sets the phone stream's to the fp16
                           minimum for fp16 runs).  Taken verbatim when score_mask_given != 0; otherwise 0 = the default, -infinity */
    int score_mask_given;  /* 1: the two values above are meant as written (an intentional 0.0 included); 0 (a zero-initialised
                              struct): a 0.0 there stands for the default */
} t2_dims;

/* Parameters of Decoder (model.py:128-207), reference state_dict names in comments. */
typedef struct t2_attention_weights {
    const float* wq;        /* query_layer.linear_layer.weight        [A, att_rnn] */
    const float* wm;        /* memory_layer.linear_layer.weight       [A, E] */
    const float* v;         /* v.weight | v.linear_layer.weight       [1, A] */
    const float* loc_conv;  /* location_layer.location_conv.conv.weight   [F,2,Kc] (LSA) */
    const float* loc_dense; /* location_layer.location_dense.linear_layer.weight [A,F] (LSA) */
    /* GMMAttention (T2_ATTN_GMM): wq = mlp.0.weight [A, att_rnn]; wm exists in the state_dict but is not used */
    const float* mlp_b1;    /* mlp.0.bias   [A] */
    const float* mlp_w2;    /* mlp.2.weight [15, A] */
    const float* mlp_b2;    /* mlp.2.bias   [15] */
    /* DynamicConvolutionAttention (T2_ATTN_DCA): wq = W.weight [A, att_rnn], mlp_b1 = W.bias [A], mlp_w2 = V.weight [168, A],
     * loc_conv = F.weight [8,1,21], loc_dense = U.weight [A,8], v = v.weight [1,A]; memory_layer is not used */
    const float* dca_T;     /* T.weight [A, 8] */
    const float* dca_bT;    /* T.bias   [A] */
    const float* dca_P;     /* P buffer [11] (prior taps; no gradient) */
} t2_attention_weights;
typedef struct t2_lstm_weights { const float *w_ih, *w_hh, *b_ih, *b_hh; } t2_lstm_weights;
typedef struct t2_decoder_weights {
    const float *prenet_w1, *prenet_w2;          /* decoder.prenet.layers.{0,1}.linear_layer.weight */
    const float *prenet_sub_w1, *prenet_sub_w2;  /* decoder.prenet_bert.layers.{0,1}... */
    t2_lstm_weights att, att_sub;                /* decoder.attention_rnn, decoder.attention_rnn_bert */
    t2_attention_weights attn, attn_sub;         /* decoder.attention_layer, decoder.attention_layer_bert */
    t2_lstm_weights dec;                         /* decoder.decoder_rnn */
    const float *proj_w, *proj_b;                /* decoder.linear_projection.linear_layer.{weight,bias} */
    const float *gate_w, *gate_b;                /* decoder.gate_layer.linear_layer.{weight,bias} */
} t2_decoder_weights;

/* Saved-activation workspace of one decoder pass.  Offsets are in floats from `ws`.
 * Per-frame buffers are TIME-MAJOR [T,B,*] (one step's rows are contiguous: the per-step kernels
 * then touch one or two pages instead of B pages); the alignment-shaped ones stay [B,T,Tin]. */
typedef struct t2_decoder_layout {
    size_t total_floats;
    size_t x, p1, p2, p1s, p2s;      /* [T,B,n_mel], prenet activations [T,B,prenet] */
    size_t pm, pms;                   /* processed memory [B,Tin,A], [B,Tsub,A] */
    size_t prea, preas;               /* attention-LSTM input pre-activations [T,B,4*Ha] */
    size_t ga, gas;                   /* activated gates i,f,g,o [T,B,4*Ha] */
    size_t cna, cnas, ca, cas;        /* cell before / after dropout [T,B,Ha] */
    size_t din;                       /* [T,B, 2*Ha+2*E] = att_h | ctx | att_h_sub | ctx_sub */
    size_t psel, psels;               /* SMA p_t [B,T,Tin], [B,T,Tsub] */
    size_t wcum, wcums;               /* LSA cumulative weights per step [B,T,Tin], [B,T,Tsub]; GMM: mixture means [T,B,8] */
    size_t pred, gd, cnd, cd;         /* decoder LSTM: pre-activations, gates [T,B,4*Hd], cells [T,B,Hd] */
    size_t dout;                      /* [T,B, Hd+2*E] = dec_h | ctx | ctx_sub */
    size_t qs, qss;                   /* processed query per step [T,B,A] */
    size_t qpart;                     /* per-step scratch [2][Ha/8][B][A] */
    size_t w1t;                       /* decode loop: transposed first prenet layers [2][n_mel][P] */
    /* bf16-operand mode (sizes in floats = bf16 elements / 2): weight shadows [W_hh | W_ih[:,P:]] per stream,
     * decoder W_hh, their transposes for the backward pass, and bf16 copies of DIN / dec_h */
    size_t w16a, w16as, w16d, wt16a, wt16as, wt16d, din16, dh16;
    size_t gemm_ws; size_t gemm_ws_floats;
    size_t chain; size_t chain_floats;   /* exchange buffers of the persistent chain kernels (t2_set_chain): word 0 = status of
                                          * the attention chain, word 1 = status of the decoder-LSTM chain (0 = OK), then arrival
                                          * counters, fragment-ordered h / ctx buffers, query partials */
    size_t usave, usaves;             /* LSA: tanh(q + pm + location term) of every step [T,B,A,Tin4], [T,B,A,Tsub4] (Tin4 = Tin rounded up to 4) and ... */
    size_t locsave, locsaves;         /* ... the location features [T,B,Tin,F], [T,B,Tsub,F]: written by the persistent forward
                                       * chain for the persistent backward chain (size 0 for the other attention kinds) */
} t2_decoder_layout;

int t2_decoder_layout_query(const t2_dims* dims, int B, int T, int Tin, int Tsub, t2_decoder_layout* out);

/* Teacher-forced decoder pass — replaces Decoder.forward (model.py:392-428) incl. both Prenets
 * (:13-24), initialize_decoder_states (:223-270) and T calls of Decoder.decode (:322-390). */
typedef struct t2_decoder_fwd_args {
    int B, T, Tin, Tsub;
    const float* memory;      /* [B,Tin,E]   linear_converter output */
    const float* memory_sub;  /* [B,Tsub,E] */
    const int32_t* mem_lengths;  /* [B] or NULL (no mask, as in Decoder.inference) */
    const int32_t* sub_lengths;
    const float* mels;        /* [B,n_mel,T] teacher frames (reference layout) */
    float* mel_out;           /* [B,T,n_mel] */
    float* gate_out;          /* [B,T] */
    float* align;             /* [B,T,Tin] */
    float* align_sub;         /* [B,T,Tsub] */
    float* ws;                /* t2_decoder_layout.total_floats floats */
    int training;             /* LSTM-state dropout + SMA noise on (model.train()) */
    int prenet_dropout;       /* 1 = reference behaviour (always on, model.py:23); 0 = off for deterministic parity */
    uint64_t seed;
    int phase;                /* 0: the whole pass.  1: only the part that does not read the memories (bf16 shadows, teacher
                               * inputs, both prenets, hoisted attention-LSTM input GEMMs) — may run on another stream while
                               * the encoders are still working.  2: the rest, on a workspace phase 1 has filled (the caller
                               * orders the two calls, e.g. with an event). */
} t2_decoder_fwd_args;
int t2_decoder_forward(const t2_dims* dims, const t2_decoder_weights* w, const t2_decoder_fwd_args* a, void* stream);

/* Backward of t2_decoder_forward (the autograd graph PyTorch builds for Decoder.forward in the
 * reference).  Gradients are written (not accumulated) into `g`, which has the shapes of the
 * weights; d_memory / d_memory_sub receive the gradient wrt the encoder memories.
 * Both attention kinds; for LSA the location-layer gradient pointers of t2_attention_grads must be set. */
typedef struct t2_lstm_grads { float *w_ih, *w_hh, *b_ih, *b_hh; } t2_lstm_grads;
typedef struct t2_attention_grads { float *wq, *wm, *v, *loc_conv, *loc_dense, *mlp_b1, *mlp_w2, *mlp_b2, *dca_T, *dca_bT, *dca_P; } t2_attention_grads;
typedef struct t2_decoder_grads {
    float *prenet_w1, *prenet_w2, *prenet_sub_w1, *prenet_sub_w2;
    t2_lstm_grads att, att_sub;
    t2_attention_grads attn, attn_sub;
    t2_lstm_grads dec;
    float *proj_w, *proj_b, *gate_w, *gate_b;
} t2_decoder_grads;
typedef struct t2_decoder_bwd_layout {
    size_t total_floats;
    size_t ddout, ddin, dgd, dga, dgas, dctx, dctxs, dq, dqs, dv, dvs, dpm, dpms, carry, carrys;
    size_t carryc, carrycs, dlconv, dlconvs, dldense, dldenses;   /* LSA: cumulative carry, per-item location-layer gradients; GMM: mean carry, per-item db2 / dW2; zero-sized for SMA */
    size_t dcd, dca, dcas, partd, parta, dp2, dp2s, dp1, dmel_t, dgate_t, dg16a, dg16d, colsum_ws, gemm_ws, gemm_ws_floats;
    size_t chain, chain_floats;   /* exchange buffers of the persistent backward chains (gate-gradient fragments, K-split partials,
                                   * arrival counters); their status words are words 2 (decoder-LSTM chain) and 3 (attention chain)
                                   * of the forward workspace's t2_decoder_layout.chain block */
} t2_decoder_bwd_layout;
int t2_decoder_bwd_layout_query(const t2_dims* dims, int B, int T, int Tin, int Tsub, t2_decoder_bwd_layout* out);
typedef struct t2_decoder_bwd_args {
    int B, T, Tin, Tsub;
    const float* memory; const float* memory_sub;
    const float* align; const float* align_sub;     /* forward outputs [B,T,Tin], [B,T,Tsub] */
    const float* d_mel;        /* [B,T,n_mel] */
    const float* d_gate;       /* [B,T] */
    const float* d_align;      /* [B,T,Tin] or NULL */
    const float* d_align_sub;  /* [B,T,Tsub] or NULL */
    float* d_memory;           /* [B,Tin,E] out */
    float* d_memory_sub;       /* [B,Tsub,E] out */
    const float* ws;           /* workspace filled by t2_decoder_forward */
    float* bws;                /* t2_decoder_bwd_layout.total_floats floats of scratch */
    int training; int prenet_dropout; uint64_t seed;   /* must equal the forward call's */
    int defer_weight_grads;    /* 1 (with t2_set_overlap on, T >= 32): d_memory / d_memory_sub are complete on `stream` when the
                                * call returns, the weight gradients in `g` are still being written on the library's side
                                * stream; call t2_side_join(stream) before reading them or releasing ws / bws.  0: everything
                                * is ordered on `stream`. */
} t2_decoder_bwd_args;
int t2_decoder_backward(const t2_dims* dims, const t2_decoder_weights* w, const t2_decoder_grads* g,
                        const t2_decoder_bwd_args* a, void* stream);

/* Autoregressive decode — replaces Decoder.inference (model.py:430-492) for any B.
 * Per-item stop rule: stop_index[b] = first t with sigmoid(gate) > threshold.
 * Runs until every item has stopped or max_steps.  Every `poll_every` steps a device counter is copied to pinned
 * memory behind an event and the host reads the copy made two polls earlier, so it only blocks when it is more than
 * 2*poll_every steps ahead of the GPU (the queue never drains); the loop therefore runs up to 3*poll_every steps past
 * the last stop.  Returns the number of steps run in *steps_run_host; frames after an item's stop index are
 * computed-but-meaningless.  poll_every <= 0: 16 steps, or 32 with the persistent kernels.
 * bf16 mode, default dims, B <= 32, SMA / LSA, t2_set_chain on: each polling interval is ONE persistent launch
 * (csrc/chain.hip, decode mode) running the whole step — attention LSTMs, attention, decoder LSTM, projections + stop
 * rule, both prenets — with every recurrent weight resident in registers; status word 0 of t2_decoder_layout.chain. */
typedef struct t2_decoder_infer_args {
    int B, Tin, Tsub, max_steps, poll_every;
    float gate_threshold;
    const float* memory; const float* memory_sub;
    const int32_t* mem_lengths; const int32_t* sub_lengths;   /* NULL as in the reference */
    float* mel_out;     /* [B,max_steps,n_mel] */
    float* gate_out;    /* [B,max_steps] */
    float* align;       /* [B,max_steps,Tin] */
    float* align_sub;   /* [B,max_steps,Tsub] */
    int32_t* stop_index;/* [B] device, -1 = never crossed */
    int32_t* done_count;/* [1] device scratch */
    float* ws;          /* layout for T = max_steps */
    int prenet_dropout; uint64_t seed;
    int* steps_run_host;
} t2_decoder_infer_args;
int t2_decoder_infer(const t2_dims* dims, const t2_decoder_weights* w, const t2_decoder_infer_args* a, void* stream);

/* out[b,c,t] = t < lengths[b] ? in[b,t,c] : fill — the layout change + masked_fill of
 * BERT_Tacotron2.parse_output (model.py:531-541).  lengths may be NULL. */
int t2_finalize_bct(const float* in_btc, float* out_bct, int B, int T, int C, const int32_t* lengths, float fill, void* stream);
int t2_mask_bt(float* x, int B, int T, const int32_t* lengths, float fill, void* stream);

/* ---- Encoder / Postnet building blocks (channels-last frames x[B*T, C]) --------------------------
 * One Conv1d(k, same padding) + BatchNorm1d + activation + dropout layer, forward and backward:
 * replaces one nn.Sequential(ConvNorm, BatchNorm1d) + torch.tanh/F.relu + F.dropout of
 * Postnet.forward (model.py:65-70) / Encoder.forward (model.py:97-99).  act: 0 none, 1 relu, 2 tanh.
 * Dropout keep-bit index = (b*T + t)*Cout + c at `site`.  Saved tensors (z, mean, invstd) are
 * caller-owned and passed back to the backward call. */
typedef struct t2_conv_bn_args {
    int B, T, Cin, Cout, K;
    const float* x;                    /* [B*T, Cin] */
    const float* w; const float* bias; /* conv.weight [Cout,Cin,K], conv.bias [Cout] */
    const float* gamma; const float* beta;   /* BatchNorm weight / bias */
    float* run_mean; float* run_var;   /* running statistics: updated in training, read in eval */
    int training; float eps; int act; float drop_p; uint64_t seed; uint32_t site;
    const float* residual;             /* optional, added to the output (mel + postnet(mel), model.py:558) */
    float* z; float* mean; float* invstd; float* var;   /* saved: [B*T,Cout], [Cout] x3 */
    float* y;                          /* [B*T, Cout] */
    float* ws; size_t ws_floats;       /* scratch >= Cout*Cin*K + 128*Cout floats; anything beyond is used for bf16 operand staging (bf16 mode) */
} t2_conv_bn_args;
int t2_conv_bn_forward(const t2_conv_bn_args* a, void* stream);
typedef struct t2_conv_bn_bwd_args {
    int B, T, Cin, Cout, K;
    const float* x; const float* w; const float* gamma; const float* beta;
    const float* z; const float* mean; const float* invstd;
    int training; float eps; int act; float drop_p; uint64_t seed; uint32_t site;
    const float* dy;                   /* [B*T, Cout] */
    float* dw; float* dbias; float* dgamma; float* dbeta;
    float* dx; int dx_accumulate;      /* [B*T, Cin] or NULL */
    float* ws; size_t ws_floats;       /* scratch >= B*T*Cout + Cout*Cin*K + 128*Cout + split-K / bf16 staging space */
} t2_conv_bn_bwd_args;
int t2_conv_bn_backward(const t2_conv_bn_bwd_args* a, void* stream);

/* nn.Embedding forward / weight gradient (model.py:501-502,546,551); ids are int64. */
int t2_embedding_forward(const int64_t* ids, const float* table, float* out, int rows, int dim, void* stream);
int t2_embedding_backward(const int64_t* ids, const float* dout, float* dtable, int rows, int dim, int vocab, void* stream);

/* LSTM recurrences over whole sequences with precomputed input pre-activations (time-major
 * [T,B,4H], biases included): the packed BiLSTM of Encoder.forward (model.py:104-112; lengths given)
 * or the unpacked one of Encoder.inference (:122-123; lengths NULL).  Up to 4 independent
 * sequences ("streams": directions / encoders) advance together, one kernel launch per time step.
 * Outputs beyond an item's length are zero and its state stays zero, which is exactly
 * pack_padded_sequence / pad_packed_sequence semantics for both directions. */
typedef struct t2_lstm_seq_args {
    int nstreams, B, T, H;
    const float* pre[4];     /* [T,B,4H] */
    const float* w_hh[4];    /* [4H,H] */
    int reverse[4];
    const int32_t* lengths;  /* [B] or NULL */
    float* h[4]; long ldh;   /* [T,B,*]: row stride ldh (lets two directions share one [T,B,2H] buffer) */
    float* c[4];             /* [T,B,H] saved cells */
    float* gates[4];         /* [T,B,4H] saved activated gates */
    float* ws; size_t ws_floats;   /* optional exchange space (t2_lstm_seq_chain_ws_floats): with it, H = 256, <= 2 streams,
                                      B <= 128 and t2_set_chain on, ALL steps run in one persistent launch (csrc/chain_enc.hip,
                                      exact fp32: the same arithmetic as the per-step kernels up to summation order) */
} t2_lstm_seq_args;
/* floats of exchange space the persistent BiLSTM chain needs (backward != 0: its BPTT, handed over as t2_lstm_seq_bwd_args.ws) */
size_t t2_lstm_seq_chain_ws_floats(int nstreams, int B, int H, int backward);
int t2_lstm_seq_forward(const t2_lstm_seq_args* a, void* stream);
typedef struct t2_lstm_seq_bwd_args {
    int nstreams, B, T, H;
    const float* w_hh[4]; int reverse[4];
    const float* h[4]; long ldh; const float* c[4]; const float* gates[4];
    const float* dh[4]; long lddh;   /* gradient on the outputs, [T,B,*] with row stride lddh */
    float* dpre[4];                  /* out: gradient wrt pre-activations [T,B,4H] */
    float* dw_hh[4];                 /* out: [4H,H] */
    float* ws; size_t ws_floats;     /* scratch >= nstreams*(B*H + 8*B*H) + split-K space; also the persistent chain's exchange
                                        space when >= t2_lstm_seq_chain_ws_floats(nstreams, B, H, 1) */
} t2_lstm_seq_bwd_args;
int t2_lstm_seq_backward(const t2_lstm_seq_bwd_args* a, void* stream);

/* General GEMM with the full descriptor (see t2_gemm); crow_mod/crow_mul permute output rows:
 * row m is written to (m % crow_mod) * crow_mul + m / crow_mod (0 = identity). */
typedef struct t2_gemm_args {
    const float* A; const float* B; float* C; int M, N, K;
    long sam, sak, sbn, sbk, ldc; int batch; long bsA, bsB, bsC;
    float alpha, beta; const float* bias; int act; int crow_mod; long crow_mul;
    float* ws; size_t ws_bytes; int splitk;
} t2_gemm_args;
int t2_gemm_ex(const t2_gemm_args* a, void* stream);
/* Measurement (bench.py): average milliseconds of `reps` launches of this product on `stream`, bracketed by HIP events —
 * ms_total as t2_gemm_ex runs it (fp32 operands in: bf16 staging casts included), ms_kernel with both bf16 operand copies
 * made beforehand (the matrix kernel and its split-K reduce alone).  bf16 mode, M, N multiples of 128, K of 64, scratch given. */
int t2_prof_gemm(const t2_gemm_args* a, int reps, float* ms_total, float* ms_kernel, void* stream);
/* out[n] = sum_m x[m*ld + n]; scratch >= 64*N floats */
int t2_colsum(const float* x, long ld, int M, int N, float* out, float* scratch, void* stream);
int t2_mask_btc(float* x, int B, int T, int C, const int32_t* lengths, float fill, void* stream);

/* Gradient-norm clipping + Adam over a list of fp32 tensors — replaces torch.nn.utils.clip_grad_norm_ +
 * torch.optim.Adam.step of the training loop (train.py:322-330; Adam with weight decay added to the gradient).
 * `table` is a DEVICE array of n_tensors rows; row i covers chunks [first_chunk, first_chunk + t2_adam_chunks(numel))
 * and the rows are sorted by first_chunk (consecutive).  partial: n_chunks floats of scratch; norm_out: 2 floats
 * (total gradient norm, clip coefficient applied).  max_norm == 0 disables clipping.  step counts from 1.
 * Gradients are read, not modified (the coefficient is applied on the fly).
 * Parameters whose step counts differ (torch.optim.Adam corrects the bias per parameter): t2_adam_norm over the whole
 * table first, then one t2_adam_step per run of rows with equal step, with max_norm < 0 (= take the coefficient
 * t2_adam_norm left in norm_out[1]), `table` pointing at the run's first row and n_chunks = the run's chunk count. */
typedef struct t2_adam_tensor { float* p; const float* g; float* m; float* v; long numel; int first_chunk; int pad_; } t2_adam_tensor;
int t2_adam_chunks(long numel);
int t2_adam_step(const t2_adam_tensor* table, int n_tensors, int n_chunks, float* partial, float* norm_out, float max_norm,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);
int t2_adam_norm(const t2_adam_tensor* table, int n_tensors, int n_chunks, float* partial, float* norm_out, float max_norm, void* stream);
/* norm_out: FOUR floats — [0] total gradient norm, [1] clip coefficient, [2] 1.0 when the update was skipped because the
 * device's sticky status word (t2_chain_status) was set, [3] reserved. */

/* In-situ kernel timing for bench.py's roofline figures: after t2_prof_enable(n) the decoder
 * drivers bracket each per-step kernel launch with HIP events on the launch stream (up to n
 * launches); t2_prof_collect synchronises on the last event and returns total milliseconds and
 * launch counts per kernel kind (host arrays of >= 8 entries):
 * 0 att-LSTM fwd, 1 attention fwd, 2 dec-LSTM fwd, 3 attention bwd, 4 att-LSTM bwd pointwise,
 * 5 att-LSTM bwd GEMM, 6 dec-LSTM bwd pointwise, 7 dec-LSTM bwd GEMM. */
int t2_prof_enable(int max_launches);
int t2_prof_collect(int n_kinds, double* total_ms_host, int* launches_host);

/* Unit-testable pieces. */
int t2_gemm(const float* A, const float* B, float* C, int M, int N, int K,
            long sam, long sak, long sbn, long sbk, long ldc,
            const float* bias, int act, float alpha, float beta,
            float* ws, size_t ws_bytes, int splitk, void* stream);
int t2_rng_keep_mask(uint64_t seed, uint32_t site, uint32_t n, float p, uint8_t* out, void* stream);
int t2_rng_normal(uint64_t seed, uint32_t site, uint32_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* T2AMD_H */
