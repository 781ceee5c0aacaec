"""ctypes binding of the C-ABI HIP library (include/t2amd.h).

The product path has no CPU fallback: if ``libt2amd.so`` is missing or a call fails, this
module raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``tacotron2_subword_amd.build.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2AMD_LIB") or os.path.join(_HERE, "libt2amd.so")     # (T2AMD_LIB: development builds of the same library)

ATTN_SMA, ATTN_LSA, ATTN_FWD2, ATTN_GMM, ATTN_DCA = 0, 1, 2, 3, 4

SITE = dict(PRENET1=1, PRENET2=2, PRENET1_SUB=3, PRENET2_SUB=4, ATT_H=5, ATT_C=6, ATT_H_SUB=7, ATT_C_SUB=8,
            DEC_H=9, DEC_C=10, NOISE=11, NOISE_SUB=12, ENC0=16, ENCSUB0=20, POSTNET0=24)

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class Dims(C.Structure):
    _fields_ = [("n_mel", C.c_int), ("prenet_dim", C.c_int), ("enc_dim", C.c_int), ("att_rnn_dim", C.c_int),
                ("dec_rnn_dim", C.c_int), ("att_dim", C.c_int), ("loc_filters", C.c_int), ("loc_kernel", C.c_int),
                ("attention_kind", C.c_int), ("p_att_dropout", C.c_float), ("p_dec_dropout", C.c_float),
                ("p_prenet_dropout", C.c_float), ("n_streams", C.c_int),
                ("score_mask_value", C.c_float), ("score_mask_value_sub", C.c_float), ("score_mask_given", C.c_int)]


class AttentionWeights(C.Structure):
    _fields_ = [("wq", C.c_void_p), ("wm", C.c_void_p), ("v", C.c_void_p), ("loc_conv", C.c_void_p), ("loc_dense", C.c_void_p),
                ("mlp_b1", C.c_void_p), ("mlp_w2", C.c_void_p), ("mlp_b2", C.c_void_p),
                ("dca_T", C.c_void_p), ("dca_bT", C.c_void_p), ("dca_P", C.c_void_p)]


class LstmWeights(C.Structure):
    _fields_ = [("w_ih", C.c_void_p), ("w_hh", C.c_void_p), ("b_ih", C.c_void_p), ("b_hh", C.c_void_p)]


class DecoderWeights(C.Structure):
    _fields_ = [("prenet_w1", C.c_void_p), ("prenet_w2", C.c_void_p), ("prenet_sub_w1", C.c_void_p), ("prenet_sub_w2", C.c_void_p),
                ("att", LstmWeights), ("att_sub", LstmWeights), ("attn", AttentionWeights), ("attn_sub", AttentionWeights),
                ("dec", LstmWeights), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("gate_w", C.c_void_p), ("gate_b", C.c_void_p)]


_LAYOUT_FIELDS = ["total_floats", "x", "p1", "p2", "p1s", "p2s", "pm", "pms", "prea", "preas", "ga", "gas",
                  "cna", "cnas", "ca", "cas", "din", "psel", "psels", "wcum", "wcums", "pred", "gd", "cnd", "cd",
                  "dout", "qs", "qss", "qpart", "w1t", "w16a", "w16as", "w16d", "wt16a", "wt16as", "wt16d", "din16", "dh16",
                  "gemm_ws", "gemm_ws_floats", "chain", "chain_floats", "usave", "usaves", "locsave", "locsaves"]


class DecoderLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in _LAYOUT_FIELDS]


class DecoderFwdArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("Tin", C.c_int), ("Tsub", C.c_int),
                ("memory", C.c_void_p), ("memory_sub", C.c_void_p), ("mem_lengths", C.c_void_p), ("sub_lengths", C.c_void_p),
                ("mels", C.c_void_p), ("mel_out", C.c_void_p), ("gate_out", C.c_void_p), ("align", C.c_void_p),
                ("align_sub", C.c_void_p), ("ws", C.c_void_p), ("training", C.c_int), ("prenet_dropout", C.c_int),
                ("seed", C.c_uint64), ("phase", C.c_int)]


class DecoderInferArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("Tin", C.c_int), ("Tsub", C.c_int), ("max_steps", C.c_int), ("poll_every", C.c_int),
                ("gate_threshold", C.c_float),
                ("memory", C.c_void_p), ("memory_sub", C.c_void_p), ("mem_lengths", C.c_void_p), ("sub_lengths", C.c_void_p),
                ("mel_out", C.c_void_p), ("gate_out", C.c_void_p), ("align", C.c_void_p), ("align_sub", C.c_void_p),
                ("stop_index", C.c_void_p), ("done_count", C.c_void_p), ("ws", C.c_void_p),
                ("prenet_dropout", C.c_int), ("seed", C.c_uint64), ("steps_run_host", C.POINTER(C.c_int))]


class LstmGrads(C.Structure):
    _fields_ = LstmWeights._fields_


class AttentionGrads(C.Structure):
    _fields_ = AttentionWeights._fields_


class DecoderGrads(C.Structure):
    _fields_ = [("prenet_w1", C.c_void_p), ("prenet_w2", C.c_void_p), ("prenet_sub_w1", C.c_void_p), ("prenet_sub_w2", C.c_void_p),
                ("att", LstmGrads), ("att_sub", LstmGrads), ("attn", AttentionGrads), ("attn_sub", AttentionGrads),
                ("dec", LstmGrads), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("gate_w", C.c_void_p), ("gate_b", C.c_void_p)]


_BWD_LAYOUT_FIELDS = ["total_floats", "ddout", "ddin", "dgd", "dga", "dgas", "dctx", "dctxs", "dq", "dqs", "dv", "dvs",
                      "dpm", "dpms", "carry", "carrys", "carryc", "carrycs", "dlconv", "dlconvs", "dldense", "dldenses", "dcd", "dca", "dcas", "partd", "parta", "dp2", "dp2s", "dp1",
                      "dmel_t", "dgate_t", "dg16a", "dg16d", "colsum_ws", "gemm_ws", "gemm_ws_floats", "chain", "chain_floats"]


class DecoderBwdLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in _BWD_LAYOUT_FIELDS]


class DecoderBwdArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("Tin", C.c_int), ("Tsub", C.c_int),
                ("memory", C.c_void_p), ("memory_sub", C.c_void_p), ("align", C.c_void_p), ("align_sub", C.c_void_p),
                ("d_mel", C.c_void_p), ("d_gate", C.c_void_p), ("d_align", C.c_void_p), ("d_align_sub", C.c_void_p),
                ("d_memory", C.c_void_p), ("d_memory_sub", C.c_void_p), ("ws", C.c_void_p), ("bws", C.c_void_p),
                ("training", C.c_int), ("prenet_dropout", C.c_int), ("seed", C.c_uint64), ("defer_weight_grads", C.c_int)]


class ConvBnArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("K", C.c_int),
                ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("run_mean", C.c_void_p), ("run_var", C.c_void_p),
                ("training", C.c_int), ("eps", C.c_float), ("act", C.c_int), ("drop_p", C.c_float), ("seed", C.c_uint64),
                ("site", C.c_uint32), ("residual", C.c_void_p),
                ("z", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("var", C.c_void_p), ("y", C.c_void_p),
                ("ws", C.c_void_p), ("ws_floats", C.c_size_t)]


class ConvBnBwdArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("K", C.c_int),
                ("x", C.c_void_p), ("w", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("z", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("training", C.c_int), ("eps", C.c_float), ("act", C.c_int), ("drop_p", C.c_float), ("seed", C.c_uint64),
                ("site", C.c_uint32), ("dy", C.c_void_p),
                ("dw", C.c_void_p), ("dbias", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("dx", C.c_void_p), ("dx_accumulate", C.c_int), ("ws", C.c_void_p), ("ws_floats", C.c_size_t)]


_P4 = C.c_void_p * 4
_I4 = C.c_int * 4


class LstmSeqArgs(C.Structure):
    _fields_ = [("nstreams", C.c_int), ("B", C.c_int), ("T", C.c_int), ("H", C.c_int),
                ("pre", _P4), ("w_hh", _P4), ("reverse", _I4), ("lengths", C.c_void_p),
                ("h", _P4), ("ldh", C.c_long), ("c", _P4), ("gates", _P4), ("ws", C.c_void_p), ("ws_floats", C.c_size_t)]


class LstmSeqBwdArgs(C.Structure):
    _fields_ = [("nstreams", C.c_int), ("B", C.c_int), ("T", C.c_int), ("H", C.c_int),
                ("w_hh", _P4), ("reverse", _I4), ("h", _P4), ("ldh", C.c_long), ("c", _P4), ("gates", _P4),
                ("dh", _P4), ("lddh", C.c_long), ("dpre", _P4), ("dw_hh", _P4), ("ws", C.c_void_p), ("ws_floats", C.c_size_t)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("sam", C.c_long), ("sak", C.c_long), ("sbn", C.c_long), ("sbk", C.c_long), ("ldc", C.c_long),
                ("batch", C.c_int), ("bsA", C.c_long), ("bsB", C.c_long), ("bsC", C.c_long),
                ("alpha", C.c_float), ("beta", C.c_float), ("bias", C.c_void_p), ("act", C.c_int),
                ("crow_mod", C.c_int), ("crow_mul", C.c_long), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("splitk", C.c_int)]


# every symbol include/t2amd.h declares (tests/test_abi.py checks the library exports them all)
ABI_VERSION = 3      # include/t2amd.h T2_ABI_VERSION: struct sizes below match that header and nothing else

EXPORTS = ["t2_last_error", "t2_version", "t2_chain_status", "t2_chain_status_clear", "t2_debug_report_abort", "t2_debug_occupy", "t2_chain_claimed", "t2_set_precision", "t2_get_precision", "t2_set_overlap", "t2_set_chain", "t2_get_chain", "t2_set_chain_bwd", "t2_set_gemm_staging", "t2_side_join", "t2_decoder_layout_query", "t2_decoder_forward", "t2_decoder_infer",
           "t2_decoder_bwd_layout_query", "t2_decoder_backward", "t2_prof_enable", "t2_prof_collect", "t2_adam_chunks", "t2_adam_step", "t2_adam_norm",
           "t2_conv_bn_forward", "t2_conv_bn_backward", "t2_embedding_forward", "t2_embedding_backward",
           "t2_lstm_seq_forward", "t2_lstm_seq_backward", "t2_lstm_seq_chain_ws_floats", "t2_gemm_ex", "t2_prof_gemm", "t2_colsum", "t2_mask_btc",
           "t2_finalize_bct", "t2_mask_bt", "t2_gemm", "t2_rng_keep_mask", "t2_rng_normal"]

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension must be built (see __graft_entry__.build); "
                               "there is no CPU fallback for the product path")
        L = C.CDLL(LIB_PATH)
        L.t2_last_error.restype = C.c_char_p
        if L.t2_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} speaks ABI version {L.t2_version()}, this binding {ABI_VERSION}: rebuild the library "
                               "(the argument structs differ in size between versions)")
        L.t2_chain_status.argtypes = [C.POINTER(C.c_uint32)]
        L.t2_debug_report_abort.argtypes = [C.c_uint32, C.c_void_p]
        L.t2_debug_occupy.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.t2_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long,
                              C.c_long, C.c_long, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_size_t,
                              C.c_int, C.c_void_p]
        L.t2_rng_keep_mask.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]
        L.t2_rng_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.t2_decoder_layout_query.argtypes = [C.POINTER(Dims), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(DecoderLayout)]
        L.t2_decoder_forward.argtypes = [C.POINTER(Dims), C.POINTER(DecoderWeights), C.POINTER(DecoderFwdArgs), C.c_void_p]
        L.t2_decoder_infer.argtypes = [C.POINTER(Dims), C.POINTER(DecoderWeights), C.POINTER(DecoderInferArgs), C.c_void_p]
        L.t2_decoder_bwd_layout_query.argtypes = [C.POINTER(Dims), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(DecoderBwdLayout)]
        L.t2_decoder_backward.argtypes = [C.POINTER(Dims), C.POINTER(DecoderWeights), C.POINTER(DecoderGrads),
                                          C.POINTER(DecoderBwdArgs), C.c_void_p]
        L.t2_conv_bn_forward.argtypes = [C.POINTER(ConvBnArgs), C.c_void_p]
        L.t2_conv_bn_backward.argtypes = [C.POINTER(ConvBnBwdArgs), C.c_void_p]
        L.t2_embedding_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.t2_embedding_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.t2_lstm_seq_forward.argtypes = [C.POINTER(LstmSeqArgs), C.c_void_p]
        L.t2_lstm_seq_backward.argtypes = [C.POINTER(LstmSeqBwdArgs), C.c_void_p]
        L.t2_lstm_seq_chain_ws_floats.argtypes, L.t2_lstm_seq_chain_ws_floats.restype = [C.c_int, C.c_int, C.c_int, C.c_int], C.c_size_t
        L.t2_gemm_ex.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
        L.t2_prof_gemm.argtypes = [C.POINTER(GemmArgs), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]
        L.t2_colsum.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.t2_mask_btc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
        L.t2_prof_enable.argtypes = [C.c_int]
        L.t2_prof_collect.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.t2_finalize_bct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
        L.t2_mask_bt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"t2amd call failed ({rc}): {lib().t2_last_error().decode()}")


def ptr(t: torch.Tensor | None) -> int | None:
    """Device pointer of a contiguous CUDA(HIP) tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("t2amd: tensors must live on the GPU (no CPU fallback in the product path)")
    if not t.is_contiguous():
        raise RuntimeError("t2amd: tensor must be contiguous")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def dims_from_hparams(hp, n_streams: int = 2) -> Dims:
    g = (lambda k: hp[k]) if isinstance(hp, dict) else (lambda k: getattr(hp, k))
    kind = {"StepwiseMonotonicAttention": ATTN_SMA, "ForwardAttentionV2": ATTN_FWD2, "GMMAttention": ATTN_GMM,
            "DynamicConvolutionAttention": ATTN_DCA}.get(g("attention"), ATTN_LSA)
    return Dims(int(g("n_mel_channels")) * int(g("n_frames_per_step")), int(g("prenet_dim")), int(g("encoder_embedding_dim")),
                int(g("attention_rnn_dim")), int(g("decoder_rnn_dim")), int(g("attention_dim")),
                int(g("attention_location_n_filters")), int(g("attention_location_kernel_size")), kind,
                float(g("p_attention_dropout")), float(g("p_decoder_dropout")), 0.5, n_streams)


def decoder_weights(P: dict, kind: int, prefix: str = "decoder.", single: bool = False) -> DecoderWeights:
    """Pack device pointers of a (reference-keyed) state dict; the tensors must outlive the call.
    single: classic one-stream decoder (no *_bert modules)."""
    p = lambda k: ptr(P[prefix + k])
    lstm = lambda n: LstmWeights(p(n + ".weight_ih"), p(n + ".weight_hh"), p(n + ".bias_ih"), p(n + ".bias_hh"))

    def attn(n):
        if kind == ATTN_SMA:
            return AttentionWeights(p(n + ".query_layer.linear_layer.weight"), p(n + ".memory_layer.linear_layer.weight"),
                                    p(n + ".v.weight"), None, None)
        if kind == ATTN_DCA:       # W plays the part of the query projection (attention.py:222, 244)
            return AttentionWeights(p(n + ".W.weight"), None, p(n + ".v.weight"), p(n + ".F.weight"), p(n + ".U.weight"),
                                    p(n + ".W.bias"), p(n + ".V.weight"), None, p(n + ".T.weight"), p(n + ".T.bias"), p(n + ".P"))
        if kind == ATTN_GMM:       # mlp.0 plays the part of the query projection (attention.py:412-415)
            return AttentionWeights(p(n + ".mlp.0.weight"), None, None, None, None,       # memory_layer is never read
                                    p(n + ".mlp.0.bias"), p(n + ".mlp.2.weight"), p(n + ".mlp.2.bias"))
        return AttentionWeights(p(n + ".query_layer.linear_layer.weight"), p(n + ".memory_layer.linear_layer.weight"),
                                p(n + ".v.linear_layer.weight"), p(n + ".location_layer.location_conv.conv.weight"),
                                p(n + ".location_layer.location_dense.linear_layer.weight"))
    none_l, none_a = LstmWeights(None, None, None, None), AttentionWeights()
    return DecoderWeights(p("prenet.layers.0.linear_layer.weight"), p("prenet.layers.1.linear_layer.weight"),
                          None if single else p("prenet_bert.layers.0.linear_layer.weight"),
                          None if single else p("prenet_bert.layers.1.linear_layer.weight"),
                          lstm("attention_rnn"), none_l if single else lstm("attention_rnn_bert"),
                          attn("attention_layer"), none_a if single else attn("attention_layer_bert"),
                          lstm("decoder_rnn"), p("linear_projection.linear_layer.weight"), p("linear_projection.linear_layer.bias"),
                          p("gate_layer.linear_layer.weight"), p("gate_layer.linear_layer.bias"))


DECODER_PARAM_KEYS_SMA = [
    "prenet.layers.0.linear_layer.weight", "prenet.layers.1.linear_layer.weight",
    "prenet_bert.layers.0.linear_layer.weight", "prenet_bert.layers.1.linear_layer.weight",
    "attention_rnn.weight_ih", "attention_rnn.weight_hh", "attention_rnn.bias_ih", "attention_rnn.bias_hh",
    "attention_rnn_bert.weight_ih", "attention_rnn_bert.weight_hh", "attention_rnn_bert.bias_ih", "attention_rnn_bert.bias_hh",
    "attention_layer.query_layer.linear_layer.weight", "attention_layer.memory_layer.linear_layer.weight", "attention_layer.v.weight",
    "attention_layer_bert.query_layer.linear_layer.weight", "attention_layer_bert.memory_layer.linear_layer.weight",
    "attention_layer_bert.v.weight",
    "decoder_rnn.weight_ih", "decoder_rnn.weight_hh", "decoder_rnn.bias_ih", "decoder_rnn.bias_hh",
    "linear_projection.linear_layer.weight", "linear_projection.linear_layer.bias",
    "gate_layer.linear_layer.weight", "gate_layer.linear_layer.bias"]


def _lsa_keys():
    out = []
    for k in DECODER_PARAM_KEYS_SMA:
        if k.endswith(".v.weight"):
            n = k[:-len(".v.weight")]
            out += [n + ".v.linear_layer.weight", n + ".location_layer.location_conv.conv.weight",
                    n + ".location_layer.location_dense.linear_layer.weight"]
        else:
            out.append(k)
    return out


DECODER_PARAM_KEYS_LSA = _lsa_keys()


def _gmm_keys():
    """Parameters that receive a gradient with GMMAttention: its memory_layer does not (never used, attention.py:401-506)."""
    out = []
    for k in DECODER_PARAM_KEYS_SMA:
        if ".query_layer." in k:
            n = k[:k.index(".query_layer.")]
            out += [n + ".mlp.0.weight", n + ".mlp.0.bias", n + ".mlp.2.weight", n + ".mlp.2.bias"]
        elif ".memory_layer." in k or k.endswith(".v.weight"):
            continue
        else:
            out.append(k)
    return out


DECODER_PARAM_KEYS_GMM = _gmm_keys()


def _dca_keys():
    """Parameters that receive a gradient with DynamicConvolutionAttention (memory_layer does not; P is a buffer)."""
    out = []
    for k in DECODER_PARAM_KEYS_SMA:
        if ".query_layer." in k:
            n = k[:k.index(".query_layer.")]
            out += [n + s for s in (".W.weight", ".W.bias", ".V.weight", ".F.weight", ".U.weight", ".T.weight", ".T.bias", ".v.weight")]
        elif ".memory_layer." in k or k.endswith(".v.weight"):
            continue
        else:
            out.append(k)
    return out


DECODER_PARAM_KEYS_DCA = _dca_keys()


def decoder_param_keys(kind: int, single: bool = False):
    keys = {ATTN_SMA: DECODER_PARAM_KEYS_SMA, ATTN_GMM: DECODER_PARAM_KEYS_GMM, ATTN_DCA: DECODER_PARAM_KEYS_DCA}.get(kind, DECODER_PARAM_KEYS_LSA)
    return [k for k in keys if "_bert" not in k] if single else keys


def decoder_grads(G: dict, prefix: str = "decoder.", single: bool = False, kind: int = ATTN_SMA) -> DecoderGrads:
    """Pack pointers of gradient buffers keyed like the weights."""
    p = lambda k: ptr(G[prefix + k])
    lstm = lambda n: LstmGrads(p(n + ".weight_ih"), p(n + ".weight_hh"), p(n + ".bias_ih"), p(n + ".bias_hh"))

    def attn(n):
        if kind == ATTN_SMA:
            return AttentionGrads(p(n + ".query_layer.linear_layer.weight"), p(n + ".memory_layer.linear_layer.weight"),
                                  p(n + ".v.weight"), None, None)
        if kind == ATTN_DCA:
            return AttentionGrads(p(n + ".W.weight"), None, p(n + ".v.weight"), p(n + ".F.weight"), p(n + ".U.weight"),
                                  p(n + ".W.bias"), p(n + ".V.weight"), None, p(n + ".T.weight"), p(n + ".T.bias"), None)
        if kind == ATTN_GMM:
            return AttentionGrads(p(n + ".mlp.0.weight"), None, None, None, None,
                                  p(n + ".mlp.0.bias"), p(n + ".mlp.2.weight"), p(n + ".mlp.2.bias"))
        return AttentionGrads(p(n + ".query_layer.linear_layer.weight"), p(n + ".memory_layer.linear_layer.weight"),
                              p(n + ".v.linear_layer.weight"), p(n + ".location_layer.location_conv.conv.weight"),
                              p(n + ".location_layer.location_dense.linear_layer.weight"))
    none_l, none_a = LstmGrads(None, None, None, None), AttentionGrads()
    return DecoderGrads(p("prenet.layers.0.linear_layer.weight"), p("prenet.layers.1.linear_layer.weight"),
                        None if single else p("prenet_bert.layers.0.linear_layer.weight"),
                        None if single else p("prenet_bert.layers.1.linear_layer.weight"),
                        lstm("attention_rnn"), none_l if single else lstm("attention_rnn_bert"),
                        attn("attention_layer"), none_a if single else attn("attention_layer_bert"),
                        lstm("decoder_rnn"), p("linear_projection.linear_layer.weight"), p("linear_projection.linear_layer.bias"),
                        p("gate_layer.linear_layer.weight"), p("gate_layer.linear_layer.bias"))


def decoder_bwd_layout(dims: Dims, B: int, T: int, Tin: int, Tsub: int) -> DecoderBwdLayout:
    L = DecoderBwdLayout()
    check(lib().t2_decoder_bwd_layout_query(C.byref(dims), B, T, Tin, Tsub, C.byref(L)))
    return L


def decoder_layout(dims: Dims, B: int, T: int, Tin: int, Tsub: int) -> DecoderLayout:
    L = DecoderLayout()
    check(lib().t2_decoder_layout_query(C.byref(dims), B, T, Tin, Tsub, C.byref(L)))
    return L


PROF_KINDS = ["att_lstm_fwd", "attention_fwd", "dec_lstm_fwd", "attention_bwd", "att_lstm_bwd_pointwise",
              "att_lstm_bwd_gemm", "dec_lstm_bwd_pointwise", "dec_lstm_bwd_gemm", "chain_a_fwd", "chain_b_fwd", "chain_b_bwd", "chain_a_bwd", "chain_dec"]


def prof_enable(max_launches: int) -> None:
    check(lib().t2_prof_enable(max_launches))


def prof_collect() -> dict:
    """{kind: (total_ms, launches)} of the launches recorded since prof_enable (HIP events on the launch stream)."""
    n = len(PROF_KINDS)
    ms, cnt = (C.c_double * n)(), (C.c_int * n)()
    check(lib().t2_prof_collect(n, ms, cnt))
    return {k: (ms[i], cnt[i]) for i, k in enumerate(PROF_KINDS)}


def set_chain(on: bool) -> None:
    """Persistent chain kernels for teacher-forced passes (include/t2amd.h: t2_set_chain)."""
    check(lib().t2_set_chain(int(bool(on))))


def set_chain_bwd(on: bool) -> None:
    check(lib().t2_set_chain_bwd(int(bool(on))))


def get_chain() -> bool:
    return bool(lib().t2_get_chain())


def set_precision(mode: str) -> None:
    """"f32": exact fp32 GEMMs (parity path, default).  "bf16": bf16 operands / fp32 accumulate for large GEMMs."""
    check(lib().t2_set_precision({"f32": 0, "fp32": 0, "bf16": 1}[mode]))


def get_precision() -> str:
    return "bf16" if lib().t2_get_precision() == 1 else "f32"
