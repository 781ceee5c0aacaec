"""Attention modules as PARAMETER CONTAINERS with the reference's constructor signature and
state_dict keys (attention.py:7-85, 291-398).  The arithmetic runs inside the fused HIP decoder
(csrc/attention.hip); these classes are never called step by step."""
from torch import nn

from .layers import ConvNorm, LinearNorm


class LocationLayer(nn.Module):
    def __init__(self, attention_n_filters, attention_kernel_size, attention_dim):
        super().__init__()
        self.location_conv = ConvNorm(2, attention_n_filters, kernel_size=attention_kernel_size,
                                      padding=int((attention_kernel_size - 1) / 2), bias=False, stride=1, dilation=1)
        self.location_dense = LinearNorm(attention_n_filters, attention_dim, bias=False, w_init_gain="tanh")


class LocationSensitiveAttention(nn.Module):
    kind = "LSA"

    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, attention_location_n_filters,
                 attention_location_kernel_size):
        super().__init__()
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.v = LinearNorm(attention_dim, 1, bias=False)
        self.location_layer = LocationLayer(attention_location_n_filters, attention_location_kernel_size, attention_dim)
        self.score_mask_value = -float("inf")


class ForwardAttentionV2(nn.Module):
    """attention.py:87-151: same parameters as LSA; score_mask_value = -1e20.  As model.py drives it (log_alpha is
    never updated, model.py:266-270,355) it is the LSA energy followed by a softmax over the first two memory
    positions — that is what the fused decoder computes for it (T2_ATTN_FWD2)."""
    kind = "FWD2"

    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, attention_location_n_filters,
                 attention_location_kernel_size):
        super().__init__()
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.v = LinearNorm(attention_dim, 1, bias=False)
        self.location_layer = LocationLayer(attention_location_n_filters, attention_location_kernel_size, attention_dim)
        self.score_mask_value = -float(1e20)


class GMMAttention(nn.Module):
    """attention.py:401-506 (version '2', K = 5): purely location-based mixture attention.  memory_layer exists (and
    is in the state_dict) but takes no part in the arithmetic, so it never receives a gradient."""
    kind = "GMM"

    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, attention_location_n_filters,
                 attention_location_kernel_size, version="2"):
        super().__init__()
        if version != "2":
            raise NotImplementedError("GMMAttention: only version '2' (the reference's default) is built")
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.score_mask_value = -float("inf")
        self.gmm_version, self.K, self.eps = version, 5, 1e-5
        self.mlp = nn.Sequential(nn.Linear(attention_rnn_dim, attention_dim, bias=True), nn.Tanh(),
                                 nn.Linear(attention_dim, 3 * self.K))


class StepwiseMonotonicAttention(nn.Module):
    kind = "SMA"

    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, attention_location_n_filters,
                 attention_location_kernel_size):
        super().__init__()
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.score_mask_value = -float("inf")
        self.v = nn.Linear(attention_dim, 1, bias=False)
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.sigmoid_noise = 2.0
