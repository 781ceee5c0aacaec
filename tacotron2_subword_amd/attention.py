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


class DynamicConvolutionAttention(nn.Module):
    """attention.py:195-289: 8 static + 8 dynamic 21-tap filters over the previous alignment plus an 11-tap beta-binomial
    prior (alpha 0.1, beta 0.9).  memory_layer exists but takes no part in the arithmetic; P is a buffer."""
    kind = "DCA"

    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, attention_location_n_filters,
                 attention_location_kernel_size):
        super().__init__()
        import numpy as np
        import torch
        from scipy.stats import betabinom
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.score_mask_value = -float("inf")
        static_channels, static_kernel_size, dynamic_channels, dynamic_kernel_size, prior_length = 8, 21, 8, 21, 11
        self.prior_length, self.dynamic_channels, self.dynamic_kernel_size = prior_length, dynamic_channels, dynamic_kernel_size
        prior = betabinom.pmf(np.arange(prior_length), prior_length - 1, 0.1, 0.9)
        self.register_buffer("P", torch.FloatTensor(prior).flip(0))
        self.W = nn.Linear(attention_rnn_dim, attention_dim)
        self.V = nn.Linear(attention_dim, dynamic_channels * dynamic_kernel_size, bias=False)
        self.F = nn.Conv1d(1, static_channels, static_kernel_size, padding=(static_kernel_size - 1) // 2, bias=False)
        self.U = nn.Linear(static_channels, attention_dim, bias=False)
        self.T = nn.Linear(dynamic_channels, attention_dim)
        self.v = nn.Linear(attention_dim, 1, bias=False)


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
