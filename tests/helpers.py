"""Shared helpers for the GPU parity tests (oracle side on CPU, HIP side through the C ABI)."""
import os

import numpy as np
import torch

from oracle import recipe
from oracle import tacotron2_oracle as O

SMA, LSA, FA2, GMM, DCA = "StepwiseMonotonicAttention", "LSA", "ForwardAttentionV2", "GMMAttention", "DynamicConvolutionAttention"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hp_for(att=SMA, **over):
    hp = O.default_hparams()
    hp["attention"] = att
    hp.update(over)
    return hp


def tiny_hp(att=SMA):
    """Smallest dims the kernels accept (multiples of 64 for the LSTM segments)."""
    return hp_for(att, n_mel_channels=8, symbols_embedding_dim=64, encoder_embedding_dim=64, BERT_embedding_dim=24,
                  decoder_rnn_dim=64, prenet_dim=64, attention_rnn_dim=128, attention_dim=16,
                  attention_location_n_filters=4, attention_location_kernel_size=5, postnet_embedding_dim=64,
                  n_symbols=20, sub_n_symbols=30)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def to_dev(P, device="cuda"):
    return {k: v.detach().to(device).contiguous() for k, v in P.items()}


def maxabs(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max())


def oracle_memories(P, hp, x, training=False, rnd=None):
    """CPU oracle encoder side: memory, memory_sub (what Decoder.forward receives)."""
    text, tl, bl, mels, _, ol, sub_ids, pcls, bcls = x
    with torch.no_grad():
        mem = O.front_end(P, hp, text, tl, pcls, "phone", training, rnd)
        mem_sub = O.front_end(P, hp, sub_ids, bl, bcls, "sub", training, rnd)
    return mem, mem_sub
