"""Shared by the dev / measurement scripts: decoder weights and dims built from the PRODUCT modules (random init, default
hparams) — nothing here touches oracle/ or tests/."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_subword_amd import _lib as L            # noqa: E402
from tacotron2_subword_amd.hparams import create_hparams  # noqa: E402
from tacotron2_subword_amd.model import BERT_Tacotron2  # noqa: E402

SMA, LSA = "StepwiseMonotonicAttention", "LSA"


def hp_for(att):
    hp = create_hparams()
    hp.attention = att
    return hp


def decoder_setup(att, seed=1234):
    """(hparams, state dict on the device keyed like the reference, dims, packed weight pointers)"""
    hp = hp_for(att)
    torch.manual_seed(seed)
    model = BERT_Tacotron2(hp).cuda()
    P = {k: v.detach() for k, v in model.state_dict().items()}
    dims = model.decoder.dims
    W = L.decoder_weights(P, dims.attention_kind)
    return hp, P, dims, W
