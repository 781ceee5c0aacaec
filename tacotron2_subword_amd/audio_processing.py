"""audio_processing.py:78-93 of the reference: log dynamic-range compression of mel magnitudes."""
import torch


def dynamic_range_compression(x, C=1, clip_val=1e-5):
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression(x, C=1):
    return torch.exp(x) / C
