"""utils.py:10-14,28-33 of the reference, device-agnostic."""
import torch


def get_mask_from_lengths(lengths, max_len=None):
    """True where index < length.  (The reference allocates a torch.cuda.LongTensor and syncs
    on .item(); max_len can be passed to avoid the sync.)"""
    if max_len is None:
        max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, device=lengths.device, dtype=lengths.dtype)
    return ids < lengths.unsqueeze(1)


def to_gpu(x):
    x = x.contiguous()
    if torch.cuda.is_available():
        x = x.cuda(non_blocking=True)
    return x
