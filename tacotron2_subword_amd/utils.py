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


def create_alignment(base_mat, duration_predictor_output):
    """utils.py:108-117: base_mat[i, frames of phone j, j] = 1 where phone j of item i lasts duration[i, j] frames
    (frames laid end to end).  Same result as the reference's triple Python loop, built from a cumulative sum."""
    dur = duration_predictor_output.long()
    end = torch.cumsum(dur, dim=1)                       # [B, n_phones]
    start = end - dur
    t = torch.arange(base_mat.size(1), device=dur.device)[None, :, None]
    hit = (t >= start[:, None, :]) & (t < end[:, None, :])
    base_mat[hit.to(base_mat.device)] = 1
    return base_mat


class Alignment_Generator(torch.nn.Module):
    """utils.py:92-106: durations [B, n_phones] -> hard alignment [B, max total frames, n_phones] (on the CPU, like
    the reference's torch.zeros)."""

    def LR(self, duration_predictor_output):
        frame_lens = torch.sum(duration_predictor_output, -1)
        expand_max_frame_len = int(torch.max(frame_lens, -1)[0])
        alignment = torch.zeros(duration_predictor_output.size(0), expand_max_frame_len, duration_predictor_output.size(1))
        return create_alignment(alignment, duration_predictor_output)

    def forward(self, duration_predictor_output):
        return self.LR(duration_predictor_output)
