"""utils.py:10-14,28-33 of the reference, device-agnostic."""
import os

import torch


def cpu_quota():
    """CPUs this process may actually use: the smaller of its affinity mask and its cgroup CPU quota (cpu.max of cgroup v2,
    cpu.cfs_quota_us / cpu.cfs_period_us of v1).  A container often SEES every core of the host (os.cpu_count() = 256 on the
    MI355X boxes) while its quota is 16."""
    n = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, float(q) / float(p))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, q / p)
        except (OSError, ValueError):
            pass
    return max(1.0, n)


def fit_cpu_threads(reserve=2):
    """Size torch's intra-op thread pool to the CPU quota (never raises it).  Why it matters for the GPU loop: with the default
    pool (one thread per VISIBLE core) every parallel CPU tensor op wakes hundreds of OpenMP workers that spin for a while
    afterwards; under a CFS quota they burn the whole period's budget in a few milliseconds and the kernel then freezes the
    process — the thread that enqueues GPU work included — until the next 100 ms period.  Measured (round 3, scripts/fed_loop.py):
    a training loop that collates on the host stalled 65-85 ms every third step that way; with the pool sized to the quota
    it runs at the resident-batch speed.  Returns the thread count in effect."""
    # several ranks of one job share the container's quota (torchrun exports LOCAL_WORLD_SIZE; bench.py's own launcher WORLD_SIZE)
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1))
    want = max(1, int(cpu_quota() / ranks) - reserve)
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)
    return torch.get_num_threads()


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
