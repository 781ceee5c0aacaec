"""Mel front end used for real (non-synthetic) GTA extraction (reference: stft.py:42-105,
layers.py:42-80, audio_processing.py:78-93).

`STFT.transform` = reflect-pad by filter_length/2, frame with stride hop_length, multiply by the windowed Fourier
basis [2*(N/2+1), N] (the reference does this as a strided conv1d), magnitude + phase.  Here the frames x basis
product runs on the HIP GEMM (`ops.gemm`, fp32 MFMA: a 1024-tap basis is a plain [frames, 1024] x [1024, 1026]
product); on CPU tensors it falls back to torch.matmul only so that the formula can be compared with the reference's
own STFT on the GPU-less build box (tests/golden/make_golden_stft.py) — this module is not part of the model's hot path.

The mel filterbank restates librosa.filters.mel (Slaney mel scale, Slaney area normalisation), which the reference
calls as librosa_mel_fn(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax) (layers.py:48-49).  librosa is
not installed here, so the filterbank itself is "parity unpinned" (property tests only); the STFT magnitudes are pinned
by the reference's stft.py."""
import numpy as np
import torch
from scipy.signal import get_window

from .audio_processing import dynamic_range_compression, dynamic_range_decompression


def _pad_center(data, size):
    n = data.shape[-1]
    lpad = (size - n) // 2
    return np.pad(data, (lpad, size - n - lpad), mode="constant")


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """[n_mels, 1 + n_fft//2] triangular filters, Slaney scale and normalisation (librosa.filters.mel defaults)."""
    fmax = float(sr) / 2 if fmax is None else fmax
    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, np.newaxis]).astype(np.float32)


class STFT(torch.nn.Module):
    def __init__(self, filter_length=800, hop_length=200, win_length=800, window="hann"):
        super().__init__()
        self.filter_length, self.hop_length, self.win_length, self.window = filter_length, hop_length, win_length, window
        fourier_basis = np.fft.fft(np.eye(filter_length))
        cutoff = int(filter_length / 2 + 1)
        fourier_basis = np.vstack([np.real(fourier_basis[:cutoff, :]), np.imag(fourier_basis[:cutoff, :])])
        forward_basis = torch.FloatTensor(fourier_basis[:, None, :])
        if window is not None:
            assert filter_length >= win_length
            fft_window = torch.from_numpy(_pad_center(get_window(window, win_length, fftbins=True), filter_length)).float()
            forward_basis *= fft_window
        self.register_buffer("forward_basis", forward_basis.float())       # [2*cutoff, 1, N], the reference's buffer name / shape

    def transform(self, input_data):
        B, n = input_data.shape
        half = int(self.filter_length / 2)
        x = torch.nn.functional.pad(input_data.view(B, 1, 1, n), (half, half, 0, 0), mode="reflect").view(B, -1)
        frames = x.unfold(1, self.filter_length, self.hop_length)           # [B, n_frames, N]
        nf = frames.shape[1]
        basis = self.forward_basis[:, 0, :]                                  # [2*cutoff, N]
        flat = frames.reshape(B * nf, self.filter_length).contiguous()
        if flat.is_cuda:
            from . import ops
            ft = ops.gemm(flat, basis.contiguous(), trans_b=True)            # HIP GEMM: [B*nf, 2*cutoff]
        else:
            ft = flat @ basis.t()
        ft = ft.view(B, nf, -1).transpose(1, 2)                              # [B, 2*cutoff, nf] like the conv1d output
        cutoff = int(self.filter_length / 2 + 1)
        real_part, imag_part = ft[:, :cutoff, :], ft[:, cutoff:, :]
        return torch.sqrt(real_part ** 2 + imag_part ** 2), torch.atan2(imag_part, real_part)


class TacotronSTFT(torch.nn.Module):
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0.0, mel_fmax=8000.0):
        super().__init__()
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        self.register_buffer("mel_basis", torch.from_numpy(mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)).float())

    def spectral_normalize(self, magnitudes):
        return dynamic_range_compression(magnitudes)

    def spectral_de_normalize(self, magnitudes):
        return dynamic_range_decompression(magnitudes)

    def mel_spectrogram(self, y):
        """y: [B, T] in [-1, 1] -> log-mel [B, n_mel_channels, frames] (layers.py:60-80)."""
        assert torch.min(y.data) >= -1
        assert torch.max(y.data) <= 1
        magnitudes, _ = self.stft_fn.transform(y)
        return self.spectral_normalize(torch.matmul(self.mel_basis, magnitudes.data))
