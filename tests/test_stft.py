"""Mel front end: STFT magnitudes against vectors recorded from the reference's stft.py
(tests/golden/make_golden_stft.py); the mel filterbank restates librosa.filters.mel, which is not installed here, so it is
checked through its defining properties only ("parity unpinned" for the filterbank values)."""
import numpy as np
import pytest
import torch

from helpers import load_golden


@pytest.mark.parametrize("name", ["default", "short_window"])
def test_stft_magnitudes_vs_reference(name):
    from tacotron2_subword_amd.stft import STFT
    g = load_golden("stft")
    fl, hop, win = (int(v) for v in g[f"cfg_{name}"])
    mag, phase = STFT(fl, hop, win).transform(torch.from_numpy(g["wave"]))
    ref = g[f"mag_{name}"]
    assert tuple(mag.shape) == ref.shape
    assert float(np.abs(mag.numpy() - ref).max()) < 2e-4 * max(1.0, float(np.abs(ref).max()))
    assert tuple(phase.shape) == ref.shape


def test_mel_filterbank_properties():
    from tacotron2_subword_amd.stft import mel_filterbank, _hz_to_mel, _mel_to_hz
    sr, n_fft, n_mels, fmin, fmax = 22050, 1024, 80, 0.0, 8000.0
    W = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    assert W.shape == (80, 513) and W.dtype == np.float32 and (W >= 0).all()
    freqs = np.linspace(0, sr / 2, 513)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    assert abs(float(_mel_to_hz(_hz_to_mel(440.0))) - 440.0) < 1e-9 and abs(float(_hz_to_mel(1000.0)) - 15.0) < 1e-12   # Slaney scale
    for i in (0, 10, 40, 79):
        nz = np.nonzero(W[i])[0]
        assert freqs[nz[0]] >= edges[i] - 1e-6 and freqs[nz[-1]] <= edges[i + 2] + 1e-6       # support = (f_i, f_{i+2})
        peak = freqs[np.argmax(W[i])]
        assert abs(peak - edges[i + 1]) <= sr / n_fft                                           # peak at the centre frequency
        assert float(W[i].max()) <= 2.0 / (edges[i + 2] - edges[i]) + 1e-9                      # Slaney area normalisation
    assert not W[:, freqs > fmax + sr / n_fft].any()


def test_mel_spectrogram_shapes_and_range():
    from tacotron2_subword_amd.stft import TacotronSTFT
    g = load_golden("stft")
    y = torch.from_numpy(g["wave"])
    mel = TacotronSTFT().mel_spectrogram(y)
    assert tuple(mel.shape) == (2, 80, 1 + y.shape[1] // 256)
    assert float(mel.min()) >= float(np.log(1e-5)) - 1e-6 and bool(torch.isfinite(mel).all())
    with pytest.raises(AssertionError):
        TacotronSTFT().mel_spectrogram(y * 3)


@pytest.mark.gpu
def test_stft_on_gpu_matches_reference():
    from tacotron2_subword_amd.stft import TacotronSTFT
    g = load_golden("stft")
    y = torch.from_numpy(g["wave"])
    cpu = TacotronSTFT()
    gpu = TacotronSTFT().cuda()
    mag, _ = gpu.stft_fn.transform(y.cuda())                      # HIP GEMM for the frames x basis product
    assert float((mag.cpu() - torch.from_numpy(g["mag_default"])).abs().max()) < 5e-4 * float(np.abs(g["mag_default"]).max())
    assert float((gpu.mel_spectrogram(y.cuda()).cpu() - cpu.mel_spectrogram(y)).abs().max()) < 2e-3
