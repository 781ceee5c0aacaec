#!/usr/bin/env python3
"""Golden STFT magnitudes from the reference's own stft.py (stft.py:42-105), imported with the harness stubs of
make_golden.py plus a real center-pad for librosa.util.pad_center (the only librosa helper the transform uses; with
win_length == filter_length it is the identity).  The mel filterbank cannot be recorded: librosa is not installed."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def waveform(B=2, n=4000, seed=11):
    g = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 22050.0
    x = 0.4 * np.sin(2 * np.pi * 440 * t)[None] + 0.2 * np.sin(2 * np.pi * (1500 + 300 * g.random((B, 1))) * t)[None][0]
    return np.clip(x + 0.05 * g.normal(size=(B, n)), -1, 1).astype(np.float32)


def main():
    import_reference()
    import librosa.util as lu

    def pad_center(data, size, **k):
        n = data.shape[-1]
        lpad = (size - n) // 2
        return np.pad(data, (lpad, size - n - lpad), mode="constant")
    lu.pad_center = pad_center
    import stft as ref_stft  # the reference's (already imported through model -> layers: rebind its copy of the helper)
    ref_stft.pad_center = pad_center
    x = torch.from_numpy(waveform())
    out = {"wave": x.numpy()}
    for name, (fl, hop, win) in {"default": (1024, 256, 1024), "short_window": (512, 128, 400)}.items():
        mag, phase = ref_stft.STFT(fl, hop, win).transform(x)
        out[f"mag_{name}"] = mag.numpy()
        out[f"cfg_{name}"] = np.array([fl, hop, win])
    np.savez_compressed(os.path.join(HERE, "stft.npz"), **out)
    print("wrote stft.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
